"""MI355X-native counterparts of the ``sbmc.modules`` pieces used by the hot path
(``ConvChain``, ``Autoencoder``, ``KernelApply``); constructor signatures follow the
call sites ``support/networks.py:18-24`` and ``train_kpcn.py:213``.

``sbmc`` is not part of the reference tree, so the definitions are
this build's stated specification -- identical, parameter name for parameter name, to
the CPU oracle ``oracle/modules.py`` they are parity-tested against.  Parameters live in
ordinary ``nn.Conv2d`` containers (``layers.<i>.weight|bias``, OIHW) so ``state_dict()``,
``optim.Adam`` and ``clip_grad_value_`` in an unmodified caller keep working; the
arithmetic runs in ``libwcmc_hip.so`` only.
"""
import torch
import torch.nn as nn

from . import ops


class _WeightNormConv(nn.Module):
    """Parameter container of one weight-normalised layer: ``weight = weight_g * weight_v / ||weight_v||`` with the norm
    over (in, kh, kw) per output channel -- ``torch.nn.utils.weight_norm(nn.Conv2d(...))``'s parametrisation and parameter
    names.  The effective weight comes from ``ops.weight_norm_multi`` (``wcmc_weight_norm_fwd`` / ``_bwd``): one launch for
    every layer of the enclosing model when a ``weight_norm_scope`` is open, else one per chain."""

    def __init__(self, cin, cout, ksize):
        super().__init__()
        # (registration order = torch.nn.utils.weight_norm(nn.Conv2d(...))'s: bias, weight_g, weight_v -- so that
        # ``parameters()`` and ``state_dict()`` enumerate like a checkpoint written by upstream sbmc)
        self.bias = nn.Parameter(torch.zeros(cout))
        self.weight_g = nn.Parameter(torch.ones(cout, 1, 1, 1))
        self.weight_v = nn.Parameter(torch.empty(cout, cin, ksize, ksize))
        self._w = None              # the effective weight of the forward in progress (weight_norm_scope)

    @property
    def weight(self):
        if self._w is not None:
            return self._w
        return ops.weight_norm_multi([self.weight_g], [self.weight_v])[0]


class weight_norm_scope:
    """``with weight_norm_scope(model): y = model_forward(...)``: the effective weights of ALL weight-normalised layers under
    `model` are formed by one launch on entry (one autograd node, whose backward is one launch too) and handed to the chains
    for the duration of the forward.  Nested scopes and models without such layers are no-ops."""

    def __init__(self, model):
        layers = getattr(model, "_wn_layers", None)
        if layers is None:
            layers = [m for m in model.modules() if isinstance(m, _WeightNormConv)]
            object.__setattr__(model, "_wn_layers", layers)         # (a plain attribute: not a submodule list)
        self.layers = layers if layers and layers[0]._w is None else []

    def __enter__(self):
        if self.layers:
            ws = ops.weight_norm_multi([l.weight_g for l in self.layers], [l.weight_v for l in self.layers])
            for l, w in zip(self.layers, ws):
                l._w = w
        return self

    def __exit__(self, *exc):
        for l in self.layers:
            l._w = None
        return False


class ConvChain(nn.Module):
    """``weight_norm`` defaults to True as upstream adobe/sbmc's ``ConvChain`` does (``sbmc`` is absent from the reference
    tree, so this is the published code as this build, the survey and the round-4 review all read it): ``sbmc.KPCN`` passes
    ``weight_norm=False`` explicitly, ``support/networks.py:18-24`` (PathNet) passes nothing and so trains the normalised
    parametrisation; checkpoints then carry ``weight_g`` / ``weight_v`` per layer."""

    def __init__(self, ninputs, noutputs, ksize=3, width=64, depth=3, pad=True,
                 activation="relu", output_type="linear", weight_norm=True):
        super().__init__()
        assert depth >= 1 and activation == "relu"
        assert output_type in ("linear", "relu", "leaky_relu")
        self.ninputs, self.noutputs = ninputs, noutputs
        self.ksize, self.width, self.depth = ksize, width, depth
        self.padding = ksize // 2 if pad else 0
        self.output_type = output_type
        self.weight_norm = weight_norm
        layers, cin = [], ninputs
        for i in range(depth):
            cout = width if i < depth - 1 else noutputs
            layers.append(_WeightNormConv(cin, cout, ksize) if weight_norm
                          else nn.Conv2d(cin, cout, ksize, padding=self.padding, bias=True))
            cin = cout
        self.layers = nn.ModuleList(layers)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        for conv in self.layers:
            if self.weight_norm:                        # as weight_norm() initialises: v = the initial weight, g = ||v||
                nn.init.xavier_uniform_(conv.weight_v, gain=gain)
                with torch.no_grad():
                    conv.weight_g.copy_(conv.weight_v.flatten(1).norm(dim=1).view(-1, 1, 1, 1))
            else:
                nn.init.xavier_uniform_(conv.weight, gain=gain)
            nn.init.zeros_(conv.bias)

    def _acts_params(self):
        acts = ["relu"] * (self.depth - 1) + [self.output_type]
        params = []
        if self.weight_norm and self.layers[0]._w is None:          # no model-level scope: this chain's layers in one launch
            ws = ops.weight_norm_multi([c.weight_g for c in self.layers], [c.weight_v for c in self.layers])
        else:
            ws = [conv.weight for conv in self.layers]
        for w, conv in zip(ws, self.layers):
            params += [w, conv.bias]
        return acts, params

    def forward(self, x):
        acts, params = self._acts_params()
        return ops.conv_chain(x, self.ksize, self.padding, acts, params)

    def forward_spp_mean(self, x, s):
        """``y = self(x); return y, y.view(B, s, ...).mean(1)`` (support/networks.py:33-36) as one node."""
        acts, params = self._acts_params()
        return ops.conv_chain_spp_mean(x, s, self.ksize, self.padding, acts, params)

    def forward_cat_broadcast(self, flat, prop, s):
        """``self(cat([flat, repeat_S(prop)], 1))`` (support/networks.py:39-42) without the fp32 concatenation."""
        acts, params = self._acts_params()
        return ops.cat_broadcast_chain(flat, prop, s, self.ksize, self.padding, acts, params)


    def forward_kernel_apply(self, x, data):
        """``kernel_apply(data, self(x))`` (one half of sbmc.KPCN.forward); data cropped to the chain's output size."""
        acts, params = self._acts_params()
        return ops.chain_kernel_apply(x, data, self.ksize, self.padding, acts, params)

    def forward_cat_upsample(self, deep, skip):
        """``self(cat([upsample2(deep), skip], 1))`` (a U-Net level's right chain) without the upsampled tensor."""
        acts, params = self._acts_params()
        return ops.cat_upsample_chain(deep, skip, self.ksize, self.padding, acts, params)


class _Level(nn.Module):
    def __init__(self, n_in, n_out, width, num_convs, ksize, output_type, next_level=None, n_up=None, weight_norm=True):
        super().__init__()
        self.is_last = next_level is None
        kw = dict(ksize=ksize, width=width, depth=num_convs, pad=True, weight_norm=weight_norm)
        if self.is_last:
            self.left = ConvChain(n_in, n_out, output_type=output_type, **kw)
        else:
            self.left = ConvChain(n_in, width, output_type="relu", **kw)
            self.next_level = next_level
            self.right = ConvChain(n_up + width, n_out, output_type=output_type, **kw)

    def forward(self, x):
        left = self.left(x)
        if self.is_last:
            return left
        # one node: the skip's and the pooled copy's gradients are summed in one pass
        skip, pooled = ops.maxpool2_skip(left)
        deeper = self.next_level(pooled)
        # cat([upsample2(deeper), skip], 1) is written once, directly as the right chain's split input, the bilinear
        # upsampling evaluated inside that kernel
        return self.right.forward_cat_upsample(deeper, skip)


class Autoencoder(nn.Module):
    def __init__(self, ninputs, noutputs, ksize=3, width=64, num_levels=3, num_convs=2, max_width=512,
                 increase_factor=1.0, output_type="linear", pooling="max", weight_norm=True):
        super().__init__()
        assert pooling == "max"
        self.num_levels = num_levels
        next_level = None
        for lvl in range(num_levels - 1, -1, -1):
            n_in = min(int(width * increase_factor ** (lvl - 1)), max_width)
            w = min(int(width * increase_factor ** lvl), max_width)
            n_up = min(int(width * increase_factor ** (lvl + 1)), max_width)
            n_out, o_type = w, "relu"
            if lvl == 0:
                n_in, n_out, o_type = ninputs, noutputs, output_type
            if lvl == num_levels - 1:
                n_up = None
            next_level = _Level(n_in, n_out, w, num_convs, ksize, o_type, next_level=next_level, n_up=n_up,
                                weight_norm=weight_norm)
        self.net = next_level

    def forward(self, x):
        div = 1 << (self.num_levels - 1)
        assert x.shape[-1] % div == 0 and x.shape[-2] % div == 0
        with weight_norm_scope(self):
            return self.net(x)


class KernelApply(nn.Module):
    """``sbmc.modules.KernelApply(softmax=True, splat=False)``; returns the tensor only."""

    def __init__(self, softmax=True, splat=False):
        super().__init__()
        assert softmax and not splat, "only the softmax gather form is on the KPCN path"

    def forward(self, data, kernels):
        return ops.kernel_apply(data, kernels)
