"""Oracle restatement of ``sbmc.modules`` pieces used by the hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED: the
``sbmc`` package is not part of ``/root/reference``; constructor arguments are
pinned by ``support/networks.py:18-24`` and the geometry by
``test_models.py:218-219`` (128 -> 92 -> 72), everything else is the published
adobe/sbmc algorithm restated as this build's specification (SURVEY.md
Appendix A).

Frozen choices (identical in the HIP path, ``wcmc_amd/modules.py``):
  * ConvChain: ``depth-1`` x [Conv2d(k, padding=k//2 if pad else 0) + ReLU], then
    one Conv2d to ``noutputs`` followed by ``output_type`` activation
    (linear | relu | leaky_relu with slope 0.01).  Bias everywhere, no batch norm.
  * Autoencoder: U-Net; level widths ``width * increase_factor**lvl``; each
    level = left ConvChain -> MaxPool2d(2) -> next level -> x2 bilinear upsample
    (align_corners=False) -> cat([up, left], 1) -> right ConvChain; the deepest
    level is a left chain only.
  * KernelApply(softmax=True, splat=False): softmax over the k*k taps, tap
    index t = (dy+r)*k + (dx+r) (row-major dy, dx), zero-extended gather
    ``out[b,c,y,x] = sum_t w[b,t,y,x] * data0[b,c,y+dy,x+dx]`` with no
    renormalisation at the border.
  * Weight normalisation: ``ConvChain(weight_norm=True)`` is the default, as in published adobe/sbmc; ``sbmc.KPCN`` passes
    False, PathNet's call sites pass nothing (``support/networks.py:18-24``) and therefore train ``g * v / ||v||``.
  * Parameter names: ``layers.<i>.weight_g`` / ``.weight_v`` / ``.bias`` inside a weight-normalised chain
    (``torch.nn.utils.weight_norm``'s), ``layers.<i>.weight`` / ``.bias`` with ``weight_norm=False``.
  * U-Net concatenation order: ``cat([upsampled deeper level, left skip], 1)``.
  * Init: xavier-uniform with ReLU gain, zero bias.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

LEAKY_SLOPE = 0.01
# Test hook, mirror of wcmc_amd.ops.DEBUG_ACTS: post-activation outputs of every non-linear layer.
DEBUG_ACTS = None


def _activation(x, kind):
    if kind == "linear":
        return x
    if kind == "relu":
        return F.relu(x)
    if kind == "leaky_relu":
        return F.leaky_relu(x, LEAKY_SLOPE)
    raise ValueError("unknown output_type %r" % (kind,))


class ConvChain(nn.Module):
    """``sbmc.modules.ConvChain`` (call sites ``support/networks.py:18-19,23-24``)."""

    def __init__(self, ninputs, noutputs, ksize=3, width=64, depth=3, pad=True,
                 activation="relu", output_type="linear", weight_norm=True):
        """weight_norm: default True, as published adobe/sbmc's ConvChain (unverifiable here, the package is absent; three
        independent readings agree: this build's, SURVEY.md Appendix A's, the round-4 review's) -- ``sbmc.KPCN`` overrides it
        to False, PathNet's calls (``support/networks.py:18-24``) do not.  With True each layer is
        ``torch.nn.utils.weight_norm(nn.Conv2d(...))``: parameters ``weight_g`` / ``weight_v``,
        ``weight = g * v / ||v||`` (norm per output channel)."""
        super().__init__()
        assert depth >= 1 and activation == "relu"
        self.ninputs, self.noutputs = ninputs, noutputs
        self.ksize, self.width, self.depth = ksize, width, depth
        self.padding = ksize // 2 if pad else 0
        self.output_type = output_type
        self.weight_norm = weight_norm
        layers = []
        cin = ninputs
        for i in range(depth):
            cout = width if i < depth - 1 else noutputs
            layers.append(nn.Conv2d(cin, cout, ksize, padding=self.padding, bias=True))
            cin = cout
        self.layers = nn.ModuleList(layers)
        self.reset_parameters()
        if weight_norm:
            import warnings
            with warnings.catch_warnings():               # (the legacy API is what sbmc-era checkpoints were written with)
                warnings.simplefilter("ignore", FutureWarning)
                for conv in self.layers:
                    nn.utils.weight_norm(conv)

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        for conv in self.layers:
            nn.init.xavier_uniform_(conv.weight, gain=gain)
            nn.init.zeros_(conv.bias)

    def forward(self, x):
        for i, conv in enumerate(self.layers):
            x = conv(x)
            kind = "relu" if i < self.depth - 1 else self.output_type
            x = _activation(x, kind)
            if DEBUG_ACTS is not None and kind != "linear":
                DEBUG_ACTS.append(x.detach())
        return x


class _Level(nn.Module):
    def __init__(self, n_in, n_out, width, num_convs, ksize, output_type,
                 next_level=None, n_up=None, weight_norm=True):
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
        down = F.max_pool2d(left, 2, 2)
        deeper = self.next_level(down)
        up = F.interpolate(deeper, scale_factor=2, mode="bilinear", align_corners=False)
        return self.right(torch.cat([up, left], 1))


class Autoencoder(nn.Module):
    """``sbmc.modules.Autoencoder`` (call site ``support/networks.py:20-22``)."""

    def __init__(self, ninputs, noutputs, ksize=3, width=64, num_levels=3, num_convs=2,
                 max_width=512, increase_factor=1.0, output_type="linear", pooling="max", weight_norm=True):
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
            next_level = _Level(n_in, n_out, w, num_convs, ksize, o_type,
                                next_level=next_level, n_up=n_up, weight_norm=weight_norm)
        self.net = next_level

    def forward(self, x):
        assert x.shape[-1] % (1 << (self.num_levels - 1)) == 0
        assert x.shape[-2] % (1 << (self.num_levels - 1)) == 0
        return self.net(x)


def kernel_apply(data, logits, softmax=True):
    """Gather-form predicted-kernel apply; see module docstring for semantics.

    data (B,C,h,w), logits (B,k*k,h,w) -> (B,C,h,w).
    """
    b, k2, h, w = logits.shape
    k = int(round(math.sqrt(k2)))
    assert k * k == k2 and k % 2 == 1
    r = k // 2
    wts = F.softmax(logits, dim=1) if softmax else logits
    c = data.shape[1]
    # unfold gives (B, C*k*k, h*w) with the k*k taps row-major (dy, dx): exactly t.
    patches = F.unfold(data, kernel_size=k, padding=r).view(b, c, k2, h, w)
    return (patches * wts.unsqueeze(1)).sum(2)


class KernelApply(nn.Module):
    """``sbmc.modules.KernelApply(softmax=True, splat=False)``; returns the tensor only."""

    def __init__(self, softmax=True, splat=False):
        super().__init__()
        assert not splat
        self.softmax = softmax

    def forward(self, data, kernels):
        return kernel_apply(data, kernels, self.softmax)
