"""``support/networks.py:7-42`` of the reference: ``PathNet`` on the HIP ops.

Same constructor, attributes, ``__str__`` and I/O contract: ``forward(samples)`` takes the
batch dict and returns the per-sample P-buffer (B, S, outc, H, W) (>= 0).  The returned
tensor is a strided view of an NHWC buffer; slicing/cropping it stays free.
"""
import torch.nn as nn

from .. import ops
from ..modules import Autoencoder, ConvChain, weight_norm_scope


class PathNet(nn.Module):
    """Path embedding network"""
    single_use_parameters = True        # every parameter feeds one autograd node per step (support/interfaces.py: _defer_scope)

    def __init__(self, ic, intermc=64, outc=3, weight_norm=True):
        """weight_norm is not a reference argument: ``support/networks.py:18-24`` passes none to its chains, so upstream
        ``sbmc.modules.ConvChain``'s default applies -- weight-normalised layers (see ``modules.ConvChain``).  False selects
        plain ``nn.Conv2d`` weights (rounds 1-4 of this build; ``checkpoint.py`` restores either layout)."""
        super(PathNet, self).__init__()
        self.ic = ic
        self.intermc = intermc
        self.outc = outc
        self.final_ic = intermc + intermc
        self.embedding = ConvChain(ic, intermc, width=intermc, depth=3, ksize=1, pad=False, weight_norm=weight_norm)
        self.propagation = Autoencoder(intermc, intermc, num_levels=3, increase_factor=2.0, num_convs=3,
                                       width=intermc, ksize=3, output_type="leaky_relu", pooling="max",
                                       weight_norm=weight_norm)
        self.final = ConvChain(self.final_ic, outc, width=self.final_ic, depth=2, ksize=1, pad=False,
                               output_type="relu", weight_norm=weight_norm)

    def __str__(self):
        return "PathNet i{}in{}o{}".format(self.ic, self.intermc, self.outc)

    @staticmethod
    def _paths_nhwc(samples):
        """Both backbones read the same ``paths`` (interfaces.py:195-196): convert it once."""
        paths = samples["paths"]
        key = (paths.data_ptr(), paths._version, tuple(paths.shape))
        cached = samples.get("_wcmc_paths_nhwc")
        if cached is not None and cached[0] == key:
            return cached[1]
        bs, spp, nf, h, w = paths.shape
        flat = paths.reshape(bs * spp, nf, h, w)
        if (ops.split_path() and not paths.requires_grad and nf <= 64 and paths.is_cuda
                and bs * spp * h <= 65535):
            # the embedding chain is the only reader: transpose + split in one pass, once for both backbones
            flat = ops.presplit_shared(flat.detach())
        else:
            flat = ops.as_nhwc(flat)
        if not paths.requires_grad:
            samples["_wcmc_paths_nhwc"] = (key, flat)
        return flat

    def forward(self, samples):
        bs, spp, nf, h, w = samples["paths"].shape
        with weight_norm_scope(self):           # the 20 layers' g * v / ||v|| in one launch (and one for their gradients)
            flat, reduced = self.embedding.forward_spp_mean(self._paths_nhwc(samples), spp)   # networks.py:33-36
            propagated = self.propagation(reduced)
            out = self.final.forward_cat_broadcast(flat, propagated, spp)   # networks.py:39-42, (B*S, outc, H, W)
        return out.unflatten(0, (bs, spp))
