"""Oracle restatement of ``sbmc.KPCN`` (call sites ``train_kpcn.py:213,229``;
result keys ``support/interfaces.py:207-211``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED (``sbmc`` is
absent); geometry pinned by ``test_models.py:218-219``: 9 valid 5x5 convs take
128 -> 92 and the 21x21 apply leaves a 72 px reliable core.
"""
import torch
import torch.nn as nn

from .modules import ConvChain, KernelApply
from .utils import crop_like


class KPCN(nn.Module):
    def __init__(self, n_in, ksize=21, depth=9, width=100):
        super().__init__()
        self.ksize = ksize
        self.diffuse = ConvChain(n_in, ksize * ksize, depth=depth, width=width, ksize=5,
                                 pad=False, output_type="linear", weight_norm=False)
        self.specular = ConvChain(n_in, ksize * ksize, depth=depth, width=width, ksize=5,
                                  pad=False, output_type="linear", weight_norm=False)
        self.kernel_apply = KernelApply(softmax=True, splat=False)

    def forward(self, data):
        k_diffuse = self.diffuse(data["kpcn_diffuse_in"])
        k_specular = self.specular(data["kpcn_specular_in"])
        b_diffuse = crop_like(data["kpcn_diffuse_buffer"], k_diffuse).contiguous()
        b_specular = crop_like(data["kpcn_specular_buffer"], k_specular).contiguous()
        r_diffuse = self.kernel_apply(b_diffuse, k_diffuse)
        r_specular = self.kernel_apply(b_specular, k_specular)
        albedo = crop_like(data["kpcn_albedo"], r_diffuse)
        radiance = albedo * r_diffuse + torch.exp(r_specular) - 1
        return dict(radiance=radiance, diffuse=r_diffuse, specular=r_specular)


class SampleDenoiserStandIn(nn.Module):
    """Stand-in for the external sample-based denoisers (``sbmc.Multisteps``, ``train_sbmc.py:80-93``; layerdenoise's
    ``LayerNet``, ``train_lbmc.py:84-97``), which are absent from the reference tree: it only honours their I/O contract
    -- batch dict with per-sample ``radiance`` (B,S,3,H,W) and ``features`` (B,S,C,H,W) in, (B,3,H',W') out -- so that
    the reference's ``SBMCInterface`` / ``LBMCInterface`` can be driven for the golden fixtures: one valid 3x3
    ConvChain per sample over cat([radiance, features]), averaged over the samples."""

    def __init__(self, n_features, width=8, depth=2):
        super().__init__()
        self.net = ConvChain(3 + n_features, 3, ksize=3, width=width, depth=depth, pad=False, output_type="linear",
                             weight_norm=False)

    def forward(self, data):
        x = torch.cat([data["radiance"], data["features"]], 2)
        b, s = x.shape[:2]
        return self.net(x.flatten(0, 1)).unflatten(0, (b, s)).mean(1)
