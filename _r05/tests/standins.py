"""Product-side twin of ``oracle.models.SampleDenoiserStandIn`` (same parameters, HIP ops): the stand-in for the
external sample-based denoisers under the SBMC / LBMC interface tests."""
import torch
import torch.nn as nn


class SampleDenoiserStandIn(nn.Module):
    def __init__(self, n_features, width=8, depth=2):
        super().__init__()
        from wcmc_amd.modules import ConvChain
        self.net = ConvChain(3 + n_features, 3, ksize=3, width=width, depth=depth, pad=False, output_type="linear", weight_norm=False)

    def forward(self, data):
        from wcmc_amd import ops
        x = torch.cat([data["radiance"], data["features"]], 2)
        b, s = x.shape[:2]
        return ops.spp_mean(self.net(x.flatten(0, 1)), s)
