"""Oracle restatement of ``support/networks.py:7-42`` (``PathNet``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  The glue (view / mean / cat /
repeat order) follows the reference file line by line in meaning; the
ConvChain / Autoencoder it is built from are the unpinned ``sbmc`` restatements
in ``oracle/modules.py``.
"""
import torch
import torch.nn as nn

from .modules import Autoencoder, ConvChain


class PathNet(nn.Module):
    def __init__(self, ic, intermc=64, outc=3, weight_norm=True):
        super().__init__()
        self.ic, self.intermc, self.outc = ic, intermc, outc
        self.final_ic = intermc + intermc
        # networks.py:18-19
        self.embedding = ConvChain(ic, intermc, width=intermc, depth=3, ksize=1, pad=False, weight_norm=weight_norm)
        # networks.py:20-22
        self.propagation = Autoencoder(intermc, intermc, num_levels=3, increase_factor=2.0,
                                       num_convs=3, width=intermc, ksize=3,
                                       output_type="leaky_relu", pooling="max", weight_norm=weight_norm)
        # networks.py:23-24
        self.final = ConvChain(self.final_ic, outc, width=self.final_ic, depth=2, ksize=1,
                               pad=False, output_type="relu", weight_norm=weight_norm)

    def __str__(self):
        return "PathNet i{}in{}o{}".format(self.ic, self.intermc, self.outc)

    def forward(self, samples):
        paths = samples["paths"]                                   # networks.py:30
        bs, spp, nf, h, w = paths.shape
        flat = paths.contiguous().view(bs * spp, nf, h, w)
        flat = self.embedding(flat).view(bs, spp, self.intermc, h, w)
        reduced = flat.mean(1)                                     # networks.py:36
        propagated = self.propagation(reduced)
        rep = propagated.unsqueeze(1).repeat(1, spp, 1, 1, 1).view(bs * spp, self.intermc, h, w)
        flat = torch.cat([flat.view(bs * spp, self.intermc, h, w), rep], 1)
        return self.final(flat).view(bs, spp, self.outc, h, w)     # networks.py:41
