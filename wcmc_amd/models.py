"""MI355X-native ``sbmc.KPCN`` (constructed at ``train_kpcn.py:213,229``; result keys consumed
at ``support/interfaces.py:207-211``).  Same specification as ``oracle/models.py``."""
import torch.nn as nn

from . import ops
from .modules import ConvChain, KernelApply
from .support.utils import crop_like


class KPCN(nn.Module):
    def __init__(self, n_in, ksize=21, depth=9, width=100):
        super().__init__()
        self.ksize = ksize
        self.diffuse = ConvChain(n_in, ksize * ksize, depth=depth, width=width, ksize=5, pad=False,
                                 output_type="linear")
        self.specular = ConvChain(n_in, ksize * ksize, depth=depth, width=width, ksize=5, pad=False,
                                  output_type="linear")
        self.kernel_apply = KernelApply(softmax=True, splat=False)

    def forward(self, data):
        with ops.on_branch(data["kpcn_specular_in"].device) as br:      # specular half on the branch stream
            k_specular = self.specular(data["kpcn_specular_in"])
            b_specular = crop_like(data["kpcn_specular_buffer"], k_specular)
            r_specular = self.kernel_apply(b_specular, k_specular)
        k_diffuse = self.diffuse(data["kpcn_diffuse_in"])
        b_diffuse = crop_like(data["kpcn_diffuse_buffer"], k_diffuse)
        r_diffuse = self.kernel_apply(b_diffuse, k_diffuse)
        br.join(r_specular)
        albedo = crop_like(data["kpcn_albedo"], r_diffuse)
        radiance = ops.recombine(albedo, r_diffuse, r_specular)
        return dict(radiance=radiance, diffuse=r_diffuse, specular=r_specular)
