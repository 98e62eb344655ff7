"""MI355X-native ``sbmc.KPCN`` (constructed at ``train_kpcn.py:213,229``; result keys consumed
at ``support/interfaces.py:207-211``).  Same specification as ``oracle/models.py``."""
import types

import torch.nn as nn

from . import ops
from .modules import ConvChain, KernelApply
from .support.utils import crop_like


class KPCN(nn.Module):
    single_use_parameters = True        # every parameter feeds one autograd node per step (support/interfaces.py: _defer_scope)

    def __init__(self, n_in, ksize=21, depth=9, width=100):
        super().__init__()
        self.ksize = ksize
        self.diffuse = ConvChain(n_in, ksize * ksize, depth=depth, width=width, ksize=5, pad=False,
                                 output_type="linear", weight_norm=False)
        self.specular = ConvChain(n_in, ksize * ksize, depth=depth, width=width, ksize=5, pad=False,
                                  output_type="linear", weight_norm=False)
        self.kernel_apply = KernelApply(softmax=True, splat=False)

    @staticmethod
    def _branch(chain, x, buffer):
        """kernel_apply(crop_like(buffer, k), k) with k = chain(x): chain and apply are one autograd node on the
        split-bf16 path (the kernel gradient goes from the apply to the chain without an fp32 round trip)."""
        shrink = chain.depth * (chain.ksize - 1 - 2 * chain.padding)
        k_like = types.SimpleNamespace(shape=tuple(x.shape[:2]) + (x.shape[2] - shrink, x.shape[3] - shrink))
        return chain.forward_kernel_apply(x, crop_like(buffer, k_like))

    def forward(self, data):
        with ops.on_branch(data["kpcn_specular_in"].device) as br:      # specular half on the branch stream
            r_specular = self._branch(self.specular, data["kpcn_specular_in"], data["kpcn_specular_buffer"])
        r_diffuse = self._branch(self.diffuse, data["kpcn_diffuse_in"], data["kpcn_diffuse_buffer"])
        br.join(r_specular)
        albedo = crop_like(data["kpcn_albedo"], r_diffuse)
        radiance = ops.recombine(albedo, r_diffuse, r_specular)
        return dict(radiance=radiance, diffuse=r_diffuse, specular=r_specular)
