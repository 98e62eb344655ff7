"""wcmc_amd -- the KPCN-Manifold training hot path of Mephisto405/WCMC, built MI355X-first.

``wcmc_amd.support.{interfaces,networks,losses,utils}`` mirror the reference's ``support``
package for this path; ``wcmc_amd.models.KPCN`` / ``wcmc_amd.modules`` stand in for the
external ``sbmc`` package.  All arithmetic runs in ``libwcmc_hip.so`` (``include/wcmc_hip.h``).
"""
from . import ops  # noqa: F401
from .models import KPCN  # noqa: F401
from .modules import Autoencoder, ConvChain, KernelApply  # noqa: F401
