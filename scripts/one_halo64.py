"""One KPCN 5x5 forward launch (16x16 tiles) and one data-gradient launch, a few times -- for instruction-count PMC passes of the debug
library's ablation instances (WCMC_DEBUG_ABLATE).   WCMC_DEBUG_LIB=1 WCMC_DEBUG_ABLATE=32 python3 scripts/one_halo64.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WCMC_DEBUG_LIB", "1")
import torch
from wcmc_amd import ops as o
dev = "cuda"
torch.manual_seed(0)
n, c, h = 8, 100, 116
xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, c, h, h, device=dev)))
w = torch.randn(100, 100, 5, 5, device=dev) * 0.02
b = torch.zeros(100, device=dev)
wp = o._pack_x(w, 0)
for _ in range(3):
    o.conv2d_x_raw(xs, (n, c, h, h), wp, b, 100, 5, 0, "relu", out_split=True, mask_out=True)
torch.cuda.synchronize()
print("done")
