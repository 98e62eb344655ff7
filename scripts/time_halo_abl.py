"""Timing-only ablation of the KPCN 5x5 halo igemm (debug library): WCMC_DEBUG_ABLATE=2 drops the weight DMA of the stage loop.
   WCMC_DEBUG_LIB=1 python3 scripts/time_halo_abl.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcmc_amd import ops as o
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
n, cin, h, cout, ks = 8, 100, 116, 100, 5
x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
b = torch.zeros(cout, device=dev)
xs = o.split_raw(x); wp = o._pack_x(w, 0)
t = timeit(lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp, b, cout, ks, 0, "relu", out_split=True))
print("WCMC_DEBUG_ABLATE=%s: %.1f us" % (os.environ.get("WCMC_DEBUG_ABLATE", "0"), t))
