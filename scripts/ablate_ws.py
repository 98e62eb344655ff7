"""Timing-only ablations of the weight-stationary halo kernel (WCMC_DEBUG_ABLATE=<mask>, one per process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcmc_amd import ops as o
dev = "cuda"
n, cin, h, cout, ks = 8, 100, 116, 100, 5
x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
b = torch.zeros(cout, device=dev)
xs = o.split_raw(x); wp = o._pack_x(w, 0)
fn = lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp, b, cout, ks, 0, "relu", out_split=True)
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
print("ablate %-4s %7.1f us" % (os.environ.get("WCMC_DEBUG_ABLATE", "0"), e0.elapsed_time(e1) * 100))
