"""Timing-only ablations of the KPCN 5x5 halo igemm (debug library), interleaved inside ONE process so that clock / box
drift cancels:  make -C wcmc_amd/csrc debug; WCMC_DEBUG_LIB=1 python3 scripts/time_halo_abl.py [h] [mode ...]
Modes (sums combine where an instance exists): 1 no MFMA, 2 no weight DMA in the stage loop, 4 one halo per tile (no slab
reloads), 8 no fragment reads, 16 no stage barrier, 32 no epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wcmc_amd import ops as o
dev = "cuda"
h = int(sys.argv[1]) if len(sys.argv) > 1 else 116
modes = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 4, 8, 16, 32, 10, 26]
n, cin, cout, ks = 8, 100, 100, 5
x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
b = torch.zeros(cout, device=dev)
xs = o.split_raw(x); wp = o._pack_x(w, 0)
fn = lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp, b, cout, ks, 0, "relu", out_split=True)
def once(mode, reps=10):
    os.environ["WCMC_DEBUG_ABLATE"] = str(mode)
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for m in modes: once(m)
res = {m: [] for m in modes}
for rnd in range(8):
    for m in modes:
        res[m].append(once(m))
for m in modes:
    a = np.array(res[m])
    print("h=%d WCMC_DEBUG_ABLATE=%-3d median %.1f us  (min %.1f max %.1f)" % (h, m, np.median(a), a.min(), a.max()))
