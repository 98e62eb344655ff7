"""Timing-only ablations of the filter-row weight-gradient kernel (debug library), interleaved inside one process:
   make -C wcmc_amd/csrc debug; WCMC_DEBUG_LIB=1 python3 scripts/time_wgrad_abl.py [h] [mode ...]
Modes: 1 no MFMA, 2 no stage fills after the first, 3 both, 8 no wait for the fragment reads; the eight-wave kernel (default;
WCMC_WGRAD_ROWS8=0: the seven-wave one) also has 32 no fragment reads, 34 = 32 + 2 (MFMAs and barriers only), 35 = all three."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
TERMS = int(os.environ.get("WCMC_WGRAD_TERMS", "1"))      # bf16 MFMAs per product of the weight gradient: 1 (hi planes only) or 3
from wcmc_amd import ops as o
from wcmc_amd.ops import _ptr, _stream, lib, check
dev = "cuda"
h = int(sys.argv[1]) if len(sys.argv) > 1 else 124
modes = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 3, 8]
n, cin, cout, ks = 8, 100, 100, 5
ho = h - ks + 1
xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev)))
dys = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, ho, ho, device=dev)))
nbytes = lib().wcmc_conv2d_wgrad_bf16x3_workspace_bytes(n, ho, ho, cout, cin, ks)
ws = torch.empty((nbytes + 3) // 4, device=dev); dw = torch.empty(cout, cin, ks, ks, device=dev); db = torch.empty(cout, device=dev)
args = (_ptr(xs), n, h, h, cin, _ptr(dys), cout, ks, 0, _ptr(dw), _ptr(db), _ptr(ws), ws.numel() * 4)
fn = lambda: check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 1, None, TERMS, _stream()), "wgrad")
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
