"""GEMM-only (phase 1) and finish (phase 2) timings of the 5x5 weight gradients at the KPCN layer sizes.
   python3 scripts/time_wgrad.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
TERMS = int(os.environ.get("WCMC_WGRAD_TERMS", "1"))      # bf16 MFMAs per product of the weight gradient: 1 (hi planes only) or 3
from wcmc_amd import ops as o
from wcmc_amd.ops import _ptr, _stream, lib, check
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def case(n, cin, h, cout, ks):
    ho = h - ks + 1
    xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev)))
    dys = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, ho, ho, device=dev)))
    nbytes = lib().wcmc_conv2d_wgrad_bf16x3_workspace_bytes(n, ho, ho, cout, cin, ks)
    ws = torch.empty((nbytes + 3) // 4, device=dev); dw = torch.empty(cout, cin, ks, ks, device=dev); db = torch.empty(cout, device=dev)
    args = (_ptr(xs), n, h, h, cin, _ptr(dys), cout, ks, 0, _ptr(dw), _ptr(db), _ptr(ws), ws.numel() * 4)
    t1 = timeit(lambda: check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 1, None, TERMS, _stream()), "wgrad"))
    t2 = timeit(lambda: check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 2, None, TERMS, _stream()), "wgrad"))
    fl = 2.0 * n * ho * ho * cout * cin * ks * ks
    print("%3d->%3d out %3d: gemm %7.1f us (%6.1f TF/s)  finish %6.1f us  workspace %.1f MB" % (cin, cout, ho, t1, fl / t1 / 1e6, t2, nbytes / 1e6))
for h in (128, 124, 120, 116, 112, 108, 104):
    case(8, 100, h, 100, 5)
case(8, 100, 96, 441, 5)
case(8, 39, 128, 100, 5)
