"""The PathNet 1x1 weight gradients: the filter-row kernel (KS = 1 instances, LDS-DMA stage ring) against the one-tap kernel
(register staging), GEMM phase only, and the bytes they stream.   python3 scripts/time_wgrad_1x1.py"""
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
def case(n, cin, h, cout):
    xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev)))
    dys = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, h, h, device=dev)))
    out = []
    res = {}
    for mode in ("0", "1"):
        os.environ["WCMC_WGRAD_ROWS_1X1"] = mode
        nbytes = lib().wcmc_conv2d_wgrad_bf16x3_workspace_bytes(n, h, h, cout, cin, 1)
        ws = torch.empty((nbytes + 3) // 4, device=dev); dw = torch.empty(cout, cin, 1, 1, device=dev); db = torch.empty(cout, device=dev)
        args = (_ptr(xs), n, h, h, cin, _ptr(dys), cout, 1, 0, _ptr(dw), _ptr(db), _ptr(ws), ws.numel() * 4)
        t1 = timeit(lambda: check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 1, None, TERMS, _stream()), "wgrad"))
        t2 = timeit(lambda: check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 2, None, TERMS, _stream()), "wgrad"))
        check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 0, None, TERMS, _stream()), "wgrad")
        res[mode] = dw.clone()
        out.append((t1, t2, nbytes))
    gb = 4.0 * n * h * h * ((cin + 7) // 8 * 8 + (cout + 7) // 8 * 8) / 1e9
    same = torch.equal(res["0"], res["1"])
    print("%3d->%3d on %d x %d^2: one-tap %6.1f us (%.2f TB/s) + finish %5.1f | rows %6.1f us (%.2f TB/s) + finish %5.1f   dw bit-identical: %s, max |diff| %.2e" %
          (cin, cout, n, h, out[0][0], gb / out[0][0] * 1e3, out[0][1], out[1][0], gb / out[1][0] * 1e3, out[1][1], same,
           (res["0"] - res["1"]).abs().max().item()))
case(64, 64, 128, 64)
case(64, 128, 128, 128)
case(64, 36, 128, 64)
case(64, 128, 128, 3)
