"""GEMM phase of the U-Net 3x3 weight gradients.   python3 scripts/time_wgrad_unet.py"""
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
tot = 0.0
for (cin, cout, h, cnt) in ((64, 64, 128, 5), (128, 128, 64, 4), (256, 256, 32, 2), (192, 64, 128, 1), (384, 128, 64, 1), (64, 128, 64, 1),
                            (128, 256, 32, 1)):
    n, ks = 8, 3
    xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev)))
    dys = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, h, h, device=dev)))
    nbytes = lib().wcmc_conv2d_wgrad_bf16x3_workspace_bytes(n, h, h, cout, cin, ks)
    ws = torch.empty((nbytes + 3) // 4, device=dev); dw = torch.empty(cout, cin, ks, ks, device=dev); db = torch.empty(cout, device=dev)
    args = (_ptr(xs), n, h, h, cin, _ptr(dys), cout, ks, 1, _ptr(dw), _ptr(db), _ptr(ws), ws.numel() * 4)
    t1 = timeit(lambda: check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 1, None, TERMS, _stream()), "wgrad"))
    t0 = timeit(lambda: check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 0, None, TERMS, _stream()), "wgrad"))
    print("%3d -> %3d at %3d^2 (x%d per backbone): gemm %6.1f us, with reduce + bias gradient %6.1f us, workspace %.1f MB" % (cin, cout, h, cnt, t1, t0, nbytes / 1e6))
    tot += cnt * t1
print("weighted gemm sum per backbone: %.3f ms" % (tot / 1e3))
