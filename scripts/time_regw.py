"""PROTOTYPE (round 4): the U-Net's 64 -> 64 3x3 layer at 128^2 x 8 images with the weights in registers (csrc/conv_regw.hip) against
the streaming halo kernel, forward, fp32 output: results and time per launch (interleaved).   python3 scripts/time_regw.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcmc_amd import ops as o
from wcmc_amd._lib import lib, check
dev = "cuda"
def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
for n, h in ((8, 128), (8, 64), (2, 40)):
    x = torch.relu(torch.randn(n, 64, h, h, device=dev))
    w = torch.randn(64, 64, 3, 3, device=dev) * 0.06
    b = torch.randn(64, device=dev) * 0.1
    xs = o.split_raw(o.to_nhwc_raw(x)); wp = o._pack_x(w, 0)
    old = lambda: o.conv2d_x_raw(xs, (n, 64, h, h), wp, b, 64, 3, 1, "relu", out_split=False)
    y = torch.empty(n, h, h, 64, device=dev)
    def new():
        check(lib().wcmc_conv3x3_regw_fwd(ctypes.c_void_p(xs.data_ptr()), n, h, h, ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                          ctypes.c_void_p(y.data_ptr()), 1, 0.01, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "regw")
    def new_nostore():
        check(lib().wcmc_conv3x3_regw_fwd(ctypes.c_void_p(xs.data_ptr()), n, h, h, ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                          ctypes.c_void_p(y.data_ptr()), 1, 7.0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "regw")
    a = old(); new(); torch.cuda.synchronize()
    yb = y.permute(0, 3, 1, 2)
    print("n=%d h=%d: max |new - old| / max|old| = %.2e  equal bitwise: %s" % (n, h, float((yb - a).abs().max() / a.abs().max()), bool(torch.equal(yb, a))))
    for rep in range(3):
        print("   streaming halo kernel %6.1f us   weights in registers %6.1f us   ... without its stores %6.1f us" % (timeit(old), timeit(new), timeit(new_nostore)))
