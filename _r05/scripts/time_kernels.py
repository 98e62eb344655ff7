"""Event-timed micro-bench of the conv kernels at benchmark shapes (isolated launches).
   python3 scripts/time_kernels.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcmc_amd import ops as o
dev = "cuda"
torch.manual_seed(0)
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def conv_case(name, n, cin, h, cout, ks, pad):
    x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    ho = h + 2 * pad - ks + 1
    dy = o.to_nhwc_raw(torch.randn(n, cout, ho, ho, device=dev))
    xs, dys = o.split_raw(x), o.split_raw(dy)
    wp, wpt = o._pack_x(w, 0), o._pack_x(w, 1)
    fl = 2.0 * n * ho * ho * cout * cin * ks * ks
    t1 = timeit(lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp, b, cout, ks, pad, "relu", out_split=True))
    t2 = timeit(lambda: o.conv2d_x_raw(dys, (n, cout, ho, ho), wpt, None, cin, ks, ks - 1 - pad, "linear", out_split=True, gate=xs, gate_act="relu"))
    t3 = timeit(lambda: o.conv2d_wgrad_x_raw(xs, (n, cin, h, h), dys, cout, ks, pad, (cout, cin, ks, ks)))
    print("%-34s fwd %7.1f us (%6.1f TF/s)  dgrad %7.1f us (%6.1f)  wgrad %7.1f us (%6.1f)" %
          (name, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6))
conv_case("kpcn 100->100 5x5 @116 B8", 8, 100, 116, 100, 5, 0)
conv_case("kpcn 100->100 5x5 @100 B8", 8, 100, 100, 100, 5, 0)
conv_case("kpcn 39->100 5x5 @128 B8", 8, 39, 128, 100, 5, 0)
conv_case("kpcn 100->441 5x5 @96 B8", 8, 100, 96, 441, 5, 0)
conv_case("unet 64->64 3x3 @128 B8", 8, 64, 128, 64, 3, 1)
conv_case("unet 128->128 3x3 @64 B8", 8, 128, 64, 128, 3, 1)
conv_case("unet 256->256 3x3 @32 B8", 8, 256, 32, 256, 3, 1)
conv_case("unet 384->128 3x3 @64 B8", 8, 384, 64, 128, 3, 1)
conv_case("emb 36->64 1x1 @128 B64", 64, 36, 128, 64, 1, 0)
conv_case("emb 64->64 1x1 @128 B64", 64, 64, 128, 64, 1, 0)
conv_case("final 128->128 1x1 @128 B64", 64, 128, 128, 128, 1, 0)
conv_case("final 128->3 1x1 @128 B64", 64, 128, 128, 3, 1, 0)
