"""Per-layer timing of the PathNet U-Net's 3x3 convolutions (forward and data gradient, pad 1).
   python3 scripts/time_unet_layers.py"""
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
n, ks = 8, 3
tot = 0.0
for (cin, cout, h, cnt) in ((64, 64, 128, 10), (128, 128, 64, 8), (256, 256, 32, 4), (192, 64, 128, 1), (384, 128, 64, 1), (64, 128, 64, 1),
                            (128, 256, 32, 1)):
    x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    xs = o.split_raw(x); wp0 = o._pack_x(w, 0); wp1 = o._pack_x(w, 1)
    tf = timeit(lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp0, b, cout, ks, 1, "relu", out_split=True, mask_out=True))
    dy = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, h, h, device=dev)))
    mask = (torch.rand(n * h * h * ((cin + 7) // 8), device=dev) * 255).to(torch.uint8)
    td = timeit(lambda: o.conv2d_x_raw(dy, (n, cout, h, h), wp1, None, cin, ks, 1, "linear", out_split=True,
                                       gate_mask=mask, gate_act="relu", colsum=True))
    print("%3d -> %3d at %3d^2 (x%d per backbone): fwd %6.1f us  dgrad %6.1f us" % (cin, cout, h, cnt, tf, td))
    tot += cnt * (tf + td)
print("weighted sum per backbone: %.3f ms" % (tot / 1e3))
