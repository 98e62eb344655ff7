"""Per-layer timing of the KPCN branch's 5x5 convolutions (forward and data gradient) as the train step launches them:
   python3 scripts/time_conv_layers.py            -> one line per layer and the sum over one branch (x2 per step)"""
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
n, ks = 8, 5
chans = [39] + [100] * 8 + [441]
tot_f = tot_d = 0.0
h = 128
for li in range(9):
    cin, cout = chans[li], chans[li + 1]
    ho = h - ks + 1
    x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    xs = o.split_raw(x); wp0 = o._pack_x(w, 0); wp1 = o._pack_x(w, 1)
    last = li == 8
    if last:
        tf = timeit(lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp0, b, cout, ks, 0, "linear", out_split=False))
    else:
        tf = timeit(lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp0, b, cout, ks, 0, "relu", out_split=True, mask_out=True))
    td = 0.0
    if li > 0:                                     # data gradient: full correlation with the flipped weights, gated by the
        dy = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, ho, ho, device=dev)))   # input's (hi > 0) bit mask
        mask = (torch.rand(n * h * h * ((cin + 7) // 8), device=dev) * 255).to(torch.uint8)
        td = timeit(lambda: o.conv2d_x_raw(dy, (n, cout, ho, ho), wp1, None, cin, ks, ks - 1, "linear", out_split=True,
                                           gate_mask=mask, gate_act="relu", colsum=True))
    print("layer %d  %3d -> %3d  in %3d: fwd %6.1f us  dgrad %6.1f us" % (li, cin, cout, h, tf, td))
    tot_f += tf; tot_d += td
    h = ho
print("one branch: fwd %.1f us + dgrad %.1f us = %.3f ms" % (tot_f, tot_d, (tot_f + tot_d) / 1e3))
