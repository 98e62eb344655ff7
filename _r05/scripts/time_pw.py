"""A/B timing of the 1x1 (PathNet) GEMM launches: persistent pointwise kernel vs the tiled streaming kernel.
   python3 scripts/time_pw.py            (same box, same process: WCMC_IGEMM_PW is read per call)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcmc_amd import ops as o

dev = torch.device("cuda", 0)
n, h = 64, 128
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


rows = []
for cin, cout, kind in [(36, 64, "fwd"), (64, 64, "fwd"), (64, 64, "fwd32"), (64, 64, "dgrad"), (128, 128, "fwd"),
                        (128, 128, "dgrad32"), (3, 128, "dgrad"), (128, 3, "fwd32")]:
    xs = o.split_raw(o.to_nhwc_raw(rnd(n, cin, h, h)))
    w = rnd(cout, cin, 1, 1) * 0.2
    b = rnd(cout) * 0.1
    wp = o._pack_x(w, 0)
    mask = None
    if kind == "dgrad":
        _, mask = o.conv2d_x_raw(xs, (n, cin, h, h), wp, b, cout, 1, 0, "relu", out_split=True, mask_out=True)
    if kind == "fwd":
        fn = lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp, b, cout, 1, 0, "relu", out_split=True, mask_out=True)

    elif kind == "fwd32":
        fn = lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp, b, cout, 1, 0, "relu", out_split=False)
    elif kind == "dgrad32":
        fn = lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp, None, cout, 1, 0, "linear", out_split=False)
    else:
        fn = lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp, None, cout, 1, 0, "linear", out_split=True, gate_act="relu",
                                    gate_mask=mask, colsum=True)
    cpi = (cin + 7) // 8 * 8
    byts = n * h * h * (4 * cpi + 4 * max(cout, 4)) / 1e6
    t = {}
    for flag in ("0", "1"):
        os.environ["WCMC_IGEMM_PW"] = flag
        t[flag] = timed(fn)
    print("%4d -> %-4d %-8s %7.1f MB   tiled %7.1f us (%.2f TB/s)   pointwise %7.1f us (%.2f TB/s)" %
          (cin, cout, kind, byts, t["0"], byts / t["0"], t["1"], byts / t["1"]), flush=True)
    del xs, mask
