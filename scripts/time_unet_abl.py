"""Timing-only ablations of conv_halo3_bf16x3_kernel's forward instance (debug library; results are WRONG by design).
   python3 scripts/time_unet_abl.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WCMC_DEBUG_LIB", "1")
import torch
from wcmc_amd import ops as o
dev = "cuda"
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
n, ks = 8, 3
names = {0: "product", 1: "no MFMA", 2: "no weight DMA", 8: "no fragment reads", 16: "no loop barrier", 32: "no epilogue", 10: "no DMA, no reads",
         26: "no DMA/reads/barrier", 27: "empty loop", 59: "empty loop, no epilogue", 33: "no MFMA, no epilogue", 18: "no DMA, no barrier", 9: "no MFMA, no reads", 64: "no result stores", 128: "no gate mask", 192: "no stores, no mask"}
for (cin, cout, h) in ((64, 64, 128), (128, 128, 64), (256, 256, 32)):
    x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
    b = torch.randn(cout, device=dev) * 0.1
    xs = o.split_raw(x); wp0 = o._pack_x(w, 0)
    f = lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp0, b, cout, ks, 1, "relu", out_split=True, mask_out=True)
    out = []
    for rep in range(2):
        for ab in (0, 1, 32, 64, 128, 192):
            os.environ["WCMC_DEBUG_ABLATE"] = str(ab)
            t = timeit(f)
            if rep: out.append("%s %.1f" % (names[ab], t))
    os.environ["WCMC_DEBUG_ABLATE"] = "0"
    print("%3d -> %3d at %3d^2: " % (cin, cout, h) + " | ".join(out), flush=True)
