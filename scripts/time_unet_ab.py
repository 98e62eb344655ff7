"""A/B of the U-Net 3x3 launches: conv_halo3_bf16x3_kernel (K split over two wave groups) against the kernel it replaces.
   python3 scripts/time_unet_ab.py      (loads the debug library, which reads WCMC_HALO3 per call)"""
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
tot = {"0": 0.0, "1": 0.0}
for (cin, cout, h, cnt) in ((64, 64, 128, 10), (128, 128, 64, 8), (256, 256, 32, 4), (192, 64, 128, 1), (384, 128, 64, 1), (64, 128, 64, 1),
                            (128, 256, 32, 1)):
    x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
    b = torch.randn(cout, device=dev) * 0.1
    xs = o.split_raw(x); wp0 = o._pack_x(w, 0); wp1 = o._pack_x(w, 1); wp2 = o._pack_x(w, 2)
    dy = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, h, h, device=dev)))
    mask = (torch.rand(n * h * h * ((cin + 7) // 8), device=dev) * 255).to(torch.uint8)
    f = lambda: o.conv2d_x_raw(xs, (n, cin, h, h), wp0, b, cout, ks, 1, "relu", out_split=True, mask_out=True)
    d2 = lambda: o.conv2d_x_raw(dy, (n, cout, h, h), wp2, None, cin, ks, 1, "linear", out_split=True, gate_mask=mask, gate_act="relu",
                                colsum=True, terms=2)
    d3 = lambda: o.conv2d_x_raw(dy, (n, cout, h, h), wp1, None, cin, ks, 1, "linear", out_split=True, gate_mask=mask, gate_act="relu",
                                colsum=True)
    res = {}
    for sw in ("0", "1"):
        os.environ["WCMC_HALO3"] = sw
        yf, mf = f(); y2, c2 = d2(); y3, c3 = d3()
        G = c2.numel() // ((cin + 15) // 16 * 16) if False else None
        res[sw] = dict(tf=timeit(f), t2=timeit(d2), t3=timeit(d3), yf=o.unsplit_debug(yf, n, cout, h, h).clone(), mf=mf.clone(),
                       y2=o.unsplit_debug(y2, n, cin, h, h).clone(), y3=o.unsplit_debug(y3, n, cin, h, h).clone(),
                       c2=o.colsum_finish_raw(c2, (n, cin, h, h)).clone() if hasattr(o, "colsum_finish_raw") else None)
        tot[sw] += cnt * (res[sw]["tf"] + res[sw]["t2"])
    rel = lambda a, b: ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
    a, b_ = res["0"], res["1"]
    extra = ""
    if a["c2"] is not None:
        extra = " colsum %.1e" % rel(a["c2"], b_["c2"])
    print("%3d -> %3d at %3d^2 (x%d): fwd %6.1f -> %6.1f us   dgrad(2) %6.1f -> %6.1f us   dgrad(3) %6.1f -> %6.1f   | rel diff fwd %.1e mask %.1e d2 %.1e d3 %.1e%s" % (
        cin, cout, h, cnt, a["tf"], b_["tf"], a["t2"], b_["t2"], a["t3"], b_["t3"], rel(a["yf"], b_["yf"]),
        (a["mf"] != b_["mf"]).float().mean().item(), rel(a["y2"], b_["y2"]), rel(a["y3"], b_["y3"]), extra), flush=True)
print("weighted sum per backbone (fwd + two-term dgrad): %.3f -> %.3f ms" % (tot["0"] / 1e3, tot["1"] / 1e3))
