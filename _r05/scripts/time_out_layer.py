"""The KPCN output layer (100 -> 441, 5x5, 96^2 -> 92^2, 8 patches) forward with 3 / 2 / 1 bf16 MFMAs per product: time per launch
(interleaved) and the result against fp64 on the operands each rung multiplies (profiles/r04_forward_ladder.txt, table "last").
   python3 scripts/time_out_layer.py"""
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
torch.manual_seed(0)
n, ks, cin, cout, h = 8, 5, 100, 441, 96
xf = torch.relu(torch.randn(n, cin, h, h, device=dev))
x = o.to_nhwc_raw(xf)
w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
b = torch.randn(cout, device=dev) * 0.1
xs = o.split_raw(x)
packs = {3: o._pack_x(w, 0), 2: o._pack_x(w, 3), 1: o._pack_x(w, 3), "h": o._pack_x(w, 4)}
run = lambda t: (o.conv2d_out_f16_raw(xs, (n, cin, h, h), packs[t], b, cout, ks, 0) if t == "h" else
                 o.conv2d_x_raw(xs, (n, cin, h, h), packs[t], b, cout, ks, 0, "linear", out_split=False, terms=t))
bf = lambda t: t.bfloat16().double()
hf = lambda t: t.half().double()
ref = {3: torch.nn.functional.conv2d(xf[:1].double().cpu(), w.double().cpu(), b.double().cpu()),
       2: torch.nn.functional.conv2d(bf(xf[:1]).cpu(), w.double().cpu(), b.double().cpu()),
       1: torch.nn.functional.conv2d(bf(xf[:1]).cpu(), bf(w).cpu(), b.double().cpu()),
       "h": torch.nn.functional.conv2d(hf(xf[:1]).cpu(), hf(w).cpu(), b.double().cpu())}
for t in (3, 2, 1, "h"):
    y = run(t)[:1].double().cpu()
    print("terms %s: max|y - fp64(operands as multiplied)| / max|y| = %.2e   vs exact fp64: %.2e" %
          (t, float((y - ref[t]).abs().max() / ref[t].abs().max()), float((y - ref[3]).abs().max() / ref[3].abs().max())))
for rep in range(3):
    print("  ".join("terms %s: %6.1f us" % (t, timeit(lambda: run(t))) for t in (3, 2, 1, "h")) + "   (h = fp16, incl. the split -> fp16 conversion of x)")
