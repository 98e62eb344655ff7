"""Do an MFMA-bound launch (KPCN 5x5 halo igemm) and an HBM-bound one (PathNet 1x1 persistent kernel) overlap when they are
enqueued on two streams?   python3 scripts/time_overlap.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcmc_amd import ops as o
dev = "cuda"
n, c, h = 8, 100, 116
xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, c, h, h, device=dev)))
wp = o._pack_x(torch.randn(100, 100, 5, 5, device=dev) * 0.02, 0); b = torch.zeros(100, device=dev)
x1s = o.split_raw(o.to_nhwc_raw(torch.randn(64, 64, 128, 128, device=dev)))
w1p, b1 = o._pack_x(torch.randn(64, 64, 1, 1, device=dev) * 0.1, 0), torch.zeros(64, device=dev)
A = lambda: o.conv2d_x_raw(xs, (n, c, h, h), wp, b, 100, 5, 0, "relu", out_split=True)
B = lambda: o.conv2d_x_raw(x1s, (64, 64, 128, 128), w1p, b1, 64, 1, 0, "relu", out_split=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(fa, fb, reps=20):
    for _ in range(5):                                   # warm-up ON the streams (their allocator pools, too)
        if fa:
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for _ in range(reps):
        if fa:
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    cur.wait_stream(s1); cur.wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ta, tb, tab = run(A, None), run(None, B), run(A, B)
ta, tb, tab = run(A, None), run(None, B), run(A, B)
print("halo igemm alone %.1f us, pointwise alone %.1f us, both on two streams %.1f us per pair (sum %.1f, max %.1f)" % (ta, tb, tab, ta + tb, max(ta, tb)))
