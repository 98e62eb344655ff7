"""Data-step kernels at the benchmark patch size: time, HBM rate, numpy baseline.
   python3 scripts/time_preprocess.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
from wcmc_amd.support.datasets import DenoisePreprocessor
import make_golden as mg
from oracle import datasets as od          # numpy restatement = the CPU baseline ("port")

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

pre = DenoisePreprocessor()
for h, s in ((128, 8), (512, 8)):
    raw = mg.raw_samples(h, h, s, 5)
    x = torch.from_numpy(raw).cuda()
    nsamp = h * h * s
    t1 = timeit(lambda: pre._preprocess_llpm(x))
    t2 = timeit(lambda: pre._preprocess_kpcn(x))
    rb = nsamp * 104 * 4
    b1 = rb + nsamp * 37 * 4                      # raw records are interleaved: whole 416-byte records are fetched
    b2 = rb + h * h * 44 * 4
    print("%4dx%-4d s=%d  llpm %7.1f us = %5.0f GB/s   kpcn %7.1f us = %5.0f GB/s   (raw %.1f MB)" %
          (h, h, s, t1, b1 / t1 / 1e3, t2, b2 / t2 / 1e3, rb / 1e6))
    if h == 128:
        t0 = time.perf_counter(); od.preprocess_llpm(raw); c1 = time.perf_counter() - t0
        t0 = time.perf_counter(); od.preprocess_kpcn(raw); c2 = time.perf_counter() - t0
        print("   numpy on the host (1 thread): llpm %.1f ms, kpcn %.1f ms" % (c1 * 1e3, c2 * 1e3))
