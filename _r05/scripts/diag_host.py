"""GPU-box diagnostic: where does the step's wall time go on the host side?"""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wcmc_amd import ops
from wcmc_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
for n in (67712, 541696):
    torch.randperm(n)
    t = time.perf_counter()
    for _ in range(10): p = torch.randperm(n)
    print("cpu randperm(%d): %.2f ms" % (n, (time.perf_counter() - t) / 10 * 1e3))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): p = torch.randperm(n, device=dev)
    torch.cuda.synchronize(); print("gpu randperm(%d): %.3f ms" % (n, (time.perf_counter() - t) / 10 * 1e3))
itf = bench.build_interface(dev, None)
batch = make_batch(8, 8, 128, seed=0, device=dev)
def step():
    itf.preprocess(batch); itf.train_batch(batch)
def timeit(label, k=6):
    for _ in range(2): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): step()
    host = (time.perf_counter() - t) / k
    torch.cuda.synchronize(); print("%-40s wall %.1f ms/step (host enqueue %.1f ms)" % (label, (time.perf_counter() - t) / k * 1e3, host * 1e3))
timeit("baseline (cpu perms)")
fm = itf.loss_funcs["l_manif"]
cache = {}
def cached(b, s, h, w):
    key = (b, s, h, w)
    if key not in cache: cache[key] = (torch.randperm(s*h*w), torch.randperm(b*s*h*w))
    return cache[key]
fm.draw_permutations = cached
timeit("cached perms (no randperm on host)")
ops.USE_SIDE_STREAM = False
timeit("cached perms, no side stream")
ops.USE_SIDE_STREAM = True
class P:
    def add(self, *a): pass
ops.set_profiler(P())
timeit("cached perms + event profiler")
ops.set_profiler(None)
ops.set_precision("fp32")
timeit("fp32 MFMA, cached perms")
