"""Where the loader-fed step loses its 4-5 %: the graphed step on a resident batch with and without the loader running beside
it, split into GPU time of the replay (events around it), host time inside graph.replay(), and the wall time of a call.
   python3 scripts/diag_loader_gap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import make_golden as mg
import bench
from wcmc_amd.support.loader import PatchLoader
from wcmc_amd.graph import GraphedTrainStep
dev = torch.device("cuda", 0)
H = W = 512; S = 8
images = [{"raw": mg.raw_samples(H, W, S, 10 + i), "gt": np.random.rand(H, W, 9).astype(np.float32), "prob": None} for i in range(2)]
loader = PatchLoader(lambda i: images[i % 2], range(6), dev, batch_size=8, patch_size=128, workers=2)
for _ in loader: pass
itf = bench.build_interface(dev, None, rng="device")
first = next(iter(loader))
step = GraphedTrainStep(itf, first)
step.after_enqueue = loader.kick
for b in loader: step(b)
g = step.graph
stat = {}
class Timed:
    def __init__(self, inner): self.inner = inner
    def replay(self):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t0 = time.perf_counter(); self.inner.replay(); t1 = time.perf_counter(); e1.record()
        stat.setdefault("ev", []).append((e0, e1)); stat.setdefault("host_replay", []).append(t1 - t0)
    def __getattr__(self, k): return getattr(self.inner, k)
step.graph = Timed(g)
def run(feed, label):
    stat.clear(); calls = []
    torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
    it = iter(loader) if feed != "none" else iter(range(96))
    pops = []
    while True:
        p0 = time.perf_counter(); b = next(it, None); pops.append(time.perf_counter() - p0)
        if b is None:
            break
        c0 = time.perf_counter(); step(b if feed == "fed" else first); calls.append(time.perf_counter() - c0); nb += 1
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / nb
    top = sorted(pops, reverse=True)
    print("    next(loader): mean %.3f ms, median %.3f, the six largest %s ms = %.0f %% of the total" %
          (sum(pops) / len(pops) * 1e3, top[len(top) // 2] * 1e3, " ".join("%.1f" % (x * 1e3) for x in top[:6]), 100 * sum(top[:6]) / sum(pops)))
    gpu = [a.elapsed_time(b) for a, b in stat["ev"]][5:]
    hr = stat["host_replay"][5:]
    print("%-34s wall %.3f ms/step | GPU time of the replay %.3f (max %.3f) | host inside replay() %.3f (max %.3f) | host per call %.3f (max %.3f)" %
          (label, wall * 1e3, sum(gpu) / len(gpu), max(gpu), sum(hr) / len(hr) * 1e3, max(hr) * 1e3, sum(calls) / len(calls) * 1e3, max(calls) * 1e3), flush=True)
for rep in range(2):
    run("none", "resident batch, loader idle")
    run("beside", "resident batch, loader beside")
    run("fed", "fed by the loader")
# what a pop costs the consumer: queue get | wait_event | record_stream of the batch's tensors | dropping the previous batch
import queue, threading
def pop_costs():
    out_q, stop = queue.Queue(maxsize=loader.prefetch), threading.Event()
    worker = threading.Thread(target=loader._produce, args=(out_q, stop), daemon=True); worker.start()
    acc = [0.0] * 5; n = 0; prev = None
    while True:
        t0 = time.perf_counter(); got = out_q.get(); t1 = time.perf_counter()
        if got is None: break
        batch, ev = got
        cur = torch.cuda.current_stream(dev); cur.wait_event(ev); t2 = time.perf_counter()
        for t in batch.values():
            if isinstance(t, torch.Tensor): t.record_stream(cur)
        t3 = time.perf_counter(); prev = None; t4 = time.perf_counter()
        step(first); t5 = time.perf_counter()
        prev = batch; del batch, got
        if n >= 5:
            for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[i] += d
        n += 1
    worker.join()
    m = max(1, n - 5)
    print("per pop: queue.get %.3f ms | wait_event %.3f | record_stream %.3f | dropping the previous batch %.3f | step call %.3f" % tuple(a / m * 1e3 for a in acc))
pop_costs()
