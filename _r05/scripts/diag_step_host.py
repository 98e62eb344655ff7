"""Host time of the eager parts of the graphed step, phase by phase (GPU box):   python3 scripts/diag_step_host.py
Each phase is timed on the host with the GPU left running; the last column is the host's wait in the step's one sync."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
itf = bench.build_interface(dev, None)
fm = itf.loss_funcs["l_manif"]
fm.rng = "device"
batch = make_batch(8, 8, 128, seed=0, device=dev)
g = GraphedTrainStep(itf, batch)
names = ["preprocess", "batch copies", "pairings", "graph launch", "logging", "optimizer + sync"]
acc = [0.0] * len(names)
def step():
    t = [time.perf_counter()]
    itf.preprocess(batch); t.append(time.perf_counter())
    dst, src = [], []
    for k in g.keys:
        v, b = g.static[k], batch[k]
        if b.data_ptr() != v.data_ptr():
            dst.append(v); src.append(b)
    torch._foreach_copy_(dst, src); t.append(time.perf_counter())
    g._draw(); fm._static_i = 0; t.append(time.perf_counter())
    g.graph.replay(); t.append(time.perf_counter())
    itf._logging(g.losses); t.append(time.perf_counter())
    itf._optimization(); t.append(time.perf_counter())
    return [b - a for a, b in zip(t[:-1], t[1:])]
for _ in range(5): step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    for i, d in enumerate(step()): acc[i] += d
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
print("step wall %.3f ms; host phases (us): " % (wall * 1e3) + ", ".join("%s %.0f" % (nm, a / n * 1e6) for nm, a in zip(names, acc)))
print("host time before the graph is on the GPU: %.0f us per step" % (sum(acc[:4]) / n * 1e6))
