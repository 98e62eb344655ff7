"""Host time of the launches in the eager tail of the graphed step (is any of them blocking?  no: 10-50 us each, the host
stays ahead of the GPU until the step's one sync).   python3 scripts/diag_tail_host.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch
from wcmc_amd import ops
dev = torch.device("cuda", 0)
itf = bench.build_interface(dev, None)
itf.loss_funcs["l_manif"].rng = "device"
batch = make_batch(8, 8, 128, seed=0, device=dev)
g = GraphedTrainStep(itf, batch)
T = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T.setdefault(label, []).append(time.perf_counter() - t); return r
    setattr(obj, name, w)
wrap(torch, "_foreach_copy_", "foreach_copy")
wrap(torch, "_foreach_add_", "foreach_add")
wrap(ops, "clip_adam_", "clip_adam launch")
wrap(torch, "stack", "stack")
wrap(torch, "cat", "cat")
for _ in range(5): g(batch)
torch.cuda.synchronize(); T.clear()
for _ in range(10): g(batch)
torch.cuda.synchronize()
for k, v in T.items():
    print("%-18s calls/step %.1f  host us: %s" % (k, len(v) / 10, " ".join("%.0f" % (x * 1e6) for x in v[:8])))
