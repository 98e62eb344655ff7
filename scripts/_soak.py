import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.optim import FusedClipAdam
from wcmc_amd.synthetic import make_batch
mode = sys.argv[1]
n = int(sys.argv[2])
dev = torch.device("cuda", 0)
if mode != "nopg":
    dist.init_process_group("nccl", store=dist.HashStore(), rank=0, world_size=1)
itf = bench.build_interface(dev, None, rng="device")
if mode == "split":
    itf.fused_optim = FusedClipAdam(itf.models, itf.optims, process_group=dist.group.WORLD, force_collective=True)
batch = make_batch(8, 8, 128, seed=0, device=dev)
if mode == "pg_used":          # a few collectives first, then the single-graph step with the group alive (what the test suite does)
    t = torch.ones(1 << 20, device=dev)
    for _ in range(20):
        w = dist.all_reduce(t, async_op=True); w.wait()
    torch.cuda.synchronize()
step = GraphedTrainStep(itf, batch)
t0 = time.time()
for i in range(n):
    step(batch)
    if i % 500 == 499:
        print(mode, i + 1, "steps", round(time.time() - t0, 1), "s", flush=True)
print(mode, "done", flush=True)
