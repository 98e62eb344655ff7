import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wcmc_amd import ops
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch
mode = sys.argv[1]
dev = torch.device("cuda", 0)
itf = bench.build_interface(dev, None)
batch = make_batch(2, 4, 64, seed=0, device=dev)
print("mode", mode, "branch", ops.USE_BRANCH_STREAM, flush=True)
if mode == "eager":
    for _ in range(3):
        itf.preprocess(batch); itf.train_batch(batch)
    torch.cuda.synchronize(); print("eager ok", flush=True)
else:
    g = GraphedTrainStep(itf, batch)
    print("captured", flush=True)
    for _ in range(3): g(batch)
    torch.cuda.synchronize(); print("graph ok", flush=True)
