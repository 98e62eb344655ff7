"""Bisect helper for stream-capture problems with the forked specular branch (one variant per process)."""
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
elif mode == "fwdonly":
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.no_grad():
            itf._manifold_forward(batch); 
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        with torch.no_grad():
            out = itf._manifold_forward(batch)
    print("captured fwd", flush=True)
    g.replay(); torch.cuda.synchronize(); print("fwdonly ok", flush=True)
else:
    if mode == "norecord":
        torch.Tensor.record_stream = lambda self, s: None
    g = GraphedTrainStep(itf, batch, side_stream=(mode != "noside"))
    print("captured", flush=True)
    for _ in range(3): g(batch)
    torch.cuda.synchronize(); print(mode, "graph ok", flush=True)
