"""How much of the graphed step is host-side serialisation (the finite check sync)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
itf = bench.build_interface(dev, None)
batch = make_batch(8, 8, 128, seed=0, device=dev)
g = GraphedTrainStep(itf, batch)
def run(label, fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); print("%-44s %.2f ms/step" % (label, (time.perf_counter() - t) / k * 1e3))
run("graphed step (as benchmarked)", lambda: g(batch))
run("graph replay only", lambda: g.graph.replay())
def no_sync():
    itf.preprocess(batch); g._draw(); g.fm._static_i = 0; g.graph.replay(); itf._optimization()
run("replay + draws + optimizer, no finite check", no_sync)
def no_draw():
    g.graph.replay(); itf._optimization()
run("replay + optimizer", no_draw)
