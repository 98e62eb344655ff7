"""The graphed KPCN-Manifold step with and without the forked streams (weight-gradient side stream, branch stream of the
specular half), interleaved in one process.   python3 scripts/time_streams.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wcmc_amd import ops
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
batch = make_batch(8, 8, 128, seed=0, device=dev)
steps = {}
for name, side, branch in (("side + branch", True, True), ("branch only (default mode)", False, True), ("side only", True, False), ("one stream", False, False)):
    ops.USE_SIDE_STREAM, ops.USE_BRANCH_STREAM = side, branch
    itf = bench.build_interface(dev, None, rng="device")
    steps[name] = GraphedTrainStep(itf, batch, side_stream=side)
    for _ in range(5): steps[name](batch)
res = {k: [] for k in steps}
for rnd in range(4):
    for name, st in steps.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): st(batch)
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 20 * 1e3)
for name, v in res.items():
    print("%-26s %s ms  -> %.1f patches/s" % (name, " ".join("%.2f" % x for x in v), 8e3 / (sorted(v)[len(v) // 2])))
