"""Capture after capture in ONE process: how much does the speed of the captured step depend on the capture?

The benchmark's step is captured N times in each of two forms -- the single forked hipGraph of rounds 2-4 and the two-stream
form (``GraphedTrainStep(two_stream=True)``: each half a linear graph of its own on its own stream) -- alternately, each capture
closed before the next (never two live steps), and every capture is timed over 20 replays with the optimiser held back by the
device guard (``GraphedTrainStep.time_replays``: the training state does not move, every capture times the same work).

    python3 scripts/graph_lottery3.py [N] > profiles/r05_graph_lottery.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    if len(sys.argv) > 2:                                   # A/B of ops.DEFER_MAX_BYTES (MiB): which slab reductions wait for the multi launch
        from wcmc_amd import ops as _o
        _o.DEFER_MAX_BYTES = int(float(sys.argv[2]) * (1 << 20))
        print("# DEFER_MAX_BYTES = %d" % _o.DEFER_MAX_BYTES)
    import bench
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.synthetic import make_batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    itf = bench.build_interface(dev, None, rng="device")
    batch = make_batch(bench.B_PER_GPU, bench.SPP, bench.PATCH, seed=0, device=dev)
    torch.manual_seed(1234)
    times = {False: [], True: []}
    for i in range(n):
        for two in (False, True):
            step = GraphedTrainStep(itf, batch, two_stream=two)
            t = step.time_replays(20)
            times[two].append(t)
            step.close()
            print("capture %2d %-11s %.3f ms per step" % (i, "two-stream" if two else "one graph", t), flush=True)
    for two in (False, True):
        ts = sorted(times[two])
        med = ts[len(ts) // 2]
        print("# %-11s %d captures: min %.3f  median %.3f  max %.3f ms  (max / min - 1 = %.1f %%; captures more than 2 %% above the fastest: %d)" %
              ("two-stream" if two else "one graph", len(ts), ts[0], med, ts[-1], (ts[-1] / ts[0] - 1) * 100, sum(t > 1.02 * ts[0] for t in ts)))
