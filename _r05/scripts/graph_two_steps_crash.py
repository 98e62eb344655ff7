"""Smallest stand-alone form of the crash profiles/HISTORY.md section 5 records (ROCm 7.0 / PyTorch 2.10 on gfx950): with TWO captured training
steps alive in one process -- two sets of hipGraphExec objects with private memory pools, built one after the other -- a later
hipGraphLaunch of either can segfault inside the HIP runtime.  The product therefore keeps at most one captured step alive
(``GraphedTrainStep.close()``, ``capture_validated`` closes a rejected capture before it makes the next one).

    python3 scripts/graph_two_steps_crash.py [ROUNDS]        # exit code 0: no crash this time; a segfault kills the process

What it does: builds step A, replays it, builds step B WITHOUT closing A, then alternates replays of A and B.  With ``--close`` it
closes A before building B (the product's rule) and only replays B: that form has never crashed.  Nothing of the reference is
involved; the models are the benchmark's, the graphs are torch.cuda.CUDAGraph captures of this package's kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if __name__ == "__main__":
    rounds = int([a for a in sys.argv[1:] if not a.startswith("--")][0]) if [a for a in sys.argv[1:] if not a.startswith("--")] else 20
    close_first = "--close" in sys.argv
    import bench
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.synthetic import make_batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    batch = make_batch(2, 4, 64, seed=0, device=dev)
    itf_a = bench.build_interface(dev, None, rng="device")
    a = GraphedTrainStep(itf_a, batch, two_stream=False)
    a(batch)
    if close_first:
        a.close()
    itf_b = bench.build_interface(dev, None, rng="device")
    b = GraphedTrainStep(itf_b, batch, two_stream=False)
    for i in range(rounds):
        if not close_first:
            a(batch)
        b(batch)
        torch.cuda.synchronize()
        print("round %d ok" % i, flush=True)
    print("no crash in %d rounds (%s)" % (rounds, "A closed before B was captured" if close_first else "A and B both alive"))
