"""The training recipe's OWN sensitivity to rounding: the bands of tests/test_gpu_trajectory.py.

A training run is a chaotic map: two correct arithmetics leave a loss plateau a few steps apart and their curves part by
several per cent for a while.  How far is "legitimately apart"?  This script trains the benchmark configuration N + 1 times in
EXACT fp32 (`--precision fp32`: v_mfma_f32_16x16x4_f32, one rounding per product) from initial weights that differ by at most
one unit in the last place (`train_trajectory.run(ulp_seed=...)`) -- the size of a single rounding difference -- with the same
batches and the same pairing keys, and records how far the perturbed runs end up from the unperturbed one in every statistic
the test holds.  The test's bands are K x the largest of them (and never below a floor that keeps the band meaningful).

    python3 scripts/trajectory_spread.py [STEPS] [NB] [N]      -> profiles/r05_trajectory_spread.json (+ a table on stdout)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import train_trajectory as tt

if __name__ == "__main__":
    argv = sys.argv[1:]
    steps = int(argv[0]) if argv else 200
    nb = int(argv[1]) if len(argv) > 1 else 16
    n = int(argv[2]) if len(argv) > 2 else 3
    from wcmc_amd.synthetic import make_batch
    dev = torch.device("cuda", 0)
    batches = [make_batch(8, 8, 128, seed=500 + i, device=dev) for i in range(nb)]
    held = make_batch(8, 8, 128, seed=999, device=dev)
    ref, vref = tt.run("fp32", steps, nb, batches=batches, held_out=held)
    spread = {k: [0.0, 0.0, 0.0, 0.0] for k in tt.KEYS}
    vals, rows = [], []
    for i in range(n):
        cur, val = tt.run("fp32", steps, nb, batches=batches, held_out=held, ulp_seed=101 + i)
        d = tt.deviations(cur, ref)
        vals.append(abs(val - vref) / vref)
        for k in tt.KEYS:
            spread[k] = [max(a, b if b == b else 0.0) for a, b in zip(spread[k], d[k])]
        rows.append((101 + i, val, d))
    out = {"steps": steps, "batches": nb, "runs": n, "arithmetic": "fp32 (exact MFMA), initial weights within one ulp of each other",
           "columns": ["steps 1-40 max", "steps 20.. max", "last-50 mean", "last-50 median"],
           "spread": spread, "validation_rel": max(vals), "validation_fp32": vref,
           "pathnet_weight_norm": True}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_trajectory_spread.json")
    for pth in (path, os.path.join(os.path.dirname(os.path.dirname(path)), "gpurun_out", "r05_trajectory_spread.json")):
        if os.path.isdir(os.path.dirname(pth)):       # (gpurun_out/ is what travels back from the GPU box)
            with open(pth, "w") as f:
                json.dump(out, f, indent=1)
    print("# fp32 vs fp32 (initial weights one ulp apart), %d steps over %d batches, %d perturbed runs; validation RelativeMSE %.6f" % (steps, nb, n, vref))
    for seed, val, d in rows:
        print("# ulp_seed %d: validation %.6f (%.2e from the unperturbed run)" % (seed, val, abs(val - vref) / vref))
        for k in tt.KEYS:
            print("    %-18s steps 1-40 %.2e  all %.2e  last-50 mean %.2e median %.2e" % ((k,) + d[k]))
    print("# largest over the runs:")
    for k in tt.KEYS:
        print("    %-18s steps 1-40 %.2e  all %.2e  last-50 mean %.2e median %.2e" % ((k,) + tuple(spread[k])))
    every = 10
    print("# the unperturbed fp32 run (every %dth step): " % every + " ".join(tt.KEYS))
    for i in range(0, steps, every):
        print("%5d " % (i + 1) + " ".join("%.6f" % ref[k][i] for k in tt.KEYS))
