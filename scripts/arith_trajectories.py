"""Trajectory-level check of the opt-in forward arithmetics: 200 graphed steps at the benchmark shape in exact
fp32 and in every mode given on the command line, deviations from the fp32 run set against the recipe's own fp32-vs-fp32 spread
(profiles/r05_trajectory_spread.json: initial weights one ulp apart).

    python3 scripts/arith_trajectories.py bf16x321h,bf16x321o [STEPS] > profiles/r05_arith_trajectories.txt
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch

import train_trajectory as tt

if __name__ == "__main__":
    modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16x321h"]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    from wcmc_amd.synthetic import make_batch
    dev = torch.device("cuda", 0)
    nb = 16
    batches = [make_batch(8, 8, 128, seed=500 + i, device=dev) for i in range(nb)]
    held = make_batch(8, 8, 128, seed=999, device=dev)
    spread = json.load(open(os.path.join(ROOT, "profiles", "r05_trajectory_spread.json")))
    ref, vref = tt.run("fp32", steps, nb, batches=batches, held_out=held)
    print("# %d graphed steps, benchmark shape, same weights / batches / pairing keys; fp32 validation RelativeMSE %.6f" % (steps, vref))
    print("# columns: largest per-step relative difference over steps 1-40 | over steps 20.. | of the last-50 means | of the last-50 medians;"
          " in brackets: the same statistic's largest value over three fp32 runs that start one ulp apart")
    for m in modes:
        cur, val = tt.run(m, steps, nb, batches=batches, held_out=held)
        d = tt.deviations(cur, ref)
        print("%-10s validation %.6f (%.2e from fp32; fp32-vs-fp32 spread %.2e)" % (m, val, abs(val - vref) / vref, spread["validation_rel"]))
        for k in tt.KEYS:
            s = spread["spread"][k]
            print("   %-18s %.2e [%.2e] | %.2e [%.2e] | %.2e [%.2e] | %.2e [%.2e]" % (k, d[k][0], s[0], d[k][1], s[1], d[k][2], s[2], d[k][3], s[3]))
