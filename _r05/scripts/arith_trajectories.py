"""Trajectory-level check of the opt-in forward arithmetics: 200 graphed steps at the benchmark shape in exact
fp32 and in every mode given on the command line, deviations from the fp32 run set against the recipe's own fp32-vs-fp32 spread
(profiles/r05_trajectory_spread.json: initial weights one ulp apart).

    python3 scripts/arith_trajectories.py bf16x321h,bf16x321o,bf16x321h+F [STEPS] > profiles/r05_arith_trajectories.txt
("<mode>+F": the mode with the KPCN chains' hidden activations rounded to fp16 by an emulation hook -- rung F of the forward ladder)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch

import train_trajectory as tt

def round_hidden_to_f16(xs, dims, ks):
    """Rung F of profiles/r04_forward_ladder.txt on the KPCN chains' hidden layers: the activation a layer leaves (hi + lo bf16
    planes, 16 bits) replaced by its fp16 rounding (11 bits), re-split exactly -- what a kernel that keeps hidden activations in
    fp16 and multiplies them by the full 16-bit weights (two MFMAs per product instead of three) would compute."""
    if ks != 5:
        return xs
    n, c, h, w = dims
    cp = (c + 7) // 8 * 8
    v = xs.view(torch.bfloat16).view(n, h, w, 2, cp).float()
    x16 = (v[:, :, :, 0] + v[:, :, :, 1]).half().float()
    hi = x16.bfloat16()
    lo = (x16 - hi.float()).bfloat16()
    return torch.stack([hi, lo], 3).reshape(-1).view(torch.int16)


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
    from wcmc_amd import ops

    for m in modes:
        emulate = m.endswith("+F")
        ops.EMULATE_HIDDEN = round_hidden_to_f16 if emulate else None
        try:
            cur, val = tt.run(m[:-2] if emulate else m, steps, nb, batches=batches, held_out=held)
        finally:
            ops.EMULATE_HIDDEN = None
        d = tt.deviations(cur, ref)
        print("%-10s validation %.6f (%.2e from fp32; fp32-vs-fp32 spread %.2e)" % (m, val, abs(val - vref) / vref, spread["validation_rel"]))
        for k in tt.KEYS:
            s = spread["spread"][k]
            print("   %-18s %.2e [%.2e] | %.2e [%.2e] | %.2e [%.2e] | %.2e [%.2e]" % (k, d[k][0], s[0], d[k][1], s[1], d[k][2], s[2], d[k][3], s[3]))
