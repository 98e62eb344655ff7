"""Training TRAJECTORIES of the three conv arithmetics ("missing" item 2 / next-round item 2a).

The reference's product is a training recipe (``README.md:45-56``, ``train_kpcn.py:359-381``), not one step: the same initial
weights and the same NB synthetic batches (cycled) are trained for STEPS graphed steps at the benchmark shape with the same
FeatureMSE pairings (``rng='device'``: the keys come from torch's CPU generator, re-seeded per run, so every mode replays
the same permutations) in the default mode (``ops.MODES[0]``), in ``bf16x3`` and in exact ``fp32``; the per-step loss scalars of
``loss_dict`` (``interfaces.py:221-249``) are recorded and a held-out batch is validated afterwards
(``KPCNInterface.validate_batch``, RelativeMSE of the denoised radiance).

    python3 scripts/train_trajectory.py [STEPS] [NB] > profiles/r04_trajectory.txt

Imported by tests/test_gpu_trajectory.py (``run``).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

KEYS = ("l_diffuse", "l_specular", "l_manif_diffuse", "l_manif_specular", "l_total", "rmse")


def run(mode, steps=200, nb=16, b=8, s=8, h=128, seed=1234, lr=1e-4, batches=None, held_out=None, ulp_seed=None):
    """Train `steps` graphed steps in arithmetic `mode`; returns ({key: [per-step value]}, validation RelativeMSE).
    ulp_seed: perturb every initial weight by at most one unit in the last place (a seeded factor 1 + d, |d| <= 2^-23) -- the
    size of ONE rounding difference, i.e. what any other correct arithmetic or summation order amounts to at step 0; the
    spread of such runs is the recipe's own sensitivity (scripts/trajectory_spread.py)."""
    import bench
    from wcmc_amd import ops
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.synthetic import make_batch
    old = ops.PRECISION
    ops.set_precision(mode)
    try:
        dev = torch.device("cuda", 0)
        itf = bench.build_interface(dev, None, rng="device")          # seed 0 weights, the bench's own constructor
        for o in itf.optims.values():
            o.param_groups[0]["lr"] = lr
        if ulp_seed is not None:
            g = torch.Generator(device="cpu").manual_seed(int(ulp_seed))
            with torch.no_grad():
                for m in itf.models.values():
                    for p in m.parameters():
                        p.mul_(1.0 + (torch.rand(p.shape, generator=g) - 0.5).to(dev) * 2.0 ** -22)
        if batches is None:
            batches = [make_batch(b, s, h, seed=500 + i, device=dev) for i in range(nb)]
        if held_out is None:
            held_out = make_batch(b, s, h, seed=999, device=dev)
        step = GraphedTrainStep(itf, batches[0])
        torch.manual_seed(seed)                                       # the pairing keys of every step, the same in every mode
        curves = {k: [] for k in KEYS}
        for i in range(steps):
            step(batches[i % len(batches)])
            vals = torch.stack([step.losses[k].reshape(()) for k in KEYS]).tolist()      # one sync per step
            for k, v in zip(KEYS, vals):
                curves[k].append(v)
        itf.to_eval_mode()
        with torch.no_grad():
            itf.validate_batch({k: v.clone() for k, v in held_out.items()})
        val = float(itf.m_losses["m_val"])
        step.close()
        return curves, val
    finally:
        ops.set_precision(old)


L1_KEYS = ("l_diffuse", "l_specular", "l_total", "rmse")
MANIF_KEYS = ("l_manif_diffuse", "l_manif_specular")


def deviations(cur, ref):
    """How far the curves `cur` are from `ref`, as {key: (early, overall, late_mean, late_median)}: the largest per-step relative difference
    over steps 1-40 (before the runs' rounding differences have been amplified by training: the deterministic regime), over all
    steps from 20 on (a training run is a chaotic map -- a loss plateau is left a few steps earlier or later and the curves part
    by several per cent for a while; what two correct arithmetics share is the SIZE of such excursions), and the relative
    difference of the means (and of the medians: the manifold terms are ~6e-5 with occasional one-step spikes of several times
    that, in every arithmetic) over the last 50 steps (where the run has settled)."""
    out = {}
    for k in KEYS:
        a, b = torch.tensor(cur[k], dtype=torch.float64), torch.tensor(ref[k], dtype=torch.float64)
        rel = (a - b).abs() / b.abs()
        late = lambda f: float((f(a[-50:]) - f(b[-50:])).abs() / f(b[-50:]))
        out[k] = (float(rel[:40].max()), float(rel[20:].max()), late(torch.mean), late(torch.median))
    return out


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = int(argv[0]) if argv else 200
    nb = int(argv[1]) if len(argv) > 1 else 16
    res = {}
    from wcmc_amd import ops as _ops
    DEF = _ops.MODES[0]
    for mode in ("fp32", "bf16x3", DEF):
        res[mode] = run(mode, steps, nb)
    print("# training trajectories, %d graphed steps over %d batches of 8 x 128x128 (S=8), lr 1e-4, same weights / batches / pairings" % (steps, nb))
    print("# validation RelativeMSE on a held-out batch after the run: " + "  ".join("%s %.6f" % (m, res[m][1]) for m in res))
    print("# relative difference of the loss curves (per step, same step of the other run): steps 1-40 max | steps 20-%d max | mean | median of the last 50 steps" % steps)
    for a, b in (("bf16x3", "fp32"), (DEF, "fp32"), (DEF, "bf16x3")):
        d = deviations(res[a][0], res[b][0])
        print("# %-8s vs %-6s: " % (a, b) + "  ".join("%s %.1e|%.1e|%.1e|%.1e" % ((k,) + d[k]) for k in KEYS))
    print("%5s | %s" % ("step", " | ".join("%-38s" % (k + " (fp32, bf16x3, %s)" % DEF) for k in KEYS)))
    every = 1 if steps <= 250 else 10
    for i in range(steps):
        if i < 20 or i % every == every - 1:
            print("%5d | %s" % (i + 1, " | ".join("%.6f %.6f %.6f              " % tuple(res[m][0][k][i] for m in ("fp32", "bf16x3", DEF)) for k in KEYS)))
