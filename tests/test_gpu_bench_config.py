"""Parity of the configuration ``bench.py`` measures -- not a smaller or simpler relative of it.

``bench.build_interface`` + ``GraphedTrainStep`` with the library's default switches (split-bf16 GEMMs, forked
weight-gradient stream, forked specular stream, fused chain glue, fused 1x1 pairs, fused clip + Adam), BASELINE
configs[2] at its per-GPU shape (8 patches of 128x128, S=8), for two consecutive steps, against
``oracle.step.train_step`` on the same weights, inputs and FeatureMSE pairings (``rng='device'``, the bench's switch:
read back from the product; ``rng='cpu'``: the reference's ``torch.randperm`` stream, ``losses.py:35,50``).  What is compared, per step (``interfaces.py:122-251``):

  * every ``loss_dict`` scalar, 1e-3 relative (north star);
  * the denoised patches ``radiance / diffuse / specular`` (8,3,92,92), 1e-3 of the tensor's max (north star);
  * every parameter gradient with the flip-robust metric of ``conftest.assert_grad_close``: relative L2 and
    cosine, no fallback (the max-norm is printed for information only);
  * the parameter DELTAS of each Adam step against the oracle's (step 2 restarts the oracle from the product's
    weights, so that it compares two implementations of one step, not two networks a few sign ties apart).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from conftest import assert_grad_close, cosine, rel_l2      # noqa: E402
from oracle import step as ostep                             # noqa: E402
from oracle.models import KPCN as OKPCN                      # noqa: E402
from oracle.networks import PathNet as OPathNet              # noqa: E402

DEV = "cuda"
# Per-tensor gradient bar, no fallback: relative L2 and 1 - cosine against the fp32 CPU oracle, on THREE draws (weights, biases,
# weight_g, batches and pairing keys all move with the seed) and stated as a RATIO to what exact fp32 MFMA arithmetic
# (``--precision fp32``: one rounding per product) reaches on the same draw in the same test run -- the worst
# tensor is the draw's as much as the arithmetic's:
#     seed   default mode        exact fp32        ratio      (profiles/r06_grad_bar_calibration.txt, round-6 scene patches)
#     0      3.37e-3 / 5.7e-6    3.0e-4 / 4.5e-8   11 (6.7 against the floor)     KPCN diffuse layer 0, second step
#     1      1.82e-3 / 1.3e-6    3.9e-4 / 7.2e-8   4.6 (3.6)
#     2      1.36e-2 / 6.6e-5    3.8e-3 / 7.4e-6   3.5                            a badly conditioned draw: one PathNet, second step
# Bar: relative L2 <= GRAD_K x max(worst fp32 tensor of the draw, GRAD_FLOOR); 1 - cos <= (that bar)^2 / 2 (the same distance for
# a small angle).  GRAD_FLOOR is the yardstick's own resolution: the fp32 CPU oracle is up to 4.7e-4 from an fp64 run of itself
# (scripts/diag_grad_floor.py).  Which GEMM role carries the distance, and what a wider rung there would change, is measured in
# profiles/r06_grad_rungs.txt (PathNet.final's one-term weight gradient, on the ill-conditioned draw); the outputs and loss
# scalars north_star names are held at 1e-3 on every draw (measured 3.5e-4 / 2e-5).
GRAD_K, GRAD_FLOOR = 10.0, 5e-4
GRAD_L2, GRAD_COS = 5e-3, 1.25e-5             # (absolute form of the same bar on a well-conditioned draw: the C2 test below)
_FP32_WORST = {}


def _fp32_yardstick(weight_norm, seed):
    """Worst per-tensor relative L2 of the same two steps in exact fp32 MFMA arithmetic (cached per draw)."""
    key = (bool(weight_norm), int(seed))
    if key not in _FP32_WORST:
        from wcmc_amd import ops
        old = ops.PRECISION
        ops.set_precision("fp32")
        try:
            report, _ = parity_report("device", weight_norm, seed=seed, pin_defaults=False, bars=(1.0, 1.0))
        finally:
            ops.set_precision(old)
        _FP32_WORST[key] = max(r[1] for r in report if " grad " in r[0])
    return _FP32_WORST[key]


def _max_rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("rng_mode,weight_norm,seed", [("device", True, 0), ("device", True, 1), ("device", True, 2),
                                                       ("cpu", True, 1), ("device", False, 1)])
def test_benchmarked_configuration_two_steps_against_oracle(rng_mode, weight_norm, seed):
    """``weight_norm=True`` is the PathNet parametrisation ``bench.py`` runs (``config.pathnet_weight_norm``: upstream sbmc's
    ConvChain default, ``support/networks.py:18-24``), with ``weight_g`` moved off ``||weight_v||`` so that the normalisation
    acts; ``False`` the plain weights of rounds 1-4 (``bench.py --no-pathnet-weight-norm``, the line's ``other_parametrisation``
    leg).  Same bars for both.  ``rng_mode='device'`` is the switch ``bench.py`` runs with (``config.feature_mse_rng``): the pairings come from the
    keyed device bijection (``GraphedTrainStep._draw``), are read back from ``fm.static_perms`` after the replay and handed to
    the oracle; ``'cpu'`` is the reference's ``torch.randperm`` stream (``losses.py:35,50``), drawn identically on both sides."""
    yard = _fp32_yardstick(weight_norm, seed)
    l2 = GRAD_K * max(yard, GRAD_FLOOR)
    report, fails = parity_report(rng_mode, weight_norm, seed=seed, bars=(l2, 0.5 * l2 * l2))
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_config_parity_%s%s_seed%d.txt" % (rng_mode, "" if weight_norm else "_plain", seed)), "w") as f:
        worst = max(r[1] for r in report if " grad " in r[0])
        f.write("# seed %d: worst gradient tensor %.3e = %.1f x exact fp32 MFMA on this draw (%.3e); bar %.3e\n" % (seed, worst, worst / yard, yard, l2))
        for name, e, c, mx in sorted(report, key=lambda r: -r[1]):
            f.write("%-60s relL2/err %.3e%s%s\n" % (name, e, "" if c is None else "  1-cos %.2e" % c,
                                                    "" if mx is None else "  max-norm %.2e" % mx))
    assert not fails, "\n".join(fails)


def parity_report(rng_mode, weight_norm, seed=0, pin_defaults=True, bars=None):
    """The comparison itself; returns (report rows, failures).  seed: shifts the weights' seed, the bias / weight_g draws, the batches
    and the pairing keys together (scripts/calibrate_grad_bar.py runs several to put the gradient bar on more than one draw).
    bars: (relative L2, 1 - cos) per gradient tensor; default: the module's absolute pair."""
    grad_l2, grad_cos = bars if bars is not None else (GRAD_L2, GRAD_COS)
    import bench
    from wcmc_amd import ops
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.synthetic import make_batch
    assert not pin_defaults or (ops.PRECISION == os.environ.get("WCMC_PRECISION", ops.MODES[0]) and not ops.USE_SIDE_STREAM and ops.USE_BRANCH_STREAM and ops.FUSE_EMBED and ops.FUSE_FINAL), \
        "this test pins the DEFAULT switches (the ones bench.py runs with)"
    B, S, H = bench.B_PER_GPU, bench.SPP, bench.PATCH
    device = torch.device("cuda", 0)
    itf = bench.build_interface(device, None, rng=rng_mode, weight_norm=weight_norm, seed=seed)      # the bench's own constructor
    hmods = itf.models
    torch.manual_seed(0)
    omods = {"dncnn": OKPCN(39), "backbone_diffuse": OPathNet(36, weight_norm=weight_norm),
             "backbone_specular": OPathNet(36, weight_norm=weight_norm)}
    assert hmods["backbone_diffuse"].embedding.weight_norm == weight_norm and not hmods["dncnn"].diffuse.weight_norm
    # biases are zero at init (a degenerate case for bias-path bugs): give both sides the same random ones
    g = torch.Generator().manual_seed(77 + seed)
    for k, m in hmods.items():
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.endswith("bias"):
                    p.copy_((torch.rand(p.shape, generator=g) * 0.2 - 0.1).to(device))
                if n.endswith("weight_g"):      # g = ||v|| at init (w = v): move it so that g * v / ||v|| is not the identity
                    p.mul_((torch.rand(p.shape, generator=g) * 0.6 + 0.7).to(device))
        omods[k].load_state_dict({n: v.detach().cpu().clone() for n, v in m.state_dict().items()})
    oopt = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-4) for k, m in omods.items()}
    cfg = dict(use_llpm_buf=True, manif_learn=True, train_branches=True, disentanglement_option="m11r11", w_manif=0.1)
    batches = [make_batch(B, S, H, seed=40 + i + 10 * seed, device="cpu") for i in range(2)]
    dbatches = [{k: v.to(device) for k, v in b.items()} for b in batches]

    graphed = GraphedTrainStep(itf, dbatches[0])                          # capture (its warm-up draws pairings)
    p_start = {mn: {k: v.detach().cpu().clone() for k, v in m.named_parameters()} for mn, m in hmods.items()}
    for mn in omods:                                                      # warm-up runs no optimiser: still equal
        for k, q in omods[mn].named_parameters():
            assert torch.equal(p_start[mn][k], q.detach()), (mn, k)

    ho = H - 36
    torch.manual_seed(1234 + seed)
    perms = [[ostep.draw_perms(B, S, ho, ho), ostep.draw_perms(B, S, ho, ho)] for _ in range(2)]
    torch.manual_seed(1234 + seed)                                        # the graph draws the same stream, same order
    report, fails = [], []
    ograds = [{}, {}]
    lr = 1e-4
    p_prev = {mn: dict(d) for mn, d in p_start.items()}
    fm = itf.loss_funcs["l_manif"]
    assert fm.rng == rng_mode
    for step in range(2):
        if rng_mode == "device":
            # the product draws; the oracle is handed what it drew (every permutation checked to be a bijection)
            graphed(dbatches[step])
            torch.cuda.synchronize()
            perms[step] = [(ip.cpu().clone(), ib.cpu().clone()) for ip, ib in fm.static_perms]
            for pair in perms[step]:
                for t in pair:
                    assert torch.equal(torch.sort(t).values, torch.arange(t.numel())), "device pairing is not a permutation"
            assert not torch.equal(perms[step][0][0], perms[step][1][0]) and not torch.equal(perms[step][0][1], perms[step][1][1])
            if step == 1:
                assert not torch.equal(perms[0][0][0], perms[1][0][0]), "the pairings must change from step to step"
            loss_o, out_o = ostep.train_step(omods, oopt, batches[step], cfg, perms[step])
        else:
            loss_o, out_o = ostep.train_step(omods, oopt, batches[step], cfg, perms[step])
            graphed(dbatches[step])
            torch.cuda.synchronize()
            assert torch.equal(fm.static_perms[1][0].cpu(), perms[step][1][0])
        for k, v in loss_o.items():
            np.testing.assert_allclose(graphed.losses[k].item(), v.item(), rtol=1e-3, err_msg="step %d %s" % (step, k))
        for k in ("radiance", "diffuse", "specular"):
            e = _max_rel(itf.last_out[k], out_o[k])
            if e > 1e-3:
                fails.append("step %d denoised %s: %.3e" % (step, k, e))
            report.append(("step%d out %s" % (step, k), e, None, None))
        for mn in omods:
            for (k, p), (_, q) in zip(hmods[mn].named_parameters(), omods[mn].named_parameters()):
                got = p.grad.clamp(-1.0, 1.0)          # the oracle's .grad is post clip_grad_value_ (interfaces.py:260-261)
                ograds[step][(mn, k)] = q.grad.detach().clone()
                try:
                    e = assert_grad_close(got, q.grad, what="step %d grad %s %s" % (step, mn, k), l2=grad_l2, cos=grad_cos)
                except AssertionError as err:
                    fails.append(str(err))
                    e = rel_l2(got, q.grad)
                report.append(("step%d grad %s %s" % (step, mn, k), e, 1.0 - cosine(got, q.grad), _max_rel(got, q.grad)))
        # The Adam step itself: parameter DELTAS of this step against the oracle's.  Adam's first step is
        # -lr * g / (|g| + eps) = -lr * sign(g): an entry whose gradient is smaller than the gradient error may go the other
        # way (2 * lr apart) in two correct implementations.  So (i) entries whose oracle gradient is well conditioned (>= half
        # the tensor's rms, in every step so far) must agree to 5 % of lr, and (ii) at most 1 % of a tensor's entries (two
        # entries for small tensors) may be such ties (more than lr / 2 apart), and in the first step every one of them must BE a
        # tie: an oracle gradient below 5 % of the tensor's rms.  "No update" or "wrong sign" fails both everywhere.
        # (Round 6: one entry -> two.  With a gradient error of ~2e-3 of the rms an entry lies within the error of zero with
        # probability ~0.16 %; over the ~130 tensors of about a hundred entries and two steps a tensor with two such entries is
        # an event of every few draws, and a change of summation order -- the exact tile heights of the KPCN launches -- met one.)
        for mn in omods:
            for (k, p), (_, q) in zip(hmods[mn].named_parameters(), omods[mn].named_parameters()):
                d_h = p.detach().cpu() - p_prev[mn][k]
                d_o = q.detach() - p_prev[mn][k]
                well = torch.ones_like(d_o, dtype=torch.bool)
                for st in range(step + 1):
                    g = ograds[st][(mn, k)]
                    well &= g.abs() >= 0.5 * g.pow(2).mean().sqrt()
                worst = float((d_h - d_o)[well].abs().max()) if bool(well.any()) else 0.0   # (a 3-entry bias may have none)
                ties = int(((d_h - d_o).abs() > 0.5 * lr).sum())
                report.append(("step%d delta %s %s" % (step, mn, k), rel_l2(d_h, d_o), None, worst / lr))
                if worst > 0.05 * lr:
                    fails.append("step %d parameter delta %s %s: %.3e lr apart on a well-conditioned entry" % (step, mn, k, worst / lr))
                if ties > max(2, d_h.numel() // 100):
                    fails.append("step %d parameter delta %s %s: %d of %d entries more than lr/2 apart" % (step, mn, k, ties, d_h.numel()))
                if step == 0 and ties:
                    g0 = ograds[0][(mn, k)]
                    big = float(g0[(d_h - d_o).abs() > 0.5 * lr].abs().max()) / max(float(g0.pow(2).mean().sqrt()), 1e-30)
                    if big > 0.05:
                        fails.append("step 0 parameter delta %s %s: an entry whose oracle gradient is %.3f of the tensor's rms went the other way" % (mn, k, big))
                if float(d_h.abs().max()) <= 0.5 * lr:
                    fails.append("step %d: parameters of %s %s did not move" % (step, mn, k))
        # Step 2 starts from the PRODUCT's weights on both sides: the ~0.1 % of entries that took the other side of a sign
        # tie above would otherwise make step 2 compare two slightly different networks (measured: up to 2.7e-3 in relative
        # L2 on the KPCN input layer) instead of two implementations of the same step.  Adam's moments stay the oracle's own.
        for mn in omods:
            with torch.no_grad():
                for (k, p), (_, q) in zip(hmods[mn].named_parameters(), omods[mn].named_parameters()):
                    q.copy_(p.detach().cpu())
                    p_prev[mn][k] = p.detach().cpu().clone()
    graphed.close()
    return report, fails


def test_c2_vanilla_full_size_graphed_step_against_oracle():
    """BASELINE configs[1]: KPCN-Vanilla (diffuse + specular, n_in = 34, no PathNet, no manifold loss), 128x128, batch 8 on one
    MI355X, default switches, one hipGraph replay: loss scalars and denoised patches at 1e-3, gradients by relative L2."""
    import types
    from wcmc_amd import KPCN, ops
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.optim import FusedClipAdam
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import RelativeMSE
    from wcmc_amd.synthetic import make_batch
    assert ops.PRECISION == os.environ.get("WCMC_PRECISION", ops.MODES[0])      # (the default; WCMC_PRECISION probes another mode against the same bars)
    torch.manual_seed(11)
    omod = {"dncnn": OKPCN(34)}
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for n, p in omod["dncnn"].named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.rand(p.shape, generator=g) * 0.2 - 0.1)
    hmod = {"dncnn": KPCN(34)}
    hmod["dncnn"].load_state_dict(omod["dncnn"].state_dict())
    hmod["dncnn"].to(DEV)
    oopt = {"optim_dncnn": torch.optim.Adam(omod["dncnn"].parameters(), lr=1e-4)}
    hopt = {"optim_dncnn": torch.optim.Adam(hmod["dncnn"].parameters(), lr=1e-4)}
    lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(), "l_test": RelativeMSE()}
    itf = KPCNInterface(hmod, hopt, lf, types.SimpleNamespace(model_name="c2"), train_branches=True)
    itf.fused_optim = FusedClipAdam(hmod, hopt)
    itf.iters = 1
    itf.to_train_mode()
    batch = make_batch(8, 8, 128, seed=60, device="cpu", use_llpm=False)
    dbatch = {k: v.to(DEV) for k, v in batch.items()}
    step = GraphedTrainStep(itf, dbatch)
    loss_o, out_o = ostep.train_step(omod, oopt, batch, dict(use_llpm_buf=False, manif_learn=False, train_branches=True), None)
    step(dbatch)
    torch.cuda.synchronize()
    for k, v in loss_o.items():
        np.testing.assert_allclose(step.losses[k].item(), v.item(), rtol=1e-3, err_msg=k)
    for k in ("radiance", "diffuse", "specular"):
        assert _max_rel(itf.last_out[k], out_o[k]) <= 1e-3, k
    worst = max(((rel_l2(p.grad.clamp(-1.0, 1.0), q.grad), 1.0 - cosine(p.grad.clamp(-1.0, 1.0), q.grad), k)
                 for (k, p), (_, q) in zip(hmod["dncnn"].named_parameters(), omod["dncnn"].named_parameters())))
    print("C2 grad worst tensor: rel L2 %.3e 1-cos %.3e %s; outputs %s" %
          (worst + (" ".join("%.2e" % _max_rel(itf.last_out[k], out_o[k]) for k in ("radiance", "diffuse", "specular")),)))
    for (k, p), (_, q) in zip(hmod["dncnn"].named_parameters(), omod["dncnn"].named_parameters()):
        assert_grad_close(p.grad.clamp(-1.0, 1.0), q.grad, what="C2 grad " + k, l2=GRAD_L2, cos=GRAD_COS)
