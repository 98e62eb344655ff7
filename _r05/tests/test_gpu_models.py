"""Module- and step-level parity on the MI355X: PathNet, KPCN and one full KPCNInterface step of the
HIP path against (i) the CPU oracle with identical weights / inputs / permutations and (ii) the golden
fixtures produced by the REAL reference interface (tests/golden/interface_*.npz).

Tolerance: 1e-3 relative (north star) on outputs and loss scalars; gradients 1e-3 of their scale.
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as mg                      # noqa: E402  (geometry/builders only)
from conftest import DEBUG_LIB, FlipCounter, cosine, rel_l2   # noqa: E402
from oracle import step as ostep              # noqa: E402
from oracle.models import KPCN as OKPCN       # noqa: E402
from oracle.networks import PathNet as OPathNet   # noqa: E402

DEV = "cuda"


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def assert_close(a, b, tol=1e-3, what=""):
    assert tuple(a.shape) == tuple(b.shape), (what, a.shape, b.shape)
    e = rel_err(a, b)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)


def T(a):
    return torch.from_numpy(np.asarray(a))


_GRAD_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "grad_l2.txt")


def grad_close(got, want, l2, what, cos=None):
    """Flip-robust gradient parity (conftest.rel_l2): relative L2 per tensor, no fallback; the value is logged."""
    assert tuple(got.shape) == tuple(want.shape), (what, got.shape, want.shape)
    e = rel_l2(got, want)
    try:
        os.makedirs(os.path.dirname(_GRAD_LOG), exist_ok=True)
        with open(_GRAD_LOG, "a") as f:
            f.write("%-90s relL2 %.3e 1-cos %.2e\n" % (what, e, 1.0 - cosine(got, want)))
    except OSError:
        pass
    assert e <= l2, "%s: relative L2 %.3e > %.1e" % (what, e, l2)
    if cos is not None:
        assert 1.0 - cosine(got, want) <= cos, "%s: 1 - cosine %.3e" % (what, 1.0 - cosine(got, want))


def randomize_bias(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.rand(p.shape, generator=g) * 0.2 - 0.1)


def test_pathnet_matches_oracle(precision):
    from wcmc_amd.support.networks import PathNet
    torch.manual_seed(0)
    ref = OPathNet(36, intermc=16, outc=3)
    randomize_bias(ref, 1)
    mod = PathNet(36, intermc=16, outc=3)
    mod.load_state_dict(ref.state_dict())
    mod.to(DEV)
    assert str(mod) == str(ref) == "PathNet i36in16o3"
    g = torch.Generator().manual_seed(2)
    paths = torch.rand(2, 3, 36, 16, 24, generator=g) - 0.4
    with FlipCounter() as fc:
        out_r = ref({"paths": paths})
        gout = torch.rand(out_r.shape, generator=g) - 0.5
        out_r.backward(gout)
        batch = {"paths": paths.to(DEV)}
        out = mod(batch)
    assert out.shape == out_r.shape and (out >= 0).all()
    out.backward(gout.to(DEV))
    assert_close(out, out_r, what="PathNet fwd")
    for (k, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        fc.check(p.grad, q.grad, 1e-3, what="PathNet grad " + k, l2=2e-2)    # 16-channel test network: 37k units / layer


def test_pathnet_with_six_output_channels_runs_the_fused_chains_and_matches_the_oracle():
    """``--pnet_out_size 6`` (the reference's m10r01 / m11r01 runs, ``train_kpcn.py:209-212``; ``networks.py:23-24``): the default
    64-wide PathNet with SIX P-buffer channels takes the fused embedding and the fused final chain (round 4 widened
    ``wcmc_final2_*`` from four to eight output channels) and matches the CPU oracle, forward and gradients."""
    from wcmc_amd import ops
    from wcmc_amd.support.networks import PathNet
    torch.manual_seed(0)
    ref = OPathNet(36, outc=6)
    randomize_bias(ref, 1)
    mod = PathNet(36, outc=6)
    mod.load_state_dict(ref.state_dict())
    mod.to(DEV)
    g = torch.Generator().manual_seed(2)
    paths = torch.rand(2, 4, 36, 16, 24, generator=g) - 0.4
    out_r = ref({"paths": paths})
    gout = torch.rand(out_r.shape, generator=g) - 0.5
    out_r.backward(gout)
    calls = []
    real = ops._FinalFusedX.apply
    ops._FinalFusedX.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
    try:
        out = mod({"paths": paths.to(DEV)})
    finally:
        ops._FinalFusedX.apply = real
    assert calls == [1], "the six-channel final chain must take the fused kernel in the default mode"
    assert out.shape == out_r.shape == (2, 4, 6, 16, 24) and (out >= 0).all()
    out.backward(gout.to(DEV))
    assert_close(out, out_r, what="PathNet(outc=6) fwd")
    for (k, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        grad_close(p.grad, q.grad, 2e-2, "PathNet(outc=6) grad " + k)


def _move_g(m, seed):
    """weight_g = ||weight_v|| at initialisation (the effective weight is v): scale it so that the normalisation acts."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("weight_g"):
                p.mul_(torch.rand(p.shape, generator=g) + 0.5)


def test_pathnet_plain_weights_option_matches_oracle(precision):
    """``PathNet(weight_norm=False)``: the parametrisation of this build's rounds 1-4, kept selectable."""
    from wcmc_amd.support.networks import PathNet
    torch.manual_seed(0)
    ref = OPathNet(36, intermc=16, outc=3, weight_norm=False)
    randomize_bias(ref, 1)
    mod = PathNet(36, intermc=16, outc=3, weight_norm=False)
    assert [k for k, _ in mod.named_parameters()][:2] == ["embedding.layers.0.weight", "embedding.layers.0.bias"]
    mod.load_state_dict(ref.state_dict())
    mod.to(DEV)
    g = torch.Generator().manual_seed(2)
    paths = torch.rand(2, 3, 36, 16, 24, generator=g) - 0.4
    with FlipCounter() as fc:
        out_r = ref({"paths": paths})
        gout = torch.rand(out_r.shape, generator=g) - 0.5
        out_r.backward(gout)
        out = mod({"paths": paths.to(DEV)})
    out.backward(gout.to(DEV))
    assert_close(out, out_r, what="PathNet(plain) fwd")
    for (k, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        fc.check(p.grad, q.grad, 1e-3, what="PathNet(plain) grad " + k, l2=2e-2)


def test_pathnet_weight_norm_full_width_against_fp64(precision):
    """The default PathNet -- 64 wide, weight-normalised chains, ``weight_g`` moved off ``||weight_v||`` -- forward and the
    gradients of ``weight_g`` / ``weight_v`` / ``bias`` of all 20 layers against ``torch.nn.utils.weight_norm`` convolutions in
    fp64, and the launch structure: ONE ``wcmc_weight_norm_fwd`` and ONE ``wcmc_weight_norm_bwd`` per PathNet pass."""
    from wcmc_amd import ops
    from wcmc_amd.support.networks import PathNet
    torch.manual_seed(0)
    ref = OPathNet(36, outc=3)
    randomize_bias(ref, 1)
    _move_g(ref, 5)
    mod = PathNet(36, outc=3)
    mod.load_state_dict(ref.state_dict())
    mod.to(DEV)
    ref = ref.double()
    g = torch.Generator().manual_seed(2)
    paths = torch.rand(2, 4, 36, 32, 48, generator=g) - 0.4
    out_r = ref({"paths": paths.double()})
    gout = torch.rand(out_r.shape, generator=g) - 0.5
    out_r.backward(gout.double())
    calls = {"fwd": 0, "bwd": 0}
    real_f, real_b = ops._WeightNormMulti.forward, ops._WeightNormMulti.backward
    ops._WeightNormMulti.forward = staticmethod(lambda ctx, *a: (calls.__setitem__("fwd", calls["fwd"] + 1), real_f(ctx, *a))[1])
    ops._WeightNormMulti.backward = staticmethod(lambda ctx, *a: (calls.__setitem__("bwd", calls["bwd"] + 1), real_b(ctx, *a))[1])
    try:
        out = mod({"paths": paths.to(DEV)})
        out.backward(gout.to(DEV))
    finally:
        ops._WeightNormMulti.forward, ops._WeightNormMulti.backward = real_f, real_b
    assert calls == {"fwd": 1, "bwd": 1}, calls
    assert_close(out, out_r, what="PathNet(weight_norm) fwd")
    named_r = dict(ref.named_parameters())
    errs = []
    for k, p in mod.named_parameters():
        assert p.grad is not None and p.grad.shape == named_r[k].grad.shape, k
        errs.append((rel_l2(p.grad, named_r[k].grad.float()), k))
    try:
        with open(_AGG_LOG, "a") as f:
            for e, k in sorted(errs, reverse=True):
                f.write("PathNet(weight_norm) full width %-10s %-60s relL2 %.3e\n" % (precision, k, e))
    except OSError:
        pass
    bad = [(e, k) for e, k in errs if e > WN_GRAD_L2[precision]]
    assert not bad, "PathNet(weight_norm) %s gradients beyond %.1e: %s" % (precision, WN_GRAD_L2[precision], sorted(bad, reverse=True)[:5])


# per-tensor relative L2 of the weight-normalised PathNet's gradients against fp64 at THIS size (2 x 4 samples of 32 x 48 pixels), by
# arithmetic, measured on MI355X in round 5 (profiles/r05_golden_bars.txt): exact fp32 2.4e-6; the split-bf16 modes 1.75e-2 /
# 1.84e-2 -- ReLU and max-pool ties on 12k pixels; the same tensors sit at 1.2e-3 on the benchmark's 131k pixels
# (tests/test_gpu_bench_config.py, which is where per-tensor arithmetic parity is held).  Bars: >= 2x the measured worst tensor.
WN_GRAD_L2 = {"fp32": 2e-5, "bf16x3": 4e-2, "bf16x321": 4e-2, "bf16x321o": 4e-2, "bf16x321h": 4e-2}


def test_kpcn_c1_config_matches_oracle(precision):
    """BASELINE config C1: KPCN-Vanilla, 64x64, batch 2, n_in=34 (the reference's CPU-runnable case)."""
    from wcmc_amd import KPCN
    torch.manual_seed(3)
    ref = OKPCN(34)
    randomize_bias(ref, 4)
    mod = KPCN(34)
    mod.load_state_dict(ref.state_dict())
    mod.to(DEV)
    ref = ref.double()         # fp64 oracle: the fp32 CPU path carries its own ~1e-3 rounding through 9 layers
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.rand(*s, generator=g)
    batch = {"kpcn_diffuse_in": r(2, 34, 64, 64) - 0.3, "kpcn_specular_in": r(2, 34, 64, 64) - 0.3,
             "kpcn_diffuse_buffer": r(2, 3, 64, 64) * 2, "kpcn_specular_buffer": r(2, 3, 64, 64),
             "kpcn_albedo": r(2, 3, 64, 64) + 0.00316}
    with FlipCounter() as fc:
        out_r = ref({k: v.double() for k, v in batch.items()})
        out = mod({k: v.to(DEV) for k, v in batch.items()})
    assert out_r["radiance"].shape == (2, 3, 28, 28)
    (out_r["diffuse"].abs().mean() + out_r["specular"].abs().mean()).backward()
    (out["diffuse"].abs().mean() + out["specular"].abs().mean()).backward()
    for k in ("radiance", "diffuse", "specular"):
        assert_close(out[k], out_r[k], tol=1e-4 if precision == "fp32" else 1e-3, what="KPCN " + k)
    for (k, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        fc.check(p.grad, q.grad, 1e-4 if precision == "fp32" else 1e-3, what="KPCN grad " + k, l2=1e-2)   # 2 x 60 x 60 px


# Golden networks are 4..8 channels wide on 20x20 images (~1,600 units per layer): ONE ReLU unit or max-pool winner
# landing on the other side moves a gradient tensor by ~1/sqrt(units) = 2.5e-2 in relative L2 (measured values are
# logged to gpurun_out/grad_l2.txt).  The bars below are per tensor, relative L2, no fallback.
# Measured (MI355X, all golden cases, gpurun_out/grad_l2.txt): exact-fp32 MFMA <= 5.3e-6 whatever happens; split-bf16
# <= 4.9e-5 when no unit flipped against the oracle forward, <= 3.4e-2 with one flipped unit, 1.07e-1 with two.
_AGG_LOG = os.path.join(os.path.dirname(_GRAD_LOG), "golden_grad_aggregate.txt")
TENSOR_SANITY_L2 = 0.5


def golden_grads_close(named, want_of, bar, what, tensor_bar=TENSOR_SANITY_L2):
    """Gradient parity on the TINY golden networks (20 x 20 pixels, 4 .. 16 channels: the U-Net's deepest level has 50 units per
    channel).  One ReLU / max-pool / L1-sign tie that falls the other way in two correct arithmetics moves a 4-entry bias
    gradient there by tens of per cent, so a per-tensor bar at this geometry measures the draw, not the kernels (round 5
    re-drew every golden with weight-normalised PathNets: profiles/r05_golden_bars.txt).  What is held:
      * the gradients of ALL parameters of a model as ONE vector: relative L2 against the golden's <= ``bar`` (a tie moves a
        handful of its entries);
      * every tensor on its own <= 0.5 (a missing, mis-scaled, transposed or wrong-sign gradient is >= 1), exact-fp32 mode: <= bar.
    Per-tensor arithmetic parity is what tests/test_gpu_bench_config.py holds, at the benchmark's size.
    named: [(name, parameter)]; want_of(name) -> golden gradient (numpy; size 0 = the reference left ``.grad`` None)."""
    got, want = [], []
    worst = (0.0, "")
    for k, p in named:
        w = want_of(k)
        if w.size == 0:
            assert p.grad is None, "%s %s: the reference leaves this gradient None" % (what, k)
            continue
        assert p.grad is not None and tuple(p.grad.shape) == tuple(w.shape), (what, k)
        e = rel_l2(p.grad, T(w))
        worst = max(worst, (e, k))
        assert e <= tensor_bar, "%s %s: relative L2 %.3e > %.1e" % (what, k, e, tensor_bar)
        got.append(p.grad.detach().double().cpu().reshape(-1))
        want.append(T(w).double().reshape(-1))
    if not got:
        return
    agg = rel_l2(torch.cat(got), torch.cat(want))
    try:
        os.makedirs(os.path.dirname(_AGG_LOG), exist_ok=True)
        with open(_AGG_LOG, "a") as f:
            f.write("%-90s aggregate %.3e  worst tensor %.3e %s  bar %.1e\n" % (what, agg, worst[0], worst[1], bar))
    except OSError:
        pass
    assert agg <= bar, "%s: relative L2 of the model's whole gradient %.3e > %.1e (worst tensor %.3e %s)" % (what, agg, bar, worst[0], worst[1])


def tensor_l2(precision):
    """Per-tensor bar of the golden tests: exact fp32 arithmetic is held to 1e-4 tensor by tensor; the split-bf16 modes to the sanity
    bar (see ``golden_grads_close``)."""
    return 1e-4 if precision == "fp32" else TENSOR_SANITY_L2


def val_close(got, want, what):
    """Validation outputs AFTER the golden's Adam step (lr 2e-3 on the PathNets): an entry whose gradient is a sign tie moves by
    2 lr the other way in two correct arithmetics -- a whole output channel when it is a ``weight_g`` -- so the max-norm sees
    single pixels move by ~1e-2 (measured up to 7.6e-3 on the P-buffer); held: relative L2 <= 1e-2 (measured <= 4.3e-3) and max-norm <= 2e-2."""
    assert tuple(got.shape) == tuple(want.shape), (what, got.shape, want.shape)
    e2, em = rel_l2(got, want), rel_err(got, want)
    try:
        with open(_AGG_LOG, "a") as f:
            f.write("%-90s relL2 %.3e max-norm %.3e\n" % (what, e2, em))
    except OSError:
        pass
    assert e2 <= 1e-2 and em <= 2e-2, "%s: relative L2 %.3e (<= 1e-2), max-norm %.3e (<= 2e-2)" % (what, e2, em)


def golden_l2(precision, flips=None):
    """Bar on the relative L2 of a golden model's WHOLE gradient (``golden_grads_close``).  Measured on MI355X with round 5's
    goldens (profiles/r05_golden_bars.txt, every case x arithmetic x model): exact fp32 <= 3e-6; bf16x3 without a flipped unit
    <= 5e-5; the reduced-backward modes without a flip <= 2.7e-3 (dy / x rounded to bf16, 800 pixels to average over); any mode
    with ONE flipped ReLU unit <= 1.0e-2; the tests that do not count flips <= 9.0e-3.  Bars: >= 2x those."""
    if precision == "fp32":
        return 1e-4
    if flips is None:
        return 2e-2
    if flips == 0:
        return 8e-3 if precision in ("bf16x321h", "bf16x321", "bf16x321o") else 2e-4
    return 2e-2 * flips


def build_hip_models(case, d):
    """HIP-path models with the golden's initial weights."""
    from wcmc_amd import KPCN
    from wcmc_amd.support.networks import PathNet
    use_llpm, manif, tb, option, pout = mg.INTERFACE_CASES[case]
    G = mg.G5_GEOM
    n_in = G["BASE_IN"]
    if use_llpm:
        c_r = pout // 2 if option in ("m10r01", "m11r01") else pout
        n_in = n_in + 1 + c_r + 1
    models = {"dncnn": KPCN(n_in, ksize=G["KS"], depth=G["DEPTH"], width=G["WIDTH"])}
    if use_llpm:
        models["backbone_diffuse"] = PathNet(36, intermc=G["INTERMC"], outc=pout)
        models["backbone_specular"] = PathNet(36, intermc=G["INTERMC"], outc=pout)
    for mn, m in models.items():
        sd = {k[len("init/%s/" % mn):]: T(d[k]) for k in d.files if k.startswith("init/%s/" % mn)}
        m.load_state_dict(sd)
        m.to(DEV)
    return models


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", list(mg.INTERFACE_CASES))
def test_interface_step_against_reference_golden(golden_dir, case, fused, precision):
    """wcmc_amd.support.interfaces.KPCNInterface on the GPU vs the real reference KPCNInterface."""
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    d = np.load(os.path.join(golden_dir, "interface_%s.npz" % case))
    use_llpm, manif, tb, option, pout = mg.INTERFACE_CASES[case]
    models = build_hip_models(case, d)
    optims = {"optim_" + mn: torch.optim.Adam(m.parameters(), lr=1e-3 if mn == "dncnn" else 2e-3)
              for mn, m in models.items()}
    loss_funcs = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
                  "l_test": RelativeMSE()}
    if manif:
        loss_funcs["l_manif"] = FeatureMSE(non_local=True)
    itf = KPCNInterface(models, optims, loss_funcs, types.SimpleNamespace(model_name="golden"),
                        use_llpm_buf=use_llpm, manif_learn=manif, w_manif=0.1, train_branches=tb,
                        disentanglement_option=option)
    if fused:
        from wcmc_amd.optim import FusedClipAdam
        itf.fused_optim = FusedClipAdam(models, optims)
    itf.iters = 1
    batch = {k[len("batch/"):]: T(d[k]).to(DEV) for k in d.files if k.startswith("batch/")}
    itf.to_train_mode()
    # ReLU flips are counted against the oracle forward on the golden's initial weights
    omods = mg.build_models(case, 0)
    for mn, m in omods.items():
        m.load_state_dict({k[len("init/%s/" % mn):]: T(d[k]) for k in d.files if k.startswith("init/%s/" % mn)})
    ocfg = dict(use_llpm_buf=use_llpm, manif_learn=False, train_branches=tb, disentanglement_option=option)
    with FlipCounter() as fc:
        with torch.no_grad():
            ostep.forward_losses(omods, {k: v.cpu() for k, v in batch.items()}, ocfg, None, train=True)
        torch.manual_seed(int(d["seed"]))          # the reference's draws: same generator, same order
        itf.preprocess(batch)
        itf.train_batch(batch)
    nflips = fc.flips()
    if manif and tb:
        assert np.array_equal(loss_funcs["l_manif"].last_perms[0].numpy(), d["perm/specular_patch"])
    for k in d.files:
        if k.startswith("m_losses/") and k != "m_losses/m_val":
            np.testing.assert_allclose(itf.m_losses[k[len("m_losses/"):]].item(), d[k], rtol=1e-3, err_msg=k)
    for mn, m in models.items():
        golden_grads_close(list(m.named_parameters()), lambda k: d["grad/%s/%s" % (mn, k)], golden_l2(precision, nflips),
                           "golden %s %s fused=%d (%d flips) post-clip grad %s" % (case, precision, fused, nflips, mn),
                           tensor_bar=tensor_l2(precision))
        for k, v in m.state_dict().items():
            g = np.abs(d["grad/%s/%s" % (mn, k)])
            want, got = d["after/%s/%s" % (mn, k)], v.cpu().numpy()
            big = g > 1e-4                     # Adam step 1 is lr*sign(g): only well-conditioned entries
            np.testing.assert_allclose(got[big], want[big], rtol=1e-3, atol=5e-5, err_msg="after %s %s" % (mn, k))
            np.testing.assert_allclose(got[~big], want[~big], atol=4.1e-3)
    itf.to_eval_mode()
    with torch.no_grad():
        rad, pb = itf.validate_batch(batch)
    val_close(rad, T(d["val/radiance"]), "validate radiance %s %s" % (case, precision))
    np.testing.assert_allclose(itf.get_epoch_summary(mode="eval", norm=1), d["val/summary"], rtol=5e-3)
    if pb is not None:
        val_close(pb["diffuse"], T(d["val/p_diffuse"]), "validate p_buffer %s %s" % (case, precision))


@pytest.mark.parametrize("case", list(mg.VARIANT_CASES))
def test_ref_and_pre_interfaces_against_reference_golden(golden_dir, case, precision):
    """KPCNRefInterface / KPCNPreInterface on the GPU vs the real reference classes."""
    from wcmc_amd import KPCN
    from wcmc_amd.support import interfaces as itf_mod
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    d = np.load(os.path.join(golden_dir, "interface_%s.npz" % case))
    kind, manif, tb = mg.VARIANT_CASES[case]
    G = mg.G5_GEOM
    if kind == "KPCNRefInterface":
        models = {"dncnn": KPCN(G["BASE_IN"] + 3, ksize=G["KS"], depth=G["DEPTH"], width=G["WIDTH"])}
    else:
        models = {"dncnn": KPCN(G["BASE_IN"] + 5, ksize=G["KS"], depth=G["DEPTH"], width=G["WIDTH"]),
                  "backbone_diffuse": PathNet(36, intermc=G["INTERMC"], outc=3),
                  "backbone_specular": PathNet(36, intermc=G["INTERMC"], outc=3)}
    for mn, m in models.items():
        m.load_state_dict({k[len("init/%s/" % mn):]: T(d[k]) for k in d.files if k.startswith("init/%s/" % mn)})
        m.to(DEV)
    optims = {"optim_" + mn: torch.optim.Adam(m.parameters(), lr=1e-3 if mn == "dncnn" else 2e-3)
              for mn, m in models.items()}
    lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
          "l_test": RelativeMSE()}
    if manif:
        lf["l_manif"] = FeatureMSE(non_local=True)
    args = types.SimpleNamespace(model_name="golden")
    if kind == "KPCNRefInterface":
        itf = itf_mod.KPCNRefInterface(models, optims, lf, args, train_branches=tb)
    else:
        itf = itf_mod.KPCNPreInterface(models, optims, lf, args, manif_learn=manif, w_manif=0.1, train_branches=tb)
    itf.iters = 1
    batch = {k[len("batch/"):]: T(d[k]).to(DEV) for k in d.files if k.startswith("batch/")}
    itf.to_train_mode()
    assert [int(m.training) for m in models.values()] == list(d["train_flags"])
    torch.manual_seed(int(d["seed"]))
    itf.preprocess(batch)
    itf.train_batch(batch)
    if manif:
        assert np.array_equal(lf["l_manif"].last_perms[0].numpy(), d["perm/specular_patch"])
    for k in d.files:
        if k.startswith("m_losses/") and k != "m_losses/m_val":
            np.testing.assert_allclose(itf.m_losses[k[len("m_losses/"):]].item(), d[k], rtol=1e-3, err_msg=k)
    for mn, m in models.items():
        golden_grads_close(list(m.named_parameters()), lambda k: d["grad/%s/%s" % (mn, k)], golden_l2(precision),
                           "golden %s %s post-clip grad %s" % (case, precision, mn), tensor_bar=tensor_l2(precision))
        for k, v in m.state_dict().items():
            want, got = d["after/%s/%s" % (mn, k)], v.cpu().numpy()
            g = np.abs(d["grad/%s/%s" % (mn, k)])
            if g.size == 0 or not itf_mod.KPCNPreInterface._trained(itf, mn) if kind != "KPCNRefInterface" else False:
                np.testing.assert_array_equal(got, d["init/%s/%s" % (mn, k)], err_msg="frozen %s %s" % (mn, k))
                continue
            big = g > 1e-4
            np.testing.assert_allclose(got[big], want[big], rtol=1e-3, atol=5e-5, err_msg="after %s %s" % (mn, k))
            np.testing.assert_allclose(got[~big], want[~big], atol=4.1e-3)
    itf.to_eval_mode()
    with torch.no_grad():
        rad, pb = itf.validate_batch(batch)
    val_close(rad, T(d["val/radiance"]), "validate radiance %s %s" % (case, precision))
    np.testing.assert_allclose(itf.get_epoch_summary(mode="eval", norm=1), d["val/summary"], rtol=5e-3)


@pytest.mark.parametrize("case", list(mg.SAMPLE_CASES))
def test_sbmc_and_lbmc_interfaces_against_reference_golden(golden_dir, case, precision):
    """SBMCInterface / LBMCInterface on the HIP path -- PathNet backbone, FeatureMSE, the
    per-sample feature assembly kernel, the stand-in denoiser on the HIP conv ops -- against the REAL reference
    interfaces driven with the oracle's twins of the same modules (tests/golden/interface_{sbmc,lbmc}_*.npz)."""
    from standins import SampleDenoiserStandIn
    from wcmc_amd.support import interfaces as itf_mod
    from wcmc_amd.support import losses as pl
    from wcmc_amd.support.networks import PathNet
    d = np.load(os.path.join(golden_dir, "interface_%s.npz" % case))
    kind, use_llpm, manif, option, pout, recon, nfeat = mg.SAMPLE_CASES[case]
    G = mg.G7_GEOM
    c_r = ((pout // 2 if option in ("m10r01", "m11r01") else pout) + 1) if use_llpm else 0
    models = {"dncnn": SampleDenoiserStandIn(nfeat + c_r, width=G["WIDTH"], depth=G["DEPTH"])}
    if use_llpm:
        models["backbone"] = PathNet(36, intermc=G["INTERMC"], outc=pout)
    for mn, m in models.items():
        m.load_state_dict({k[len("init/%s/" % mn):]: T(d[k]) for k in d.files if k.startswith("init/%s/" % mn)})
        m.to(DEV)
    optims = {"optim_" + mn: torch.optim.Adam(m.parameters(), lr=1e-3 if mn == "dncnn" else 2e-3)
              for mn, m in models.items()}
    lf = {"l_recon": torch.nn.L1Loss() if recon == "L1Loss" else getattr(pl, recon)(), "l_test": pl.RelativeMSE()}
    if manif:
        lf["l_manif"] = pl.FeatureMSE(non_local=True)
    itf = getattr(itf_mod, kind)(models, optims, lf, types.SimpleNamespace(model_name="golden"), use_llpm_buf=use_llpm,
                                 manif_learn=manif, w_manif=0.1, disentangle=option)
    itf.iters = 1
    batch = {k[len("batch/"):]: T(d[k]).to(DEV) for k in d.files if k.startswith("batch/")}
    itf.to_train_mode()
    torch.manual_seed(int(d["seed"]))
    itf.preprocess(batch)
    itf.train_batch(batch)
    if manif:
        assert np.array_equal(lf["l_manif"].last_perms[0].numpy(), d["perm/patch"])
    for k in d.files:
        if k.startswith("m_losses/") and k != "m_losses/m_val":
            np.testing.assert_allclose(itf.m_losses[k[len("m_losses/"):]].item(), d[k], rtol=1e-3, err_msg=k)
    for mn, m in models.items():
        golden_grads_close(list(m.named_parameters()), lambda k: d["grad/%s/%s" % (mn, k)], golden_l2(precision),
                           "golden %s %s post-clip grad %s" % (case, precision, mn), tensor_bar=tensor_l2(precision))
        norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
        np.testing.assert_allclose(norm, d["gradnorm/" + mn], rtol=5e-2)
        for k, v in m.state_dict().items():
            want, got = d["after/%s/%s" % (mn, k)], v.cpu().numpy()
            big = np.abs(d["grad/%s/%s" % (mn, k)]) > 1e-4 * max(1.0, float(d["gradnorm/" + mn]))
            np.testing.assert_allclose(got[big], want[big], rtol=1e-3, atol=5e-5, err_msg="after %s %s" % (mn, k))
            np.testing.assert_allclose(got[~big], want[~big], atol=4.1e-3)
    itf.to_eval_mode()
    with torch.no_grad():
        out, pb = itf.validate_batch(batch)
    # validation runs on the weights AFTER the Adam step, -lr * sign(g) on entries whose gradient is below the gradient noise:
    # in these 4-channel networks the bf16-rounded backward operands of the default mode turn more of them (one ReLU output
    # pixel of the P-buffer then differs by a few per cent of the tensor's max)
    vt = 5e-2 if precision in ("bf16x321h", "bf16x321", "bf16x321o") else 5e-3
    assert_close(out, T(d["val/out"]), tol=vt, what="validate output")
    np.testing.assert_allclose(itf.get_epoch_summary(mode="eval", norm=1), d["val/summary"], rtol=vt)
    if pb is not None:
        assert_close(pb, T(d["val/p_buffer"]), tol=vt, what="validate p_buffer")


def test_full_size_step_against_oracle(precision):
    """One KPCN-Manifold step at the benchmark geometry (128x128, S=8, pnet_out 3) with B=1 against the
    CPU oracle: same weights, inputs and permutations."""
    from wcmc_amd import KPCN
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    from wcmc_amd.synthetic import make_batch
    torch.manual_seed(7)
    omods = {"dncnn": OKPCN(39), "backbone_diffuse": OPathNet(36), "backbone_specular": OPathNet(36)}
    hmods = {"dncnn": KPCN(39), "backbone_diffuse": PathNet(36), "backbone_specular": PathNet(36)}
    for k in omods:
        randomize_bias(omods[k], 8)
        hmods[k].load_state_dict(omods[k].state_dict())
        hmods[k].to(DEV)
    batch = make_batch(1, 8, 128, seed=9, device="cpu")
    cfg = dict(use_llpm_buf=True, manif_learn=True, train_branches=True, disentanglement_option="m11r11",
               w_manif=0.1)
    torch.manual_seed(10)
    perms = [ostep.draw_perms(1, 8, 92, 92), ostep.draw_perms(1, 8, 92, 92)]
    oopt = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-4) for k, m in omods.items()}
    fc = FlipCounter().__enter__()
    loss_o, out_o = ostep.train_step(omods, oopt, batch, cfg, perms)
    hopt = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-4) for k, m in hmods.items()}
    lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
          "l_test": RelativeMSE(), "l_manif": FeatureMSE(non_local=True)}
    itf = KPCNInterface(hmods, hopt, lf, types.SimpleNamespace(model_name="t"), use_llpm_buf=True,
                        manif_learn=True, w_manif=0.1, train_branches=True)
    itf.iters = 1
    itf.to_train_mode()
    dbatch = {k: v.to(DEV) for k, v in batch.items()}
    torch.manual_seed(10)
    itf.preprocess(dbatch)
    itf.train_batch(dbatch)
    fc.__exit__()
    nfl = fc.flips()
    for k, v in loss_o.items():
        np.testing.assert_allclose(itf.m_losses["m_" + k].item(), v.item(), rtol=1e-3, err_msg=k)
    for k in ("radiance", "diffuse", "specular"):        # the denoised patches (north star: 1e-3)
        assert_close(itf.last_out[k], out_o[k], tol=1e-4 if precision == "fp32" else 1e-3, what="denoised " + k)
    for mn in omods:
        for (k, p), (_, q) in zip(hmods[mn].named_parameters(), omods[mn].named_parameters()):
            # one patch: 1/8 of the benchmark's units, so sqrt(8) x its relative L2 (tests/test_gpu_bench_config.py holds
            # B=8 to 2e-3); measured here 3.9e-3 (exact-fp32 MFMA: fp32 against fp32 in another summation order) and 3.5e-3
            # (the opt-in output-layer modes move the outputs by 1e-5 .. 1e-4 and with them the sign of the L1 derivative at the
            # pixels whose residual is that small: measured 1 - cos 2.1e-5 with the fp16 layer; the default's bar stays)
            grad_close(p.grad, q.grad, 8e-3, "full-size B=1 %s (%d flips) grad %s %s" % (precision, nfl, mn, k),
                       cos=3e-5 if precision in ("bf16x321h", "bf16x321o") else 2e-5)


def test_graphed_step_equals_eager_step():
    """GraphedTrainStep (one hipGraph replay; the optimiser eager behind it, or captured with it) == preprocess + train_batch,
    bit for bit in fp32 MFMA mode (deterministic kernels, same CPU-generator pairings), including a learning-rate change."""
    from wcmc_amd import KPCN, ops
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.optim import FusedClipAdam
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    from wcmc_amd.synthetic import make_batch
    old = ops.PRECISION
    ops.set_precision("fp32")
    try:
        results = []
        # eager | graph + eager optimiser | graph with the optimiser captured | the same with the deferred non-finite check | the halves
        # as two graphs on two streams + a tail graph | the validated capture (two captures, the faster within 5 % kept)
        for graphed in (False, True, "tail", "deferred", "two_stream", "validated"):
            torch.manual_seed(21)
            kw = dict(ksize=21, depth=3, width=24)
            models = {"dncnn": KPCN(39, **kw), "backbone_diffuse": PathNet(36, intermc=16),
                      "backbone_specular": PathNet(36, intermc=16)}
            for m in models.values():
                m.to(DEV)
            optims = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-3) for k, m in models.items()}
            lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
                  "l_test": RelativeMSE(), "l_manif": FeatureMSE(non_local=True, rng="cpu")}
            itf = KPCNInterface(models, optims, lf, types.SimpleNamespace(model_name="g"), use_llpm_buf=True,
                                manif_learn=True, w_manif=0.1, train_branches=True)
            itf.fused_optim = FusedClipAdam(models, optims)
            itf.iters = 1
            itf.to_train_mode()
            batches = [make_batch(2, 4, 48, seed=30 + i, device=DEV) for i in range(3)]
            if graphed == "validated":
                from wcmc_amd.graph import capture_validated
                step = capture_validated(itf, batches[0], two_stream=True)
                assert step.tail_captured and 1 <= step.capture_attempts <= 3 and len(step.capture_ms) == step.capture_attempts
            elif graphed:
                step = GraphedTrainStep(itf, batches[0], capture_optimizer=(graphed in ("tail", "deferred", "two_stream")),
                                        defer_check=(graphed == "deferred"), two_stream=(graphed == "two_stream"))
                assert step.tail_captured == (graphed in ("tail", "deferred", "two_stream"))
                if graphed == "two_stream":
                    assert len(step.half_graphs) == 2 and step.graph_h is not None
                    assert step.time_replays(3) > 0.0                      # (timing replays leave the training state alone)
            else:
                def step(b):
                    itf.preprocess(b)
                    itf.train_batch(b)
            torch.manual_seed(22)
            for i, b in enumerate(batches):
                if i == 2:                              # a learning-rate change between steps must reach a captured optimiser
                    for o in optims.values():
                        o.param_groups[0]["lr"] = 3e-4
                step(b)
            if graphed:
                step.flush()
                step.close()
            results.append(({k: v.item() for k, v in itf.m_losses.items()},
                            torch.cat([p.detach().reshape(-1) for m in models.values() for p in m.parameters()]).cpu(),
                            itf.iters, [float(o.state[next(iter(o.state))]["step"]) for o in optims.values()]))
        l0, p0, i0, s0 = results[0]
        for l1, p1, i1, s1 in results[1:]:
            assert i0 == i1 == 4 and s0 == s1 == [3.0, 3.0, 3.0]
            assert l0.keys() == l1.keys()
            for k in l0:
                np.testing.assert_allclose(l1[k], l0[k], rtol=1e-6, err_msg=k)
            assert torch.equal(p0, p1)
    finally:
        ops.set_precision(old)


def test_first_layer_data_gradient_restricted_to_the_pbuffer_channels_changes_nothing_downstream(monkeypatch):
    """``ops.pbuffer_cat`` marks the channels of its output whose gradient its backward reads (the P-buffer's mean: 3 of 39);
    ``ops.conv_chain`` then forms the first layer's data gradient for the 8-aligned rows round them only.  The gradients of the
    P-buffer and of every chain parameter must equal those of the full data gradient (same arithmetic per element; the sliced
    launch runs another tile shape, hence rounding-level differences only)."""
    from wcmc_amd import ops
    from wcmc_amd.modules import ConvChain
    torch.manual_seed(5)
    chain = ConvChain(39, 24, depth=3, width=48, ksize=5, pad=False, output_type="linear", weight_norm=False).to(DEV)
    base = torch.randn(2, 35, 40, 40, device=DEV)
    p0 = torch.rand(2, 4, 3, 40, 40, device=DEV)
    gout = torch.randn(2, 24, 28, 28, device=DEV)
    res = {}
    for on in (True, False):
        chain.zero_grad()
        p = p0.clone().requires_grad_(True)
        x = ops.pbuffer_cat(base, p)
        assert x._wcmc_grad_channels == (35, 38)
        if not on:
            del x._wcmc_grad_channels              # without the hint the chain forms the full data gradient
        y = chain(x)
        y.backward(gout)
        res[on] = (y.detach().clone(), p.grad.clone(), [q.grad.clone() for q in chain.parameters()])
    assert torch.equal(res[True][0], res[False][0])
    assert rel_l2(res[True][1], res[False][1]) <= (2e-3 if ops.reduced_backward() else 1e-5)       # (two-term instance against the generic three-term one)
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)                              # the weight gradients do not see the slice at all


def test_graphed_step_close_releases_its_graphs_and_memory():
    """``GraphedTrainStep.close()``: the graphs, their memory pool and the static batch go at once (a session that builds one
    graphed step after another keeps one alive); the interface trains on eagerly or under a new capture, the closed object
    refuses to be called."""
    import gc
    import bench
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.synthetic import make_batch
    device = torch.device("cuda", 0)
    itf = bench.build_interface(device, None, rng="device")
    batch = make_batch(2, 4, 64, seed=72, device=device)
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    before = torch.cuda.memory_reserved(device)
    step = GraphedTrainStep(itf, batch)
    torch.manual_seed(73)
    step(batch)
    held = torch.cuda.memory_reserved(device) - before
    assert held > 50e6                                   # the captured step's activations live in the graph's private pool
    step.close()
    gc.collect()
    torch.cuda.empty_cache()
    assert torch.cuda.memory_reserved(device) - before < 0.25 * held
    with pytest.raises((AttributeError, TypeError)):
        step(batch)
    again = GraphedTrainStep(itf, batch)                 # the interface is reusable: a new capture, more steps
    again(batch)
    assert torch.isfinite(again.losses["l_total"]).item() and itf.iters >= 3
    again.close()
    itf.preprocess(batch)
    itf.train_batch(batch)                               # and eagerly (the loss draws its own pairings again)


def test_captured_optimizer_tail_guard_and_epoch_summary():
    """The step's tail inside the hipGraph (one rank): a non-finite loss raises the reference's error
    (``interfaces.py:254-257``) with parameters, moments, step counters and running sums untouched -- the update sits behind a
    device guard -- and training continues afterwards; ``get_epoch_summary`` (which swaps the running sums for fresh zeros,
    ``interfaces.py:320-333``) keeps working between replays."""
    import bench
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.synthetic import make_batch
    device = torch.device("cuda", 0)
    itf = bench.build_interface(device, None, rng="device")
    good = make_batch(2, 4, 64, seed=70, device=device)
    step = GraphedTrainStep(itf, good)
    assert step.tail_captured
    torch.manual_seed(71)
    step(good)
    step(good)
    flat = lambda: torch.cat([fl.flat for fl in itf.fused_optim.flats.values()]).clone()
    moments = lambda: torch.cat([fl.m for fl in itf.fused_optim.flats.values()]).clone()
    p_before, m_before = flat(), moments()
    sums_before = {k: v.item() for k, v in itf.m_losses.items()}
    steps_before = [fl.steps for fl in itf.fused_optim.flats.values()]
    assert steps_before == [2, 2, 2]
    bad = {k: v.clone() for k, v in good.items()}
    bad["target_diffuse"][0, 0, 30, 30] = float("nan")
    with pytest.raises(RuntimeError, match="Non-finite loss at train time"):
        step(bad)
    assert torch.equal(flat(), p_before) and torch.equal(moments(), m_before)
    assert [fl.steps for fl in itf.fused_optim.flats.values()] == steps_before
    assert {k: v.item() for k, v in itf.m_losses.items()} == sums_before
    step(good)                                          # and on it goes
    assert [fl.steps for fl in itf.fused_optim.flats.values()] == [3, 3, 3] and not torch.equal(flat(), p_before)
    three = {k: v.item() for k, v in itf.m_losses.items()}
    assert all(three[k] > sums_before[k] for k in three if k != "m_val")
    assert itf.get_epoch_summary(mode="train", norm=3) == -1.0            # prints the means and zeroes the sums
    assert all(v.item() == 0.0 for k, v in itf.m_losses.items() if k != "m_val")
    step(good)
    one = {k: v.item() for k, v in itf.m_losses.items()}
    assert all(0.0 < one[k] < three[k] for k in one if k != "m_val"), (one, three)


def test_deferred_check_raises_one_call_later_with_the_update_skipped():
    """``GraphedTrainStep(defer_check=True)``: the host does not wait for a step before it enqueues the next one.  A non-finite
    loss still leaves parameters and moments untouched AT ONCE (device guard); the reference's error surfaces at the next
    call -- or at ``flush()`` when there is none -- and the step counters are taken back."""
    import bench
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.synthetic import make_batch
    device = torch.device("cuda", 0)
    itf = bench.build_interface(device, None, rng="device")
    good = make_batch(2, 4, 64, seed=70, device=device)
    step = GraphedTrainStep(itf, good, defer_check=True)
    assert step.tail_captured and step.defer_check
    torch.manual_seed(71)
    step(good)
    step(good)
    step.flush()
    flat = lambda: torch.cat([fl.flat for fl in itf.fused_optim.flats.values()]).clone()
    p_before = flat()
    bad = {k: v.clone() for k, v in good.items()}
    bad["target_diffuse"][0, 0, 30, 30] = float("nan")
    step(bad)                                           # returns: its check is pending
    torch.cuda.synchronize()
    assert torch.equal(flat(), p_before)                # ... but the device guard has already skipped the update
    with pytest.raises(RuntimeError, match="Non-finite loss at train time"):
        step.flush()
    assert [fl.steps for fl in itf.fused_optim.flats.values()] == [2, 2, 2]
    step(bad)
    with pytest.raises(RuntimeError, match="Non-finite loss at train time"):
        step(good)                                      # raised by the NEXT call when there is one
    # ... and that next step, enqueued behind the non-finite one before its flags were read, did NOT update either (ADVICE r3:
    # the failed guard poisons the guards behind it until the error has been raised; interfaces.py:254-257 aborts before
    # optim.step, so no update may follow a non-finite step unseen)
    torch.cuda.synchronize()
    assert torch.equal(flat(), p_before)
    assert [fl.steps for fl in itf.fused_optim.flats.values()] == [2, 2, 2] and step._pending is None
    sums = {k: v.item() for k, v in itf.m_losses.items()}
    step(good)                                          # the poison is cleared with the raise: training goes on
    step.flush()
    assert [fl.steps for fl in itf.fused_optim.flats.values()] == [3, 3, 3] and not torch.equal(flat(), p_before)
    assert all(itf.m_losses[k].item() > sums[k] for k in sums if k != "m_val")


def test_captured_optimizer_adopts_state_loaded_after_construction():
    """an optimiser state loaded AFTER ``FusedClipAdam`` was built (``init_model`` resumes that way) must be adopted
    before the capture begins -- adopted inside it, the moment copies would be recorded into the hipGraph and every replay would
    reset Adam's moments to the checkpoint's.  Two graphed steps, checkpoint, two more; against a fresh interface that loads
    the checkpoint into already-built optimisers and runs the same two steps: parameters, moments and step counts bit-identical."""
    import copy
    import bench
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.synthetic import make_batch
    device = torch.device("cuda", 0)
    batches = [make_batch(2, 4, 64, seed=80 + i, device=device) for i in range(4)]
    itf = bench.build_interface(device, None, rng="cpu")
    step = GraphedTrainStep(itf, batches[0])
    torch.manual_seed(81)
    step(batches[0]); step(batches[1])
    w = {n: copy.deepcopy(m.state_dict()) for n, m in itf.models.items()}
    o = {n: copy.deepcopy(op.state_dict()) for n, op in itf.optims.items()}
    torch.manual_seed(82)
    step(batches[2]); step(batches[3])
    flat = lambda i: torch.cat([torch.cat([fl.flat, fl.m, fl.v]) for fl in i.fused_optim.flats.values()]).clone()
    want = flat(itf)

    itf2 = bench.build_interface(device, None, rng="cpu")           # FusedClipAdam built here ...
    for n, m in itf2.models.items():
        m.load_state_dict(w[n])
    for n, op in itf2.optims.items():
        op.load_state_dict(o[n])                                   # ... the state arrives afterwards
    assert not any(fl.bound(itf2.optims["optim_" + n]) for n, fl in itf2.fused_optim.flats.items())
    step2 = GraphedTrainStep(itf2, batches[0])
    assert step2.tail_captured and all(fl.bound(itf2.optims["optim_" + n]) for n, fl in itf2.fused_optim.flats.items())
    torch.manual_seed(82)
    step2(batches[2]); step2(batches[3])
    assert torch.equal(flat(itf2), want)
    for op in itf2.optims.values():
        assert all(float(st["step"]) == 4.0 for st in op.state_dict()["state"].values())


def test_graphed_unfused_step_with_grad_sync_equals_eager(rccl_one_rank_group):
    """GraphedTrainStep on the UN-fused path INTEGRATION.md documents (itf.grad_sync = wd.average_gradients, torch's
    clip_grad_value_ + Adam.step): ``p.grad`` must keep pointing at the buffers the captured backward writes, so the
    gradient average is written back in place.  Three steps on three different batches against the eager step, bit for
    bit, default (split-bf16) arithmetic, over a one-rank RCCL group."""
    import torch.distributed as dist
    from wcmc_amd import KPCN
    from wcmc_amd import distributed as wd
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    from wcmc_amd.synthetic import make_batch
    own_group = False                                   # (the session's group: conftest.rccl_one_rank_group)
    try:
        results = []
        for graphed in (False, True):
            torch.manual_seed(21)
            kw = dict(ksize=21, depth=3, width=24)
            models = {"dncnn": KPCN(39, **kw), "backbone_diffuse": PathNet(36, intermc=16),
                      "backbone_specular": PathNet(36, intermc=16)}
            for m in models.values():
                m.to(DEV)
            optims = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-3) for k, m in models.items()}
            lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
                  "l_test": RelativeMSE(), "l_manif": FeatureMSE(non_local=True, rng="cpu")}
            itf = KPCNInterface(models, optims, lf, types.SimpleNamespace(model_name="g"), use_llpm_buf=True,
                                manif_learn=True, w_manif=0.1, train_branches=True)
            itf.grad_sync = wd.average_gradients
            itf.iters = 1
            itf.to_train_mode()
            batches = [make_batch(2, 4, 48, seed=30 + i, device=DEV) for i in range(3)]
            if graphed:
                step = GraphedTrainStep(itf, batches[0])
            else:
                def step(b):
                    itf.preprocess(b)
                    itf.train_batch(b)
            torch.manual_seed(22)
            trace = []
            for b in batches:
                step(b)
                trace.append(torch.cat([p.detach().reshape(-1) for m in models.values() for p in m.parameters()]).cpu())
            results.append(({k: v.item() for k, v in itf.m_losses.items()}, trace))
        (l0, t0), (l1, t1) = results
        for k in l0:
            np.testing.assert_allclose(l1[k], l0[k], rtol=1e-6, err_msg=k)
        for i, (a, b) in enumerate(zip(t0, t1)):
            assert torch.equal(a, b), "parameters differ after step %d" % (i + 1)
        assert not torch.equal(t0[0], t0[1]) and not torch.equal(t0[1], t0[2])
    finally:
        if own_group:
            dist.destroy_process_group()


def test_collective_branch_on_a_one_rank_rccl_group_equals_the_world_1_path(rccl_one_rank_group):
    """``FusedClipAdam.step``'s collective branch -- three asynchronous bucket all-reduces on RCCL's stream,
    ``w.wait()`` per bucket, the guard flag in the first bucket's slot, scale -> clip -> Adam -- and ``GraphedTrainStep``'s
    two-graph form of it (graph A ... gradient gather | eager all-reduces | graph B: global guard, sums, clip + Adam) had never
    executed on RCCL: world 1 skips them.  ``force_collective=True`` runs them on a ONE-rank RCCL group, where the sum is the
    identity: parameters, moments, loss sums and step counts must equal the world-1 path's BIT FOR BIT over three steps, eager
    and graphed; a non-finite loss raises with everything untouched, and training goes on."""
    import torch.distributed as dist
    from wcmc_amd import KPCN
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.optim import FusedClipAdam
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    from wcmc_amd.synthetic import make_batch
    own_group = False                                   # (the session's group: conftest.rccl_one_rank_group)
    try:
        assert dist.get_backend() == "nccl"
        results = {}
        for coll in (False, True):
            # (False: eager | True: one forked graph A | "overlap": the cut backward | "halves": what the launcher builds on several ranks --
            # the two half-step graphs + gather graph in front of the all-reduces, graph B behind them, the flags read one step late)
            for graphed in (False, True, "overlap", "halves"):
                if graphed in ("overlap", "halves") and not coll:
                    continue
                torch.manual_seed(21)
                kw = dict(ksize=21, depth=3, width=24)
                models = {"dncnn": KPCN(39, **kw), "backbone_diffuse": PathNet(36, intermc=16),
                          "backbone_specular": PathNet(36, intermc=16)}
                for m in models.values():
                    m.to(DEV)
                optims = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-3) for k, m in models.items()}
                lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
                      "l_test": RelativeMSE(), "l_manif": FeatureMSE(non_local=True, rng="cpu")}
                itf = KPCNInterface(models, optims, lf, types.SimpleNamespace(model_name="g"), use_llpm_buf=True,
                                    manif_learn=True, w_manif=0.1, train_branches=True)
                # "overlap": the backward cut at the P-buffers, the dncnn bucket's all-reduce issued while the PathNets' backward
                # (a third graph) runs -- SURVEY 8e's overlap, again on the one-rank group
                fo = FusedClipAdam(models, optims, process_group=dist.group.WORLD if coll else None, force_collective=coll,
                                   order=("dncnn", "backbone_diffuse", "backbone_specular") if graphed == "overlap" else None)
                assert fo.collective == coll and fo.world == 1
                itf.fused_optim = fo
                itf.iters = 1
                itf.to_train_mode()
                batches = [make_batch(2, 4, 48, seed=30 + i, device=DEV) for i in range(3)]
                if graphed:
                    step = GraphedTrainStep(itf, batches[0], overlap_allreduce=(graphed == "overlap"), two_stream=(graphed == "halves"),
                                            defer_check=(graphed == "halves"))
                    assert step.tail_split == coll and step.tail_captured == (not coll) and step.overlap == (graphed == "overlap")
                    assert step.defer_check == (graphed == "halves")
                else:
                    def step(b):
                        itf.preprocess(b)
                        itf.train_batch(b)
                torch.manual_seed(22)
                for b in batches:
                    step(b)
                if graphed:
                    step.flush()
                state = lambda: torch.cat([torch.cat([fo.flats[n].flat, fo.flats[n].m, fo.flats[n].v]) for n in sorted(fo.flats)]).clone()
                results[(coll, graphed)] = (state(), {k: v.item() for k, v in itf.m_losses.items()}, [fl.steps for fl in fo.flats.values()])
                if coll:
                    # the rank-global guard through the flag slot of the first bucket: nothing moves, the reference's error is raised
                    before, sums = state(), {k: v.item() for k, v in itf.m_losses.items()}
                    bad = {k: v.clone() for k, v in batches[0].items() if isinstance(v, torch.Tensor)}
                    bad["target_total"][0, 0, 20, 20] = float("inf")        # (enters l_total and rmse only: FeatureMSE has its own check)
                    with pytest.raises(RuntimeError, match="Non-finite loss at train time"):
                        step(bad)
                        if graphed:
                            step.flush()                                # (deferred check: the error is raised by the flush)
                    torch.cuda.synchronize()
                    assert torch.equal(state(), before) and [fl.steps for fl in fo.flats.values()] == [3, 3, 3]
                    assert {k: v.item() for k, v in itf.m_losses.items()} == sums
                    step(batches[1])
                    if graphed:
                        step.flush()
                    assert [fl.steps for fl in fo.flats.values()] == [4, 4, 4] and not torch.equal(state(), before)
                if graphed:
                    step.close()                                        # one graphed step alive at a time
                del step, itf, fo, models, optims
        ref = results[(False, False)]
        for key, got in results.items():
            assert got[2] == [3, 3, 3], key
            assert torch.equal(got[0], ref[0]), "parameters / moments differ: %s" % (key,)
            for k in ref[1]:
                np.testing.assert_allclose(got[1][k], ref[1][k], rtol=1e-6, err_msg="%s %s" % (key, k))
    finally:
        if own_group:
            dist.destroy_process_group()


def test_nonfinite_loss_raises_and_skips_the_update():
    """interfaces.py:254-257: a non-finite loss raises RuntimeError; with the fused optimiser the raise comes
    after the (guarded, hence skipped) update has been enqueued -- parameters must be untouched."""
    from wcmc_amd import KPCN
    from wcmc_amd.optim import FusedClipAdam
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import RelativeMSE
    from wcmc_amd.synthetic import make_batch
    torch.manual_seed(0)
    models = {"dncnn": KPCN(34, ksize=5, depth=2, width=8).to(DEV)}
    optims = {"optim_dncnn": torch.optim.Adam(models["dncnn"].parameters(), lr=1e-3)}
    lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
          "l_test": RelativeMSE()}
    itf = KPCNInterface(models, optims, lf, types.SimpleNamespace(model_name="n"))
    itf.fused_optim = FusedClipAdam(models, optims)
    itf.to_train_mode()
    batch = make_batch(1, 2, 32, seed=1, device=DEV, use_llpm=False)
    itf.preprocess(batch)
    itf.train_batch(batch)                                   # a healthy step first
    before = torch.cat([p.detach().reshape(-1).clone() for p in models["dncnn"].parameters()])
    bad = dict(batch)
    bad["target_diffuse"] = batch["target_diffuse"].clone()
    bad["target_diffuse"][0, 0, 16, 16] = float("nan")
    itf.preprocess(bad)
    with pytest.raises(RuntimeError, match="l_diffuse: Non-finite loss at train time."):
        itf.train_batch(bad)
    after = torch.cat([p.detach().reshape(-1) for p in models["dncnn"].parameters()])
    assert torch.equal(before, after)
    # the reference never reaches optim.step() on a non-finite loss: Adam's step counter stays where it was
    p0 = next(models["dncnn"].parameters())
    assert float(optims["optim_dncnn"].state[p0]["step"]) == 1.0 and itf.fused_optim.flats["dncnn"].steps == 1
    itf.preprocess(batch)
    itf.train_batch(batch)                                   # and training can go on
    assert float(optims["optim_dncnn"].state[p0]["step"]) == 2.0


# ---------------------------------------------------------------------------------------------- A/B switch matrix
# Every behaviour switch of the library against the default configuration on one full KPCN-Manifold step (default-width
# PathNets so that the persistent 1x1 kernel and its fused pairs engage; 100-wide KPCN so that the 8x16 halo tiling and the
# filter-row weight gradient engage).  "exact": the alternative path must reproduce the default bit for bit (same MFMA
# sequence / same order of additions per output); "close": another K order, split-K order or grouping of the bias sums,
# held to 1.5e-3 relative L2 (measured: <= 6.9e-4 with round 5's weights -- WCMC_IGEMM_HALO=0 on the KPCN input layer, at the far end
# of the backward pass; <= 1.3e-4 with round 4's).
_SWITCHES = [
    # the switches the release library and ops.py still have (wcmc_amd/ops.py: WCMC_BRANCH_STREAM, WCMC_SIDE_STREAM; the two PathNet
    # fusions have their own tests in tests/test_gpu_ops.py; WCMC_JOINT_BACKWARD: support/interfaces.py)
    ("attr", "USE_BRANCH_STREAM", False, "exact"),       # specular half on the main stream
    ("attr", "USE_SIDE_STREAM", True, "exact"),          # weight gradients on a forked stream (off in every mode since round 4)
    ("env", "WCMC_JOINT_BACKWARD", "0", "exact"),        # the reference's one autograd engine run per branch loss
] + ([
    # kernel A/B switches: DEBUG build of the library only (conftest.needs_debug_lib)
    ("env", "WCMC_IGEMM_PW", "0", "close"),              # tiled kernel for the 1x1 layers (bias sums group per tile)
    ("env", "WCMC_PW_TAIL", "0", "exact"),               # no fused 1x1 layer pairs
    ("env", "WCMC_KA_TILE", "1", "exact"),               # tile kernel-apply instead of the strip kernel
    ("env", "WCMC_HALO_TH8_5X5", "0", "close"),          # 16x16 tiles, 56/48-channel slabs (another K order)
    ("env", "WCMC_HALO_NB", "2", "exact"),               # two weight stages in the 16x16 halo igemm
    ("env", "WCMC_IGEMM_DBUF", "0", "exact"),            # single-buffered streaming igemm
    ("env", "WCMC_IGEMM_HALO", "0", "close"),            # streaming igemm for the 3x3 / 5x5 layers (another K order)
    ("env", "WCMC_WGRAD_ROWS", "0", "close"),            # one-tap weight-gradient kernel (another split-K order)
    ("env", "WCMC_HALO64", "0", "close"),                # the 8x16 5x5 kernel (32-channel slabs: another K order)
    ("env", "WCMC_HALO64_PT3", "0", "close"),            # 16x16 tiles only (bias sums group per tile)
    ("env", "WCMC_HALO64_CS32", "0", "close"),           # 16-channel slabs for the 441-channel data gradient too (another K order)
    ("env", "WCMC_WGRAD_ROWS_3X3", "0", "close"),        # filter-row weight gradient only from 256 input channels up
    ("env", "WCMC_WGRAD_ROWS_1X1", "0", "close"),        # one-tap kernel for the 128->128 1x1 weight gradient
    ("env", "WCMC_WGRAD_ROWS8", "1", "exact"),           # eight-wave filter-row kernel for the 100 -> 100 5x5 layers (same slabs; the
                                                         # one-plane launches of the default mode run the seven-wave one)
    ("env", "WCMC_WGRAD_ROWS8_PRIO", "0", "exact"),      # no priority hand-over between the two waves of a SIMD
    ("env", "WCMC_WGRAD_ROWS8_XE", "0", "exact"),        # left-over tiles as three pairs x one cout tile per wave
] if DEBUG_LIB else [])


def test_default_mode_forward_is_the_three_term_forward_and_its_gradients_stay_close():
    """Round 3's default ("bf16x321") changes the BACKWARD GEMMs only: loss scalars and denoised patches equal the all-three-term
    mode's bit for bit; the gradients differ by the bf16 rounding of dy / x (2 x 4 x 64 x 64 patches here: little to average over);
    and WCMC_DGRAD_AP1=0 (three-term data gradients, one-term weight gradients) lies between the two.  The default mode
    ("bf16x321h"; and the opt-in "bf16x321o") differs from it in the forward of the two KPCN OUTPUT layers only (one fp16 / bf16
    MFMA per product: exact on the rounded operands, tests/test_gpu_ops.py::test_fp16_output_layer_forward_... /
    test_one_term_output_layer_forward_...): everything upstream of them -- the P-buffers, hence the manifold losses -- is still
    bit-identical, the denoised patches and the image losses move by less than 5e-5 (fp16) / a third of north_star's 1e-3 (bf16)
    (profiles/r04_forward_ladder.txt: 1.3e-5 / 1.1e-4 at the bench shape)."""
    import os
    from conftest import rel_l2
    from wcmc_amd import ops
    runs = {}
    old = ops.PRECISION
    try:
        for mode, env in (("bf16x3", None), ("bf16x321", None)) + ((("bf16x321", "0"),) if DEBUG_LIB else ()) + (("bf16x321o", None), ("bf16x321h", None)):
            ops.set_precision(mode)
            if env is not None:
                os.environ["WCMC_DGRAD_AP1"] = env
            try:
                runs[(mode, env)] = _switch_step_body()
            finally:
                os.environ.pop("WCMC_DGRAD_AP1", None)
    finally:
        ops.set_precision(old)
    base = runs[("bf16x3", None)]
    worst = {}
    for key, got in runs.items():
        for k, want in base.items():
            if key[0] in ("bf16x321o", "bf16x321h") and k.startswith(("loss/", "out/")) and "manif" not in k:
                e = ((got[k] - want).abs().max() / want.abs().max()).item()
                assert 0.0 < e <= (3e-4 if key[0] == "bf16x321o" else 5e-5), (key, k, e)
            elif k.startswith(("loss/", "out/")):
                assert torch.equal(got[k], want), (key, k)
            else:
                worst[key] = max(worst.get(key, 0.0), rel_l2(got[k], want))
    assert worst[("bf16x3", None)] == 0.0
    assert (not DEBUG_LIB or 0.0 < worst[("bf16x321", "0")] <= 6e-3) and 0.0 < worst[("bf16x321", None)] <= 6e-3, worst
    assert 0.0 < worst[("bf16x321o", None)] <= 1.2e-2, worst      # (measured 6.9e-3: the output layers' rounded logits move d_logits)
    assert 0.0 < worst[("bf16x321h", None)] <= 6e-3, worst


def _switch_step():
    from wcmc_amd import KPCN
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    from wcmc_amd.synthetic import make_batch
    from wcmc_amd import ops as _o
    old_mode = _o.PRECISION
    _o.set_precision("bf16x3")
    try:
        return _switch_step_body()
    finally:
        _o.set_precision(old_mode)


def _switch_step_body():
    from wcmc_amd import KPCN
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    from wcmc_amd.synthetic import make_batch
    torch.manual_seed(5)
    models = {"dncnn": KPCN(39, ksize=21, depth=3, width=100), "backbone_diffuse": PathNet(36), "backbone_specular": PathNet(36)}
    for m in models.values():
        randomize_bias(m, 6)
        m.to(DEV)
    optims = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-3) for k, m in models.items()}
    lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
          "l_test": RelativeMSE(), "l_manif": FeatureMSE(non_local=True, rng="cpu")}
    itf = KPCNInterface(models, optims, lf, types.SimpleNamespace(model_name="s"), use_llpm_buf=True, manif_learn=True,
                        w_manif=0.1, train_branches=True)
    itf.iters = 1
    itf.to_train_mode()
    batch = make_batch(2, 4, 64, seed=50, device=DEV)
    torch.manual_seed(51)
    loss = itf._forward_backward(batch)
    torch.cuda.synchronize()
    out = {"loss/" + k: v.detach().clone().reshape(1) for k, v in loss.items()}
    out.update({"out/" + k: v.clone() for k, v in itf.last_out.items()})
    for mn, m in models.items():
        for k, p in m.named_parameters():
            out["grad/%s/%s" % (mn, k)] = p.grad.detach().clone()
    return out


@pytest.fixture(scope="module")
def switch_baseline():
    return _switch_step()


@pytest.mark.parametrize("kind,name,value,how", _SWITCHES, ids=[s[1] for s in _SWITCHES])
def test_switch_matrix_against_default_step(switch_baseline, kind, name, value, how, monkeypatch):
    """(Run on the three-term arithmetic, where a kernel switch changes the kernel and nothing else: in the default mode
    WCMC_IGEMM_HALO=0 / WCMC_HALO64=0 also take the two-term data-gradient instances away, i.e. change what is computed.)"""
    from wcmc_amd import ops
    if kind == "attr":
        monkeypatch.setattr(ops, name, value)
    else:
        monkeypatch.setenv(name, value)
    got = _switch_step()
    assert got.keys() == switch_baseline.keys()
    for k, want in switch_baseline.items():
        if how == "exact":
            assert torch.equal(got[k], want), "%s=%s changes %s (max |diff| %.3e)" % (
                name, value, k, (got[k] - want).abs().max().item())
        else:
            e = rel_l2(got[k], want)
            assert e <= 1.5e-3, "%s=%s: %s relative L2 %.3e" % (name, value, k, e)
