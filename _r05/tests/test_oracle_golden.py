"""The CPU oracle against the golden vectors generated from the real reference
(tests/golden/make_golden.py).  No GPU."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import losses as ol
from oracle import step as ostep
from oracle.utils import crop_like

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as mg  # noqa: E402  (geometry + model builders only; the reference is not imported)


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_crop_like_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "crop_like.npz"))
    for i in range(int(d["n"])):
        ss, ts = tuple(d["src_shape_%d" % i]), tuple(d["tgt_shape_%d" % i])
        src = torch.arange(int(np.prod(ss)), dtype=torch.float32).view(ss)
        out = crop_like(src, torch.zeros(ts))
        assert out.shape == d["out_%d" % i].shape
        assert np.array_equal(out.numpy(), d["out_%d" % i])
        assert out.data_ptr() >= src.data_ptr()  # still a view of src


def test_feature_mse_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "losses_fmse.npz"))
    for i in range(int(d["n"])):
        p = T(d["p_%d" % i]).requires_grad_(True)
        ref = T(d["ref_%d" % i])
        ib = T(d["idx_batch_%d" % i]) if bool(d["non_local_%d" % i]) else None
        loss = ol.feature_mse(p, ref, T(d["idx_patch_%d" % i]), ib)
        loss.backward()
        np.testing.assert_allclose(loss.item(), d["loss_%d" % i], rtol=1e-6)
        np.testing.assert_allclose(p.grad.numpy(), d["grad_%d" % i], rtol=1e-5, atol=1e-9)


def test_feature_mse_module_draws_like_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "losses_fmse.npz"))
    for i in range(int(d["n"])):
        p, ref = T(d["p_%d" % i]), T(d["ref_%d" % i])
        torch.manual_seed(int(d["seed_%d" % i]))
        m = ol.FeatureMSE(non_local=bool(d["non_local_%d" % i]))
        loss = m(p, ref)
        assert np.array_equal(m.last_perms[0].numpy(), d["idx_patch_%d" % i])
        np.testing.assert_allclose(loss.item(), d["loss_%d" % i], rtol=1e-6)


def test_grs_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "losses_grs.npz"))
    for i in range(int(d["n"])):
        p = T(d["p_%d" % i]).requires_grad_(True)
        loss = ol.global_relative_similarity(p, T(d["ref_%d" % i]), T(d["idx_patch_%d" % i]),
                                             T(d["idx_batch_%d" % i]))
        loss.backward()
        np.testing.assert_allclose(loss.item(), d["loss_%d" % i], rtol=1e-6)
        np.testing.assert_allclose(p.grad.numpy(), d["grad_%d" % i], rtol=1e-5, atol=1e-9)


def test_image_losses_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "losses_image.npz"))
    ref = T(d["ref"])
    table = {"RelativeMSE": ol.RelativeMSE(), "SMAPE": ol.SMAPE(), "TonemappedMSE": ol.TonemappedMSE(),
             "TonemappedRelativeMSE": ol.TonemappedRelativeMSE(), "L1": torch.nn.L1Loss()}
    for name, fn in table.items():
        x = T(d["im"]).requires_grad_(True)
        loss = fn(x, ref)
        loss.backward()
        np.testing.assert_allclose(loss.item(), d[name], rtol=1e-6)
        np.testing.assert_allclose(x.grad.numpy(), d[name + "_grad"], rtol=1e-5, atol=1e-9)


def load_case(golden_dir, case):
    d = np.load(os.path.join(golden_dir, "interface_%s.npz" % case))
    use_llpm, manif, tb, option, pout = mg.INTERFACE_CASES[case]
    models = mg.build_models(case, 0)
    for mn, m in models.items():
        sd = {k[len("init/%s/" % mn):]: T(d[k]) for k in d.files if k.startswith("init/%s/" % mn)}
        m.load_state_dict(sd)
    batch = {k[len("batch/"):]: T(d[k]) for k in d.files if k.startswith("batch/")}
    perms = None
    if manif and tb:
        perms = [(T(d["perm/%s_patch" % br]), T(d["perm/%s_batch" % br])) for br in ("diffuse", "specular")]
    cfg = dict(use_llpm_buf=use_llpm, manif_learn=manif, train_branches=tb, disentanglement_option=option,
               w_manif=0.1)
    return d, models, batch, perms, cfg


@pytest.mark.parametrize("case", list(mg.INTERFACE_CASES))
def test_interface_step_golden(golden_dir, case):
    """oracle.step (functional restatement of KPCNInterface) vs the real reference interface."""
    d, models, batch, perms, cfg = load_case(golden_dir, case)
    optims = {"optim_" + mn: torch.optim.Adam(m.parameters(), lr=1e-3 if mn == "dncnn" else 2e-3)
              for mn, m in models.items()}
    loss_dict, _ = ostep.train_step(models, optims, batch, cfg, perms)
    for k in d.files:
        if k.startswith("m_losses/") and k != "m_losses/m_val":
            np.testing.assert_allclose(loss_dict[k[len("m_losses/m_"):]].item(), d[k], rtol=2e-5, err_msg=k)
    for mn, m in models.items():
        for k, p in m.named_parameters():
            np.testing.assert_allclose(p.grad.numpy(), d["grad/%s/%s" % (mn, k)], rtol=1e-4, atol=1e-7,
                                       err_msg="grad %s %s" % (mn, k))
        for k, v in m.state_dict().items():
            # Adam's first step is lr*g/(|g|+1e-8): ill-conditioned where |g| ~ eps, so those
            # entries are only checked to be within one lr of the reference.
            g = np.abs(d["grad/%s/%s" % (mn, k)])
            want, got = d["after/%s/%s" % (mn, k)], v.numpy()
            big = g > 1e-5
            np.testing.assert_allclose(got[big], want[big], rtol=1e-4, atol=2e-6, err_msg="after %s %s" % (mn, k))
            np.testing.assert_allclose(got[~big], want[~big], atol=4.1e-3, err_msg="after %s %s" % (mn, k))
    with torch.no_grad():
        out, p_regress, losses = ostep.forward_losses(models, batch, cfg, train=False)
    np.testing.assert_allclose(out["radiance"].numpy(), d["val/radiance"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(losses["val"].item() / 2, d["val/summary"], rtol=1e-5)
    if p_regress is not None:
        np.testing.assert_allclose(p_regress["diffuse"].numpy(), d["val/p_diffuse"], rtol=1e-4, atol=1e-6)


def test_preprocess_golden(golden_dir):
    """G6: oracle/datasets.py == DenoiseDataset._preprocess_llpm / _preprocess_kpcn / _gradients (bit-exact:
    the same numpy operations in the same order)."""
    from oracle import datasets as od
    d = np.load(os.path.join(golden_dir, "preprocess.npz"))
    for name in ("a", "b", "zero_depth"):
        raw = d[name + "/raw"]
        np.testing.assert_array_equal(od.preprocess_llpm(raw), d[name + "/llpm"])
        np.testing.assert_array_equal(od.preprocess_kpcn(raw), d[name + "/kpcn"])
    np.testing.assert_array_equal(od.gradients(d["grad/buf"]), d["grad/out"])
    assert d["a/llpm"].shape[-1] == 37 and d["a/kpcn"].shape[-1] == 44


def test_patch_loader_item_golden(golden_dir):
    """oracle.datasets.sample_patch_origins / assemble_kpcn_patch == the real DenoiseDataset.__getitem__
    (datasets.py:795-840,1026-1146) on the fixture image: same numpy draws, bit-identical items."""
    from oracle import datasets as od
    d = np.load(os.path.join(golden_dir, "patches.npz"))
    P = int(d["patch"])
    for tag in ("llpm", "vanilla"):
        np.random.seed(int(d["seed"]))
        origins = od.sample_patch_origins(d["prob"], int(d[tag + "/patches_per_image"]))
        n = len([k for k in d.files if k.startswith(tag + "/") and k.endswith("/target_total")])
        for i in range(n):
            item = od.assemble_kpcn_patch(d["kpcn"], d["llpm"] if tag == "llpm" else None, d["gt"], origins[i], P)
            keys = {k.split("/")[-1] for k in d.files if k.startswith("%s/%d/" % (tag, i))}
            assert keys == set(item)
            for k in keys:
                np.testing.assert_array_equal(item[k], d["%s/%d/%s" % (tag, i, k)], err_msg="%s %d %s" % (tag, i, k))


@pytest.mark.parametrize("case", list(mg.SAMPLE_CASES))
def test_oracle_sample_step_against_reference_golden(golden_dir, case):
    """``oracle.step.sample_train_step`` -- the restatement of ``SBMCInterface`` / ``LBMCInterface.train_batch``
    (``interfaces.py:360-464,771-839``) that the full-size configs[3] / configs[4] GPU test compares with -- against the
    fixtures of the REAL classes: logged losses, gradients after the norm clamp, gradient norms, parameters after Adam."""
    from oracle.models import SampleDenoiserStandIn
    from oracle.networks import PathNet
    d = np.load(os.path.join(golden_dir, "interface_%s.npz" % case))
    kind, use_llpm, manif, option, pout, recon, nfeat = mg.SAMPLE_CASES[case]
    G = mg.G7_GEOM
    c_r = ((pout // 2 if option in ("m10r01", "m11r01") else pout) + 1) if use_llpm else 0
    models = {"dncnn": SampleDenoiserStandIn(nfeat + c_r, width=G["WIDTH"], depth=G["DEPTH"])}
    if use_llpm:
        models["backbone"] = PathNet(36, intermc=G["INTERMC"], outc=pout)
    for mn, m in models.items():
        m.load_state_dict({k[len("init/%s/" % mn):]: T(d[k]) for k in d.files if k.startswith("init/%s/" % mn)})
    optims = {"optim_" + mn: torch.optim.Adam(m.parameters(), lr=1e-3 if mn == "dncnn" else 2e-3) for mn, m in models.items()}
    cfg = dict(use_llpm_buf=use_llpm, manif_learn=manif, w_manif=0.1, disentangle=option,
               recon=torch.nn.L1Loss() if recon == "L1Loss" else getattr(ol, recon)(),
               clip_norm=1000 if kind == "SBMCInterface" else 250)
    batch = {k[len("batch/"):]: T(d[k]) for k in d.files if k.startswith("batch/")}
    perms = (T(d["perm/patch"]), T(d["perm/batch"])) if manif else None
    loss, out, pb, norms = ostep.sample_train_step(models, optims, batch, cfg, perms)
    for k in d.files:
        if k.startswith("m_losses/") and k != "m_losses/m_val":
            np.testing.assert_allclose(loss[k[len("m_losses/m_"):]].item(), d[k], rtol=1e-5, err_msg=k)
    for mn, m in models.items():
        post = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))     # (the fixture holds the norm AFTER the clamp)
        np.testing.assert_allclose(post, d["gradnorm/" + mn], rtol=1e-4)
        assert float(norms[mn]) >= post * (1 - 1e-6)
        for k, p in m.named_parameters():
            np.testing.assert_allclose(p.grad.numpy(), d["grad/%s/%s" % (mn, k)], rtol=1e-4, atol=1e-7, err_msg="%s %s" % (mn, k))
        for k, v in m.state_dict().items():
            g = np.abs(d["grad/%s/%s" % (mn, k)])         # (Adam's first step is ill-conditioned where |g| ~ eps: see above)
            want, got = d["after/%s/%s" % (mn, k)], v.numpy()
            big = g > 1e-5
            np.testing.assert_allclose(got[big], want[big], rtol=1e-4, atol=1e-6, err_msg="after %s %s" % (mn, k))
            np.testing.assert_allclose(got[~big], want[~big], atol=4.1e-3, err_msg="after %s %s" % (mn, k))
