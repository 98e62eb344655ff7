"""BASELINE configs[3] (LBMC-Manifold) and configs[4] (SBMC-Manifold) at FULL size on the MI355X: the part of them the
reference owns -- ``SBMCInterface`` / ``LBMCInterface`` (``support/interfaces.py:336-523,753-839``) with the single
``PathNet`` backbone at S = 8 (``support/networks.py``), ``FeatureMSE`` on the 5-D per-sample P-buffer
(``support/losses.py:82-113``), the per-sample feature assembly (``wcmc_sample_cat_fwd``) and the reconstruction losses
of ``train_sbmc.py`` / ``train_lbmc.py`` -- 128x128 patches, 8 spp, batch 8, around the stand-in for the external base
denoisers (``sbmc.Multisteps``, layerdenoise's ``LayerNet``: absent from the reference tree), against
``oracle.step.sample_train_step`` (pinned by the goldens of the REAL classes, tests/test_oracle_golden.py) on the same
weights, inputs and pairings.  Loss scalars and the denoised output at 1e-3 (north star); gradients after the norm
clamp by relative L2; parameters after the Adam step."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from conftest import assert_grad_close                      # noqa: E402
from oracle import losses as ol                              # noqa: E402
from oracle import step as ostep                             # noqa: E402
from oracle.models import SampleDenoiserStandIn as OStandIn  # noqa: E402
from oracle.networks import PathNet as OPathNet              # noqa: E402

DEV = "cuda"
CASES = {
    # name: (interface, disentangle, pnet_out, recon loss, per-sample features, clip norm)
    "configs4_sbmc_manifold": ("SBMCInterface", "m11r11", 3, "TonemappedRelativeMSE", 7, 1000),     # train_sbmc.py:125-135
    "configs3_lbmc_manifold": ("LBMCInterface", "m11r01", 4, "SMAPE", 5, 250),                       # train_lbmc.py:129-139
}


@pytest.mark.parametrize("case", list(CASES))
def test_sample_based_manifold_step_at_full_size_against_oracle(case):
    from standins import SampleDenoiserStandIn
    from wcmc_amd import ops
    from wcmc_amd.support import interfaces as itf_mod
    from wcmc_amd.support import losses as pl
    from wcmc_amd.support.networks import PathNet
    assert ops.PRECISION == ops.MODES[0]
    kind, option, pout, recon, nfeat, clip = CASES[case]
    B, S, H, WIDTH, DEPTH = 8, 8, 128, 8, 2
    c_r = (pout // 2 if option in ("m10r01", "m11r01") else pout) + 1
    torch.manual_seed(31)
    omods = {"dncnn": OStandIn(nfeat + c_r, width=WIDTH, depth=DEPTH), "backbone": OPathNet(36, outc=pout)}
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for m in omods.values():
            for n, p in m.named_parameters():
                if n.endswith("bias"):
                    p.copy_(torch.rand(p.shape, generator=g) * 0.2 - 0.1)
    hmods = {"dncnn": SampleDenoiserStandIn(nfeat + c_r, width=WIDTH, depth=DEPTH), "backbone": PathNet(36, outc=pout)}
    for k in omods:
        hmods[k].load_state_dict(omods[k].state_dict())
        hmods[k].to(DEV)
    start = {mn: {k: v.detach().clone() for k, v in m.named_parameters()} for mn, m in omods.items()}
    lr = {"dncnn": 1e-4, "backbone": 1e-4}
    oopt = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=lr[k]) for k, m in omods.items()}
    hopt = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=lr[k]) for k, m in hmods.items()}
    # the sample-based batch (interfaces.py:360-366): noisy per-sample radiance, per-sample features, the KPCN bench's paths
    from wcmc_amd.synthetic import make_batch
    gb = torch.Generator().manual_seed(33)
    kb = make_batch(B, S, H, seed=34, device="cpu")
    target = kb["target_total"].clamp_min(0)
    batch = {"target_image": target,
             "radiance": (target.unsqueeze(1) * (0.5 + torch.rand(B, S, 3, H, H, generator=gb))).contiguous(),
             "features": torch.rand(B, S, nfeat, H, H, generator=gb) - 0.3,
             "paths": kb["paths"]}
    hc = H - 2 * DEPTH                                       # the stand-in's valid 3x3 convs crop the P-buffer too
    torch.manual_seed(35)
    perms = ostep.draw_perms(B, S, hc, hc)
    cfg = dict(use_llpm_buf=True, manif_learn=True, w_manif=0.1, disentangle=option,
               recon=getattr(ol, recon)(), clip_norm=clip)
    loss_o, out_o, pb_o, _ = ostep.sample_train_step(omods, oopt, batch, cfg, perms)

    lf = {"l_recon": getattr(pl, recon)(), "l_test": pl.RelativeMSE(), "l_manif": pl.FeatureMSE(non_local=True)}
    itf = getattr(itf_mod, kind)(hmods, hopt, lf, types.SimpleNamespace(model_name=case), use_llpm_buf=True,
                                 manif_learn=True, w_manif=0.1, disentangle=option)
    itf.iters = 1
    itf.to_train_mode()
    dbatch = {k: v.to(DEV) for k, v in batch.items()}
    torch.manual_seed(35)                                    # FeatureMSE draws the same two permutations (losses.py:35,50)
    itf.preprocess(dbatch)
    itf.train_batch(dbatch)
    torch.cuda.synchronize()
    assert torch.equal(lf["l_manif"].last_perms[0].cpu(), perms[0]) and torch.equal(lf["l_manif"].last_perms[1].cpu(), perms[1])
    for k, v in loss_o.items():
        np.testing.assert_allclose(itf.m_losses["m_" + k].item(), v.item(), rtol=1e-3, err_msg=k)
    for mn in omods:
        for (k, p), (_, q) in zip(hmods[mn].named_parameters(), omods[mn].named_parameters()):
            assert_grad_close(p.grad, q.grad, what="%s grad %s %s" % (case, mn, k), l2=2e-3, cos=2e-6)
            d_h, d_o = p.detach().cpu() - start[mn][k], q.detach() - start[mn][k]
            well = q.grad.abs() >= 0.5 * q.grad.pow(2).mean().sqrt()
            if bool(well.any()):
                assert float((d_h - d_o)[well].abs().max()) <= 0.05 * lr[mn], (mn, k, "Adam update differs on a well-conditioned entry")
            assert float(d_h.abs().max()) > 0.5 * lr[mn], (mn, k, "parameters did not move")
    itf.to_eval_mode()
    with torch.no_grad():
        for m in omods.values():
            m.eval()
        out_h, pb_h = itf.validate_batch(dbatch)
        # the oracle's validation forward with the UPDATED weights of its own step (interfaces.py:466-499)
        pbo = omods["backbone"](batch)
        if option in ("m10r01", "m11r01"):
            pbo = pbo[:, :, :pbo.shape[2] // 2]
        pv = torch.stack([pbo.var(1).mean(1, keepdim=True) / S] * S, 1)
        out_v = omods["dncnn"]({"radiance": batch["radiance"], "features": torch.cat([batch["features"], pbo, pv], 2)})
    e = ((out_h.cpu().double() - out_v.double()).abs().max() / out_v.double().abs().max()).item()
    assert e <= 2e-3, "validate output %.3e" % e            # (two networks one Adam step apart from equal starts)
    e = ((pb_h.cpu().double() - pbo.double()).abs().max() / pbo.double().abs().max()).item()
    assert e <= 2e-3, "validate p_buffer %.3e" % e
