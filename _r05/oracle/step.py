"""Oracle restatement of one ``KPCNInterface`` step (``support/interfaces.py:108-318``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Functional form of
``preprocess`` + ``train_batch`` (and ``validate_batch``) so that the same code
drives (i) the parity tests, (ii) golden G5 (checked against the real reference
``KPCNInterface`` in the build container, ``tests/golden/make_golden.py``) and
(iii) the ``cpu_baseline`` leg of ``bench.py``.

The manifold-loss permutations are explicit (``perms``), drawn by the caller in
the reference's order: diffuse(patch, batch) then specular(patch, batch)
(``interfaces.py:225-232`` -> ``losses.py:105-109``).
"""
import torch
import torch.nn as nn

from .losses import RelativeMSE, feature_mse
from .utils import crop_like

OPTIONS = ("m11r11", "m10r01", "m11r01", "m10r11")


def split_pbuffers(p, option, train):
    """Feature disentanglement (train ``interfaces.py:139-163``; val ``:284-291``).

    Returns (manifold-loss view, regression view) of one (B,S,C,H,W) P-buffer.
    """
    c = p.shape[2]
    assert c >= 2 and option in OPTIONS
    lo, hi = p[:, :, :c // 2], p[:, :, c // 2:]
    if not train:
        return None, (lo if option in ("m10r01", "m11r01") else p)
    if option == "m11r11":
        return p, p
    if option == "m10r01":
        return hi, lo
    if option == "m11r01":
        return p, lo
    return hi, p  # m10r11


def assemble_input(kpcn_in, p_regress):
    """``interfaces.py:165-176``: cat([in, mean_s P, var_s P .mean_c / S (detached)])."""
    s = p_regress.shape[1]
    p_var = p_regress.var(1).mean(1, keepdim=True).detach() / s
    return torch.cat([kpcn_in, p_regress.mean(1), p_var], 1)


def draw_perms(b, s, h, w, non_local=True):
    """One FeatureMSE call's draws, in reference order (``losses.py:35,50``)."""
    idx_patch = torch.randperm(s * h * w)
    idx_batch = torch.randperm(b * s * h * w) if non_local else None
    return idx_patch, idx_batch


def forward_losses(models, batch, cfg, perms=None, train=True):
    """Forward of one step; returns (out dict, p_regress dict, losses dict of graph tensors)."""
    use_llpm, manif = cfg.get("use_llpm_buf", False), cfg.get("manif_learn", False)
    option = cfg.get("disentanglement_option", "m11r11")
    out_manif, p_regress = None, None
    if use_llpm:
        out_manif, p_regress = {}, {}
        for br in ("diffuse", "specular"):
            p = models["backbone_" + br](batch)
            out_manif[br], p_regress[br] = split_pbuffers(p, option, train)
        batch = dict(batch)
        for br in ("diffuse", "specular"):
            batch["kpcn_%s_in" % br] = assemble_input(batch["kpcn_%s_in" % br], p_regress[br])
    out = models["dncnn"](batch)
    total, diffuse, specular = out["radiance"], out["diffuse"], out["specular"]
    l1 = nn.L1Loss()
    tgt_total = crop_like(batch["target_total"], total)
    losses = {}
    if not train:
        losses["val"] = RelativeMSE()(total, tgt_total)
        return out, p_regress, losses
    if cfg.get("train_branches", True):
        tgt = {"diffuse": crop_like(batch["target_diffuse"], diffuse),
               "specular": crop_like(batch["target_specular"], specular)}
        pred = {"diffuse": diffuse, "specular": specular}
        for i, br in enumerate(("diffuse", "specular")):
            base = l1(pred[br], tgt[br])
            full = base
            if manif:
                pb = crop_like(out_manif[br], pred[br])
                ip, ib = perms[i]
                lm = feature_mse(pb, tgt[br], ip, ib)
                losses["l_manif_" + br] = lm.detach()
                full = base + lm * cfg.get("w_manif", 0.1)
            # Reference quirk: loss_dict['l_diffuse'] = L_diffuse.detach() shares storage with
            # L_diffuse, and `L_diffuse += L_manif * w` (interfaces.py:221,227) is in place, so the
            # LOGGED branch loss is L1 + w * manifold, not the bare L1.
            losses["l_" + br] = full.detach()
            losses["_L_" + br] = full
        with torch.no_grad():
            losses["l_total"] = l1(total, tgt_total)
    else:
        lt = l1(total, tgt_total)                     # interfaces.py:243-246 (no manifold term)
        losses["l_total"] = lt.detach()
        losses["_L_total"] = lt
    with torch.no_grad():
        losses["rmse"] = RelativeMSE()(total, tgt_total)
    return out, p_regress, losses


def train_step(models, optims, batch, cfg, perms=None, clip=1.0):
    """``preprocess`` + ``train_batch`` (``interfaces.py:108-192,206-271``).

    ``optims`` maps ``'optim_<model>'`` to a torch optimizer.  Returns the dict of
    detached loss scalars (keys of ``loss_dict`` at ``interfaces.py:221-249``).
    """
    for m in models.values():
        m.zero_grad()
    out, _, losses = forward_losses(models, batch, cfg, perms, train=True)
    for k in ("_L_diffuse", "_L_specular", "_L_total"):
        if k in losses:
            losses[k].backward()
    loss_dict = {k: v.detach() for k, v in losses.items() if not k.startswith("_")}
    for k, v in loss_dict.items():
        if not torch.isfinite(v).all():
            raise RuntimeError("%s: Non-finite loss at train time." % k)
    for name, m in models.items():
        nn.utils.clip_grad_value_(m.parameters(), clip_value=clip)
    for name in models:
        optims["optim_" + name].step()
    return loss_dict, out


def sample_train_step(models, optims, batch, cfg, perms=None):
    """One ``SBMCInterface`` / ``LBMCInterface`` step (``interfaces.py:360-464`` and ``:771-839``) around any base
    denoiser honouring their batch contract (``radiance`` (B,S,3,H,W), ``features`` (B,S,C,H,W) -> (B,3,H',W')).

    cfg: ``use_llpm_buf``, ``manif_learn``, ``w_manif``, ``disentangle``, ``recon`` (the ``l_recon`` module) and
    ``clip_norm`` (1000 for SBMC ``:455``, 250 for LBMC ``:826``).  perms = (idx_patch, idx_batch) of the one FeatureMSE call.
    Returns (loss_dict, out, p_buffer fed to the regression).  Pinned by tests/golden/interface_{sbmc,lbmc}_*.npz (the
    REAL classes): tests/test_oracle_golden.py::test_oracle_sample_step_against_reference_golden."""
    option = cfg.get("disentangle", "m11r11")
    assert option in OPTIONS
    for m in models.values():
        m.zero_grad()
    out_manif, p_buffer = None, None
    if cfg.get("use_llpm_buf", False):
        p_buffer = models["backbone"](batch)                     # (B,S,C,H,W)
        s, c = p_buffer.shape[1], p_buffer.shape[2]
        assert c >= 2
        if option == "m11r11":
            out_manif = p_buffer
        elif option == "m10r01":
            out_manif, p_buffer = p_buffer[:, :, c // 2:], p_buffer[:, :, :c // 2]
        elif option == "m11r01":
            out_manif, p_buffer = p_buffer, p_buffer[:, :, :c // 2]
        else:                                                     # m10r11
            out_manif = p_buffer[:, :, c // 2:]
        p_var = p_buffer.var(1).mean(1, keepdim=True) / s        # :394-396 (unbiased over spp, mean over channels)
        p_var = torch.stack([p_var] * s, 1).detach()
        batch = {"target_image": batch["target_image"], "radiance": batch["radiance"],
                 "features": torch.cat([batch["features"], p_buffer, p_var], 2)}
    out = models["dncnn"](batch)
    tgt = crop_like(batch["target_image"], out)
    loss_dict = {}
    total = cfg["recon"](out, tgt)
    if cfg.get("manif_learn", False):
        lm = feature_mse(crop_like(out_manif, out), tgt, perms[0], perms[1])
        loss_dict["l_manif"] = lm.detach()
        total = total + lm * cfg.get("w_manif", 0.1)
        # reference quirk (:432-434): l_recon = L_total.detach() aliases the tensor `L_total += L_manif * w` updates in place
        loss_dict["l_recon"] = total.detach()
    loss_dict["l_total"] = total.detach()
    total.backward()
    with torch.no_grad():
        loss_dict["rmse"] = RelativeMSE()(out, tgt)
    for k, v in loss_dict.items():
        if not torch.isfinite(v).all():
            raise RuntimeError("%s: Non-finite loss at train time." % k)
    norms = {}
    for name, m in models.items():
        norms[name] = nn.utils.clip_grad_norm_(m.parameters(), max_norm=cfg["clip_norm"])
    for name in models:
        optims["optim_" + name].step()
    return loss_dict, out, p_buffer, norms
