#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Run in the build container only (``/root/reference`` is not on the GPU box):

    python tests/golden/make_golden.py

It imports ``support.losses`` / ``support.utils`` / ``support.interfaces`` from
``/root/reference`` (with a stub for the unused ``kornia`` import,
``support/losses.py:2``) and stores inputs + expected outputs as ``.npz``.  Only
data is written; no reference source travels.

Fixtures
  crop_like.npz        G1  support/utils.py:24-42
  losses_fmse.npz      G2  support/losses.py:9-113   (FeatureMSE, fwd + dL/dP)
  losses_grs.npz       G3  support/losses.py:116-211 (GlobalRelativeSimilarityLoss)
  losses_image.npz     G4  support/losses.py:245-320 (RelativeMSE, SMAPE, Tonemapped*)
  interface_<case>.npz G5  support/interfaces.py:80-333 driven with the build's
                           oracle modules (``oracle/``) as stand-ins for ``sbmc``.
  patches.npz          G8  support/datasets.py:795-840,1026-1146 (DenoiseDataset.__getitem__: patch sampling, batch keys)
  interface_{sbmc,lbmc}_*.npz  G7  support/interfaces.py:336-523, 753-839 (SBMCInterface, LBMCInterface) around
                           ``oracle.models.SampleDenoiserStandIn`` for the external base denoisers.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def import_reference():
    sys.path.insert(0, REF)
    k = types.ModuleType("kornia")
    k.rgb_to_hls = lambda x: x
    sys.modules["kornia"] = k
    import matplotlib
    matplotlib.use("Agg")
    from support import interfaces as ref_itf
    from support import losses as ref_losses
    from support import utils as ref_utils
    return ref_losses, ref_utils, ref_itf


def np_(t):
    return t.detach().cpu().numpy().copy()   # copy: the tensor may be updated in place later


def gen_crop_like(ref_utils):
    out = {}
    cases = [((2, 3, 16, 16), (2, 3, 10, 10)), ((1, 2, 3, 15, 12), (1, 7, 8)),
             ((2, 3, 9, 9), (2, 3, 9, 9)), ((1, 1, 8, 8), (1, 1, 12, 5)),
             ((2, 4, 3, 128, 128), (2, 3, 92, 92)), ((1, 3, 11, 10), (1, 3, 4, 7))]
    for i, (ss, ts) in enumerate(cases):
        src = torch.arange(int(np.prod(ss)), dtype=torch.float32).view(ss)
        tgt = torch.zeros(ts)
        out["src_shape_%d" % i] = np.array(ss)
        out["tgt_shape_%d" % i] = np.array(ts)
        out["out_%d" % i] = np_(ref_utils.crop_like(src, tgt))
    out["n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "crop_like.npz"), **out)


def _loss_inputs(shape, seed):
    g = torch.Generator().manual_seed(seed)
    b, s, c, h, w = shape
    p = torch.rand(shape, generator=g) * 1.5            # PathNet output is >= 0 (ReLU)
    ref = torch.randn(b, 3, h, w, generator=g).exp() - 0.5   # includes negatives (clamped)
    return p, ref


def gen_fmse(ref_losses):
    out = {}
    cases = [((2, 4, 3, 12, 12), True), ((2, 8, 6, 10, 10), True), ((1, 2, 3, 8, 8), True),
             ((2, 4, 3, 12, 12), False), ((3, 2, 2, 5, 7), True)]
    for i, (shape, non_local) in enumerate(cases):
        p, ref = _loss_inputs(shape, 100 + i)
        p.requires_grad_(True)
        b, s, c, h, w = shape
        torch.manual_seed(1000 + i)
        idx_patch = torch.randperm(s * h * w)
        idx_batch = torch.randperm(b * s * h * w) if non_local else None
        torch.manual_seed(1000 + i)
        loss = ref_losses.FeatureMSE(non_local=non_local)(p, ref)
        loss.backward()
        out["p_%d" % i], out["ref_%d" % i] = np_(p), np_(ref)
        out["idx_patch_%d" % i] = np_(idx_patch)
        out["idx_batch_%d" % i] = np_(idx_batch) if non_local else np.zeros(0, np.int64)
        out["non_local_%d" % i] = np.array(non_local)
        out["seed_%d" % i] = np.array(1000 + i)
        out["loss_%d" % i] = np_(loss)
        out["grad_%d" % i] = np_(p.grad)
    out["n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "losses_fmse.npz"), **out)


def gen_grs(ref_losses):
    out = {}
    cases = [(2, 4, 3, 12, 12), (2, 8, 6, 10, 10), (1, 2, 3, 8, 8)]
    for i, shape in enumerate(cases):
        p, ref = _loss_inputs(shape, 200 + i)
        p.requires_grad_(True)
        b, s, c, h, w = shape
        torch.manual_seed(2000 + i)
        idx_patch = torch.randperm(s * h * w)
        idx_batch = torch.randperm(b * s * h * w)
        torch.manual_seed(2000 + i)
        loss = ref_losses.GlobalRelativeSimilarityLoss()(p, ref)
        loss.backward()
        out["p_%d" % i], out["ref_%d" % i] = np_(p), np_(ref)
        out["idx_patch_%d" % i], out["idx_batch_%d" % i] = np_(idx_patch), np_(idx_batch)
        out["loss_%d" % i], out["grad_%d" % i] = np_(loss), np_(p.grad)
    out["n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "losses_grs.npz"), **out)


def gen_image_losses(ref_losses):
    g = torch.Generator().manual_seed(7)
    im = torch.randn(2, 3, 16, 16, generator=g).exp() - 0.7
    ref = torch.randn(2, 3, 16, 16, generator=g).exp() - 0.7
    ref[0, 0, :2] = 0.0
    out = {"im": np_(im), "ref": np_(ref)}
    for name in ("RelativeMSE", "SMAPE", "TonemappedMSE", "TonemappedRelativeMSE"):
        x = im.clone().requires_grad_(True)
        loss = getattr(ref_losses, name)()(x, ref)
        loss.backward()
        out[name], out[name + "_grad"] = np_(loss), np_(x.grad)
    x = im.clone().requires_grad_(True)
    loss = torch.nn.L1Loss()(x, ref)
    loss.backward()
    out["L1"], out["L1_grad"] = np_(loss), np_(x.grad)
    np.savez_compressed(os.path.join(HERE, "losses_image.npz"), **out)


# ---------------------------------------------------------------------------------- G5
INTERFACE_CASES = {
    # name: (use_llpm, manif_learn, train_branches, option, pnet_out)
    "vanilla": (False, False, True, "m11r11", 0),
    "path_only": (True, False, True, "m11r11", 3),
    "manifold_m11r11": (True, True, True, "m11r11", 3),
    "manifold_m10r01": (True, True, True, "m10r01", 6),
    "manifold_m10r11": (True, True, True, "m10r11", 4),
    "manifold_m11r01": (True, True, True, "m11r01", 4),
    "post_train": (True, True, False, "m11r11", 3),
}
G5_GEOM = dict(B=2, S=2, H=20, KS=5, DEPTH=3, WIDTH=8, INTERMC=4, BASE_IN=11)


def small_batch(case_seed, use_llpm):
    g = torch.Generator().manual_seed(case_seed)
    B, S, H = G5_GEOM["B"], G5_GEOM["S"], G5_GEOM["H"]
    n_in = G5_GEOM["BASE_IN"] + (1 if use_llpm else 0)
    r = lambda *s: torch.rand(*s, generator=g)
    batch = {
        "kpcn_diffuse_in": r(B, n_in, H, H) - 0.3,
        "kpcn_specular_in": r(B, n_in, H, H) - 0.3,
        "kpcn_diffuse_buffer": r(B, 3, H, H) * 2,
        "kpcn_specular_buffer": r(B, 3, H, H),
        "kpcn_albedo": r(B, 3, H, H) + 0.00316,
        "target_diffuse": r(B, 3, H, H) * 2,
        "target_specular": r(B, 3, H, H),
        "target_total": r(B, 3, H, H) * 3,
    }
    if use_llpm:
        batch["paths"] = r(B, S, 36, H, H) - 0.4
    return batch


def build_models(case, seed):
    from oracle.models import KPCN
    from oracle.networks import PathNet
    use_llpm, manif, tb, option, pout = INTERFACE_CASES[case]
    torch.manual_seed(seed)
    n_in = G5_GEOM["BASE_IN"]
    if use_llpm:
        c_r = pout // 2 if option in ("m10r01", "m11r01") else pout
        n_in = n_in + 1 + c_r + 1
    models = {"dncnn": KPCN(n_in, ksize=G5_GEOM["KS"], depth=G5_GEOM["DEPTH"], width=G5_GEOM["WIDTH"])}
    if use_llpm:
        models["backbone_diffuse"] = PathNet(36, intermc=G5_GEOM["INTERMC"], outc=pout)
        models["backbone_specular"] = PathNet(36, intermc=G5_GEOM["INTERMC"], outc=pout)
    # non-zero biases so that bias handling is exercised
    with torch.no_grad():
        for m in models.values():
            for n, p in m.named_parameters():
                if n.endswith("bias"):
                    p.uniform_(-0.1, 0.1)
                if n.endswith("weight_g"):      # weight-normalised PathNet layers (oracle/modules.py): g != ||v||, so that the
                    p.mul_(torch.empty_like(p).uniform_(0.7, 1.3))      # normalisation is exercised, not the identity it starts as
    return models


def gen_interface(ref_losses, ref_itf):
    from oracle.step import draw_perms
    for ci, case in enumerate(INTERFACE_CASES):
        use_llpm, manif, tb, option, pout = INTERFACE_CASES[case]
        models = build_models(case, 300 + ci)
        init_state = {"%s/%s" % (mn, k): np_(v) for mn, m in models.items()
                      for k, v in m.state_dict().items()}
        optims = {"optim_" + mn: torch.optim.Adam(m.parameters(), lr=1e-3 if mn == "dncnn" else 2e-3)
                  for mn, m in models.items()}
        loss_funcs = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(),
                      "l_recon": torch.nn.L1Loss(), "l_test": ref_losses.RelativeMSE()}
        if manif:
            loss_funcs["l_manif"] = ref_losses.FeatureMSE(non_local=True)
        args = types.SimpleNamespace(model_name="golden")
        itf = ref_itf.KPCNInterface(models, optims, loss_funcs, args, use_llpm_buf=use_llpm,
                                    manif_learn=manif, w_manif=0.1, train_branches=tb,
                                    disentanglement_option=option)
        itf.iters = 1            # skip the iters % 1000 == 1 PNG dump (interfaces.py:130-137)
        batch = small_batch(400 + ci, use_llpm)
        out = {"batch/" + k: np_(v) for k, v in batch.items()}
        out.update({"init/" + k: v for k, v in init_state.items()})

        B, S, H = G5_GEOM["B"], G5_GEOM["S"], G5_GEOM["H"]
        h_out = H - 4 * G5_GEOM["DEPTH"]      # DEPTH valid 5x5 convs
        seed = 500 + ci
        torch.manual_seed(seed)
        perms = [draw_perms(B, S, h_out, h_out), draw_perms(B, S, h_out, h_out)]
        for i, br in enumerate(("diffuse", "specular")):
            out["perm/%s_patch" % br] = np_(perms[i][0])
            out["perm/%s_batch" % br] = np_(perms[i][1])
        out["seed"] = np.array(seed)

        itf.to_train_mode()
        torch.manual_seed(seed)
        itf.preprocess(batch)
        itf.train_batch(batch)
        for k, v in itf.m_losses.items():
            out["m_losses/" + k] = np_(v)
        for mn, m in models.items():
            for k, v in m.state_dict().items():
                out["after/%s/%s" % (mn, k)] = np_(v)
            for k, p in m.named_parameters():
                out["grad/%s/%s" % (mn, k)] = np_(p.grad)     # post-clip grads
        itf.to_eval_mode()
        with torch.no_grad():
            rad, pb = itf.validate_batch(batch)
        out["val/radiance"] = np_(rad)
        if pb is not None:
            out["val/p_diffuse"], out["val/p_specular"] = np_(pb["diffuse"]), np_(pb["specular"])
        out["val/summary"] = np.array(itf.get_epoch_summary(mode="eval", norm=1))
        np.savez_compressed(os.path.join(HERE, "interface_%s.npz" % case), **out)
        print("G5", case, {k: float(v) for k, v in itf.m_losses.items()})


# ---------------------------------------------------------------------------------- G5b
VARIANT_CASES = {
    # name: (class, manif_learn, train_branches)
    "ref_vanilla": ("KPCNRefInterface", False, True),         # interfaces.py:526-585
    "pre_manifold": ("KPCNPreInterface", True, True),         # interfaces.py:588-750, PathNet pre-training
    "pre_regress": ("KPCNPreInterface", False, True),         # ... KPCN on frozen PathNets
}


def build_variant_models(case, seed):
    from oracle.models import KPCN
    from oracle.networks import PathNet
    kind, manif, tb = VARIANT_CASES[case]
    torch.manual_seed(seed)
    G = G5_GEOM
    if kind == "KPCNRefInterface":
        models = {"dncnn": KPCN(G["BASE_IN"] + 3, ksize=G["KS"], depth=G["DEPTH"], width=G["WIDTH"])}
    else:
        models = {"dncnn": KPCN(G["BASE_IN"] + 1 + 3 + 1, ksize=G["KS"], depth=G["DEPTH"], width=G["WIDTH"]),
                  "backbone_diffuse": PathNet(36, intermc=G["INTERMC"], outc=3),
                  "backbone_specular": PathNet(36, intermc=G["INTERMC"], outc=3)}
    with torch.no_grad():
        for m in models.values():
            for n, p in m.named_parameters():
                if n.endswith("bias"):
                    p.uniform_(-0.1, 0.1)
                if n.endswith("weight_g"):      # weight-normalised PathNet layers (oracle/modules.py): g != ||v||, so that the
                    p.mul_(torch.empty_like(p).uniform_(0.7, 1.3))      # normalisation is exercised, not the identity it starts as
    return models


def gen_interface_variants(ref_losses, ref_itf):
    from oracle.step import draw_perms
    for ci, case in enumerate(VARIANT_CASES):
        kind, manif, tb = VARIANT_CASES[case]
        models = build_variant_models(case, 600 + ci)
        init_state = {"%s/%s" % (mn, k): np_(v) for mn, m in models.items() for k, v in m.state_dict().items()}
        optims = {"optim_" + mn: torch.optim.Adam(m.parameters(), lr=1e-3 if mn == "dncnn" else 2e-3)
                  for mn, m in models.items()}
        loss_funcs = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(),
                      "l_recon": torch.nn.L1Loss(), "l_test": ref_losses.RelativeMSE()}
        if manif:
            loss_funcs["l_manif"] = ref_losses.FeatureMSE(non_local=True)
        args = types.SimpleNamespace(model_name="golden")
        if kind == "KPCNRefInterface":
            itf = ref_itf.KPCNRefInterface(models, optims, loss_funcs, args, train_branches=tb)
        else:
            itf = ref_itf.KPCNPreInterface(models, optims, loss_funcs, args, manif_learn=manif, w_manif=0.1,
                                           train_branches=tb)
        itf.iters = 1
        use_llpm = kind != "KPCNRefInterface"
        batch = small_batch(700 + ci, use_llpm)
        if not use_llpm:                       # the Ref interface takes the vanilla 11-channel inputs
            pass
        out = {"batch/" + k: np_(v) for k, v in batch.items()}
        out.update({"init/" + k: v for k, v in init_state.items()})
        B, S, H = G5_GEOM["B"], G5_GEOM["S"], G5_GEOM["H"]
        seed = 800 + ci
        torch.manual_seed(seed)
        perms = [draw_perms(B, S, H, H), draw_perms(B, S, H, H)]      # pre-training pairs FULL-size P-buffers
        for i, br in enumerate(("diffuse", "specular")):
            out["perm/%s_patch" % br] = np_(perms[i][0])
            out["perm/%s_batch" % br] = np_(perms[i][1])
        out["seed"] = np.array(seed)
        itf.to_train_mode()
        out["train_flags"] = np.array([int(m.training) for m in models.values()])
        torch.manual_seed(seed)
        itf.preprocess(batch)
        itf.train_batch(batch)
        for k, v in itf.m_losses.items():
            out["m_losses/" + k] = np_(v)
        for mn, m in models.items():
            for k, v in m.state_dict().items():
                out["after/%s/%s" % (mn, k)] = np_(v)
            for k, p in m.named_parameters():
                out["grad/%s/%s" % (mn, k)] = np_(p.grad) if p.grad is not None else np.zeros(0, np.float32)
        itf.to_eval_mode()
        with torch.no_grad():
            rad, pb = itf.validate_batch(batch)
        out["val/radiance"] = np_(rad)
        if pb is not None:
            out["val/p_diffuse"], out["val/p_specular"] = np_(pb["diffuse"]), np_(pb["specular"])
        out["val/summary"] = np.array(itf.get_epoch_summary(mode="eval", norm=1))
        np.savez_compressed(os.path.join(HERE, "interface_%s.npz" % case), **out)
        print("G5b", case, {k: float(v) for k, v in itf.m_losses.items()})


# ---------------------------------------------------------------------------------- G7
SAMPLE_CASES = {
    # name: (class, use_llpm, manif_learn, disentangle, pnet_out, recon loss, features)
    "sbmc_vanilla": ("SBMCInterface", False, False, "m11r11", 0, "TonemappedRelativeMSE", 7),     # train_sbmc.py:125-135
    "sbmc_manifold": ("SBMCInterface", True, True, "m11r11", 3, "TonemappedRelativeMSE", 7),
    "sbmc_m10r01": ("SBMCInterface", True, True, "m10r01", 6, "TonemappedRelativeMSE", 7),
    "lbmc_manifold": ("LBMCInterface", True, True, "m11r01", 4, "SMAPE", 5),                       # train_lbmc.py:129-139
    "lbmc_m10r11": ("LBMCInterface", True, True, "m10r11", 4, "SMAPE", 5),
    # features x 3000 under an L1 reconstruction loss: gradient norms in the thousands, so the 250 / 1000 norm clamps bite
    "lbmc_clipped": ("LBMCInterface", True, False, "m11r11", 3, "L1Loss", 5),
    "sbmc_clipped": ("SBMCInterface", False, False, "m11r11", 0, "L1Loss", 7),
}
SAMPLE_FEATURE_SCALE = {"lbmc_clipped": 3000.0, "sbmc_clipped": 20000.0}
G7_GEOM = dict(B=2, S=3, H=16, INTERMC=4, WIDTH=8, DEPTH=2)


def sample_batch(seed, use_llpm, nfeat):
    g = torch.Generator().manual_seed(seed)
    B, S, H = G7_GEOM["B"], G7_GEOM["S"], G7_GEOM["H"]
    r = lambda *s: torch.rand(*s, generator=g)
    batch = {"target_image": r(B, 3, H, H) * 3, "radiance": r(B, S, 3, H, H) * 3, "features": r(B, S, nfeat, H, H) - 0.3}
    if use_llpm:
        batch["paths"] = r(B, S, 36, H, H) - 0.4
    return batch


def build_sample_models(case, seed):
    from oracle.models import SampleDenoiserStandIn
    from oracle.networks import PathNet
    kind, use_llpm, manif, option, pout, recon, nfeat = SAMPLE_CASES[case]
    torch.manual_seed(seed)
    c_r = 0
    if use_llpm:
        c_r = (pout // 2 if option in ("m10r01", "m11r01") else pout) + 1
    models = {"dncnn": SampleDenoiserStandIn(nfeat + c_r, width=G7_GEOM["WIDTH"], depth=G7_GEOM["DEPTH"])}
    if use_llpm:
        models["backbone"] = PathNet(36, intermc=G7_GEOM["INTERMC"], outc=pout)
    with torch.no_grad():
        for m in models.values():
            for n, p in m.named_parameters():
                if n.endswith("bias"):
                    p.uniform_(-0.1, 0.1)
                if n.endswith("weight_g"):      # weight-normalised PathNet layers (oracle/modules.py): g != ||v||, so that the
                    p.mul_(torch.empty_like(p).uniform_(0.7, 1.3))      # normalisation is exercised, not the identity it starts as
    return models


def gen_interface_samples(ref_losses, ref_itf):
    """The reference's SBMCInterface / LBMCInterface (interfaces.py:336-523, 753-839) around stand-ins for the external
    base denoisers: one train step (loss sums, gradients after the norm clip, parameters after Adam) and one
    validation step.  The large initial gradient scale (x 4000 on the target) makes the LBMC clamp (250) bite."""
    from oracle.step import draw_perms
    for ci, case in enumerate(SAMPLE_CASES):
        kind, use_llpm, manif, option, pout, recon, nfeat = SAMPLE_CASES[case]
        models = build_sample_models(case, 900 + ci)
        init_state = {"%s/%s" % (mn, k): np_(v) for mn, m in models.items() for k, v in m.state_dict().items()}
        optims = {"optim_" + mn: torch.optim.Adam(m.parameters(), lr=1e-3 if mn == "dncnn" else 2e-3)
                  for mn, m in models.items()}
        loss_funcs = {"l_recon": torch.nn.L1Loss() if recon == "L1Loss" else getattr(ref_losses, recon)(),
                      "l_test": ref_losses.RelativeMSE()}
        if manif:
            loss_funcs["l_manif"] = ref_losses.FeatureMSE(non_local=True)
        args = types.SimpleNamespace(model_name="golden")
        if kind == "SBMCInterface":
            itf = ref_itf.SBMCInterface(models, optims, loss_funcs, args, use_llpm_buf=use_llpm, manif_learn=manif,
                                        w_manif=0.1, disentangle=option)
        else:
            itf = ref_itf.LBMCInterface(models, optims, loss_funcs, args, use_llpm_buf=use_llpm, manif_learn=manif,
                                        w_manif=0.1, disentangle=option)
        itf.iters = 1
        batch = sample_batch(950 + ci, use_llpm, nfeat)
        batch["features"] = batch["features"] * SAMPLE_FEATURE_SCALE.get(case, 1.0)
        out = {"batch/" + k: np_(v) for k, v in batch.items()}
        out.update({"init/" + k: v for k, v in init_state.items()})
        B, S, H = G7_GEOM["B"], G7_GEOM["S"], G7_GEOM["H"]
        hc = H - 2 * G7_GEOM["DEPTH"]                        # the stand-in's valid 3x3 convs crop the P-buffer too
        seed = 980 + ci
        torch.manual_seed(seed)
        ip, ib = draw_perms(B, S, hc, hc)
        out["perm/patch"], out["perm/batch"], out["seed"] = np_(ip), np_(ib), np.array(seed)
        itf.to_train_mode()
        torch.manual_seed(seed)
        itf.preprocess(batch)
        itf.train_batch(batch)
        for k, v in itf.m_losses.items():
            out["m_losses/" + k] = np_(v)
        for mn, m in models.items():
            for k, v in m.state_dict().items():
                out["after/%s/%s" % (mn, k)] = np_(v)
            for k, p in m.named_parameters():
                out["grad/%s/%s" % (mn, k)] = np_(p.grad) if p.grad is not None else np.zeros(0, np.float32)
            out["gradnorm/" + mn] = np.array(float(torch.sqrt(sum((p.grad ** 2).sum() for p in m.parameters()
                                                                   if p.grad is not None))))
        itf.to_eval_mode()
        with torch.no_grad():
            rad, pb = itf.validate_batch(batch)
        out["val/out"] = np_(rad)
        if pb is not None:
            out["val/p_buffer"] = np_(pb)
        out["val/summary"] = np.array(itf.get_epoch_summary(mode="eval", norm=1))
        np.savez_compressed(os.path.join(HERE, "interface_%s.npz" % case), **out)
        print("G7", case, {k: float(v) for k, v in itf.m_losses.items()}, {k: float(out[k]) for k in out if k.startswith("gradnorm/")})


# ---------------------------------------------------------------------------------- G6
def raw_samples(h, w, s, seed, zero_depth=False):
    """Random raw renderer output (h, w, s, 104) with the value ranges the preprocessors care about:
    signed radiance (exercises the max(., 0) clamps), throughputs / intensities spanning decades with exact
    zeros (path ended), roughness in [0, 1], bounce-type codes."""
    rng = np.random.RandomState(seed)
    x = rng.rand(h, w, s, 104).astype(np.float32)
    x[..., 2:8] = (rng.randn(h, w, s, 6) * 2.0).astype(np.float32)                  # radiance, diffuse: signed
    x[..., 66:69] = rng.rand(h, w, s, 3).astype(np.float32)                         # albedo at first diffuse
    x[..., 69:72] = (rng.randn(h, w, s, 3)).astype(np.float32)                      # normal
    x[..., 72:73] = 0.0 if zero_depth else (rng.rand(h, w, s, 1) * 40.0).astype(np.float32)
    x[..., 73:74] = np.exp(rng.randn(h, w, s, 1) * 3.0).astype(np.float32)          # path weight
    x[..., 74:77] = np.exp(rng.randn(h, w, s, 3) * 2.0).astype(np.float32)
    x[..., 77:80] = (rng.rand(h, w, s, 3) * 1e4).astype(np.float32)
    thr = np.exp(rng.randn(h, w, s, 18) * 2.0).astype(np.float32)
    thr[rng.rand(h, w, s, 18) < 0.3] = 0.0
    x[..., 80:98] = thr
    x[..., 60:66] = rng.randint(0, 20, size=(h, w, s, 6)).astype(np.float32)        # bounce types
    return x


def gen_preprocess(ref_datasets):
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "train", "gt"))
        ds = ref_datasets.DenoiseDataset(tmp, 4, base_model="kpcn", mode="train", use_llpm_buf=True)
        out = {}
        for name, (h, w, s, seed, zd) in {"a": (12, 10, 4, 901, False), "b": (9, 17, 8, 902, False),
                                          "zero_depth": (6, 5, 2, 903, True)}.items():
            x = raw_samples(h, w, s, seed, zd)
            out[name + "/raw"] = x
            out[name + "/llpm"] = ds._preprocess_llpm(x.copy())
            out[name + "/kpcn"] = ds._preprocess_kpcn(x.copy())
        buf = np.random.RandomState(904).randn(7, 9, 5).astype(np.float32)
        out["grad/buf"], out["grad/out"] = buf, ds._gradients(buf)
        np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)
        print("G6", {k: v.shape for k, v in out.items()})


def gen_patches(ref_datasets):
    """G8: the REAL DenoiseDataset.__getitem__ (datasets.py:1026-1146, with _sample_patches :795-840 and _transpose
    :760-791) over a one-image dataset written to a temporary directory in the layout it expects; PATCH_SIZE is
    lowered from 128 to 12 on the instance so that the fixture stays small (the code path is the same)."""
    H, W, S, P, NPI = 28, 30, 2, 12, 4
    rs = np.random.RandomState(1234)
    kpcn = rs.rand(H, W, 44).astype(np.float32) * 2 - 0.5
    llpm = rs.rand(H, W, S, 37).astype(np.float32) - 0.3
    gt = rs.rand(H, W, 9).astype(np.float32) * 2            # total, diffuse, albedo (total - diffuse may be negative but > -1)
    gt[..., 0:3] = gt[..., 3:6] + rs.rand(H, W, 3).astype(np.float32)
    prob = np.zeros((H, W), np.float64)
    prob[:H - P + 1, :W - P + 1] = rs.rand(H - P + 1, W - P + 1)
    prob /= prob.sum()
    out = {"kpcn": kpcn, "llpm": llpm, "gt": gt, "prob": prob, "patch": np.array(P), "seed": np.array(4321)}
    for use_llpm in (True, False):
        with tempfile.TemporaryDirectory() as tmp:
            for sub in ("KPCN/train/gt", "KPCN/train/input", "LLPM/train/input"):
                os.makedirs(os.path.join(tmp, sub))
            np.save(os.path.join(tmp, "KPCN/train/gt/img0.npy"), gt)
            np.save(os.path.join(tmp, "KPCN/train/input/img0_kpcn_%d.npy" % S), kpcn)
            np.save(os.path.join(tmp, "KPCN/train/input/img0_prob_imp.npy"), prob)
            np.save(os.path.join(tmp, "LLPM/train/input/img0_llpm.npy"), llpm)
            ds = ref_datasets.DenoiseDataset(os.path.join(tmp, "KPCN"), S, base_model="kpcn", mode="train", batch_size=8,
                                             sampling="random", use_llpm_buf=use_llpm)
            ds.PATCH_SIZE = P
            np.random.seed(4321)
            items = [ds[i] for i in range(NPI)]             # item 0 draws all patches_per_image origins
            tag = "llpm" if use_llpm else "vanilla"
            out[tag + "/patches_per_image"] = np.array(ds.patches_per_image)
            for i, it in enumerate(items):
                for k, v in it.items():
                    out["%s/%d/%s" % (tag, i, k)] = np.ascontiguousarray(v)
    np.savez_compressed(os.path.join(HERE, "patches.npz"), **out)
    print("G8", sorted(set(k.split("/")[-1] for k in out if "/0/" in k)), {k: out[k].shape for k in out if k.startswith("llpm/0/")})


def main():
    ref_losses, ref_utils, ref_itf = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "preprocess":          # only the data-step functions
        import support.datasets as ref_datasets
        gen_preprocess(ref_datasets)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "patches":           # only the loader item (rank 3)
        import support.datasets as ref_datasets
        gen_patches(ref_datasets)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "samples":           # only the rank-2 "next" interfaces (SBMC / LBMC glue)
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                gen_interface_samples(ref_losses, ref_itf)
            finally:
                os.chdir(cwd)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "variants":          # only the rank-1 "next" interfaces
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                gen_interface_variants(ref_losses, ref_itf)
            finally:
                os.chdir(cwd)
        return
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            gen_crop_like(ref_utils)
            gen_fmse(ref_losses)
            gen_grs(ref_losses)
            gen_image_losses(ref_losses)
            gen_interface(ref_losses, ref_itf)
            gen_interface_variants(ref_losses, ref_itf)
            gen_interface_samples(ref_losses, ref_itf)
            import support.datasets as ref_datasets
            gen_preprocess(ref_datasets)
            gen_patches(ref_datasets)
        finally:
            os.chdir(cwd)
    print("goldens written to", HERE)


if __name__ == "__main__":
    main()
