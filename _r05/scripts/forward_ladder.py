"""Precision ladder of the FORWARD GEMMs.

The benchmarked KPCN-Manifold step (8 patches of 128x128, S=8; eager; backward as the default mode runs it: data gradient
dy_hi x (W_hi + W_lo), weight gradient dy_hi x x_hi) is run once per rung with the operands of the forward GEMMs rounded to
the rung's storage format BEFORE the library's three-term split-bf16 kernels multiply them:

  * activations -- a chain's input and every hidden activation a layer writes -- are rounded where they are STORED, so the
    next layer's forward, the layer's weight gradient and the ReLU gates all see the rounded tensor, as they would with
    narrower storage;
  * the forward-orientation weight pack is made from rounded weights; the data gradient keeps W_hi + W_lo.

Why the emulation is faithful: a bf16 (8 bits) or fp16 (11 bits) number is EXACTLY a bf16 hi + lo pair, and products of such
numbers are exact in fp32, so `hi*hi + hi*lo + lo*hi` differs from the rung's own MFMA sequence by the dropped lo*lo term
only (2^-16 relative), with the same fp32 accumulation.  Rounding and re-splitting are torch ops: a diagnostic script, not
product code.  The fused PathNet chains (embed3 / final2 / layer pairs) are switched off so that every layer passes through
the hooks; their forward is bit-identical to the layer-by-layer path (tests/test_gpu_models.py).

Formats:  full = bf16 hi + lo (today)      bf16 = hi plane only      fp16 = one fp16 plane of x * 2^k (k per tensor, max|x|
          in [2^top, 2^(top+1)))           fp16u = plain fp16, no scale
MFMAs per product: full x full 3, full x single 2, single x single 1.

Measured per rung, against an fp64 CPU run and the fp32 CPU oracle of the same step (same weights, inputs, pairings):
max|a-b|/max|b| of radiance / diffuse / specular, the relative error of every loss scalar, per-tensor relative L2 and 1 - cos
of the parameter gradients, and the share of hidden ReLU units whose sign differs from rung A's.

   python3 scripts/forward_ladder.py [B] [--chains kpcn,unet,pw,all,last  (comma list: one table per entry; last = the KPCN output layers only)] [--top 14]
"""
import copy
import math
import os
import statistics as st
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.set_num_threads(min(32, os.cpu_count() or 1))
from oracle import step as ostep
from oracle.models import KPCN as OKPCN
from oracle.networks import PathNet as OPathNet
from wcmc_amd import KPCN, ops
from wcmc_amd._lib import lib
from wcmc_amd.support.interfaces import KPCNInterface
from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
from wcmc_amd.support.networks import PathNet
from wcmc_amd.synthetic import make_batch

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(argv[0]) if argv and argv[0].isdigit() else 8
CHAIN_SETS = (sys.argv[sys.argv.index("--chains") + 1] if "--chains" in sys.argv else "all").split(",")
TOP = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 14
KS_SETS = {"kpcn": (5,), "unet": (3,), "pw": (1,), "all": (5, 3, 1), "conv": (5, 3), "last": ("last",)}
KS_OF = [KS_SETS[CHAIN_SETS[0]]]
assert ops.PRECISION == "bf16x321"

FMT = dict(x="full", w="full")
IN_FWD = [False, 0]


def q_values(v, fmt):
    if fmt == "full":
        return v
    if fmt == "bf16":
        return v.bfloat16().float()
    if fmt == "fp16u":
        return v.half().float()
    assert fmt == "fp16"
    m = float(v.abs().max())
    if m == 0.0 or not math.isfinite(m):
        return v
    s = 2.0 ** (TOP - math.floor(math.log2(m)))
    return (v * s).half().float() / s


def q_split(t, dims, fmt):
    """Round the values of a split tensor (int16 storage of [N][H][W][2][round_up(C,8)] bf16) and split them again."""
    if fmt == "full":
        return t
    n, c, h, w = dims
    cp = (c + 7) // 8 * 8
    v = t.view(torch.bfloat16).view(n, h, w, 2, cp).float()
    val = q_values(v[:, :, :, 0] + v[:, :, :, 1], fmt)
    hi = val.bfloat16()
    lo = (val - hi.float()).bfloat16()
    assert torch.equal(hi.float() + lo.float(), val), "a rounded value must be exact as a hi + lo pair"
    return torch.stack([hi, lo], 3).contiguous().view(torch.int16).view(-1)


_igemm, _packx, _fwd = ops.conv2d_x_raw, ops._pack_x, ops._chainx_forward


def active():
    return IN_FWD[0] and IN_FWD[1] in KS_OF[0]


def igemm(xs, dims, wp, bias, cout, ks, pad, act, out_split, *a, **kw):
    if IN_FWD[0] and KS_OF[0] == ("last",) and ks == 5 and not out_split:
        # "last": only the GEMM of the KPCN chains' OUTPUT layer (100 -> 441 logits, no ReLU behind it: nothing to flip) sees the
        # rounded activation; what is stored (the weight gradient's x, the gates) stays as it is
        xs = q_split(xs, dims, FMT["x"])
    out = _igemm(xs, dims, wp, bias, cout, ks, pad, act, out_split, *a, **kw)
    if active() and out_split:          # a hidden activation: stored in the rung's format
        n, _, h, w = dims
        od = (n, cout, h + 2 * pad - ks + 1, w + 2 * pad - ks + 1)
        if isinstance(out, tuple):
            out = (q_split(out[0], od, FMT["x"]),) + tuple(out[1:])
        else:
            out = q_split(out, od, FMT["x"])
    return out


def packx(weight, mode):
    if mode == 0 and IN_FWD[0] and KS_OF[0] == ("last",) and weight.shape[0] == 441:
        weight = q_values(weight.detach(), FMT["w"])
    if active() and mode == 0:
        weight = q_values(weight.detach(), FMT["w"])
    return _packx(weight, mode)


def fwd(ctx, xs0, dims0, spec, params, extra_saved=None):
    IN_FWD[0], IN_FWD[1] = True, spec[0]
    try:
        if active():
            xs0 = q_split(xs0, dims0, FMT["x"])
        return _fwd(ctx, xs0, dims0, spec, params, extra_saved)
    finally:
        IN_FWD[0] = False


ops.conv2d_x_raw, ops._pack_x, ops._chainx_forward = igemm, packx, fwd
ops.PACK_CHAIN = False            # per-layer packing through packx (the data-gradient pack keeps the full weights)
ops.FUSE_EMBED = ops.FUSE_FINAL = False
lib().wcmc_conv1x1_pair_supported = lambda *a: 0

# ------------------------------------------------------------------ the step, three ways
torch.manual_seed(0)
o32 = {"dncnn": OKPCN(39), "backbone_diffuse": OPathNet(36), "backbone_specular": OPathNet(36)}
g = torch.Generator().manual_seed(77)
for m in o32.values():
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.rand(p.shape, generator=g) * 0.2 - 0.1)
o64 = {k: copy.deepcopy(m).double() for k, m in o32.items()}
hm = {"dncnn": KPCN(39), "backbone_diffuse": PathNet(36), "backbone_specular": PathNet(36)}
for k in hm:
    hm[k].load_state_dict(o32[k].state_dict())
    hm[k].to("cuda")
batch = make_batch(B, 8, 128, seed=40, device="cpu")
cfg = dict(use_llpm_buf=True, manif_learn=True, train_branches=True, disentanglement_option="m11r11", w_manif=0.1)
torch.manual_seed(1234)
perms = [ostep.draw_perms(B, 8, 92, 92), ostep.draw_perms(B, 8, 92, 92)]
mk = lambda ms: {"optim_" + k: torch.optim.SGD(m.parameters(), lr=0.0) for k, m in ms.items()}
t0 = time.time()
l32, out32 = ostep.train_step(o32, mk(o32), batch, cfg, perms)
t1 = time.time()
l64, out64 = ostep.train_step(o64, mk(o64), {k: v.double() for k, v in batch.items()}, cfg, perms)
print("# CPU oracle: fp32 step %.1f s, fp64 step %.1f s (B=%d)" % (t1 - t0, time.time() - t1, B), flush=True)
lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
      "l_test": RelativeMSE(), "l_manif": FeatureMSE(non_local=True)}
itf = KPCNInterface(hm, mk(hm), lf, types.SimpleNamespace(model_name="d"), use_llpm_buf=True, manif_learn=True,
                    w_manif=0.1, train_branches=True)
itf.iters = 1
itf.to_train_mode()
db = {k: v.to("cuda") for k, v in batch.items()}
rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
cosd = lambda a, b: 1.0 - float((a.double().flatten() @ b.double().flatten()) / (a.double().norm() * b.double().norm()))
mrel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())

names = [(mn, k) for mn in o32 for k, _ in o32[mn].named_parameters()]
g32 = {(mn, k): p.grad for mn in o32 for k, p in o32[mn].named_parameters()}
g64 = {(mn, k): p.grad for mn in o64 for k, p in o64[mn].named_parameters()}
OUTS = ("radiance", "diffuse", "specular")
LOSSES = [k for k in l32]
SIGNS_A = [None]


def run(fmt, signs=True):
    FMT.update(fmt)
    for m in hm.values():
        m.zero_grad()
    torch.manual_seed(1234)          # same pairings (CPU generator, reference order)
    ops.DEBUG_ACTS = [] if signs else None
    itf.preprocess(db)
    itf.train_batch(db)
    torch.cuda.synchronize()
    flips = None
    if signs:
        sg = [(t > 0) for t in ops.DEBUG_ACTS]
        ops.DEBUG_ACTS = None
        if SIGNS_A[0] is None:
            SIGNS_A[0] = sg
        flips = sum(int((a != b).sum()) for a, b in zip(sg, SIGNS_A[0])) / float(sum(a.numel() for a in sg))
    r = dict(flips=flips)
    r["out64"] = {k: mrel(itf.last_out[k].cpu(), out64[k]) for k in OUTS}
    r["out32"] = {k: mrel(itf.last_out[k].cpu(), out32[k]) for k in OUTS}
    r["loss64"] = {k: abs(float(itf.last_loss_dict[k]) - float(l64[k])) / abs(float(l64[k])) for k in LOSSES}
    r["loss32"] = {k: abs(float(itf.last_loss_dict[k]) - float(l32[k])) / abs(float(l32[k])) for k in LOSSES}
    gr = {}
    for mn in hm:
        for k, p in hm[mn].named_parameters():
            gh = p.grad.detach().cpu()
            gr[(mn, k)] = (rl2(gh, g64[(mn, k)]), rl2(gh, g32[(mn, k)]), cosd(gh, g32[(mn, k)]))
    r["grad"] = gr
    return r


RUNGS = [
    # name, formats, MFMAs per product
    ("A  x full, W full (today)", dict(x="full", w="full"), 3),
    ("B  x bf16 hi-only, W full", dict(x="bf16", w="full"), 2),
    ("C  x full, W bf16 hi-only", dict(x="full", w="bf16"), 2),
    ("D  x bf16, W bf16", dict(x="bf16", w="bf16"), 1),
    ("F  x fp16 (scaled), W full [fp16 two-term]", dict(x="fp16", w="full"), 2),
    ("G  x full, W fp16 (scaled) [fp16 two-term]", dict(x="full", w="fp16"), 2),
    ("E  x fp16, W fp16 (both scaled per tensor)", dict(x="fp16", w="fp16"), 1),
    ("Eu x fp16, W fp16 (no scales)", dict(x="fp16u", w="fp16u"), 1),
    ("Fu x fp16 (no scale), W full", dict(x="fp16u", w="full"), 2),
]
for CHAINS in CHAIN_SETS:
    KS_OF[0] = KS_SETS[CHAINS]
    print("# forward-GEMM precision ladder: chains %s (ksize %s), B=%d, fp16 scale: max|x| -> [2^%d, 2^%d)" % (CHAINS, KS_OF[0], B, TOP, TOP + 1))
    print("# north_star: outputs and loss scalars within 1e-3 relative; adoption bar (VERDICT r3): <= 3e-4 and the gradient bars of"
          " tests/test_gpu_bench_config.py unchanged (vs fp32 oracle: relative L2 <= 2e-3, 1 - cos <= 2e-6 per tensor)")
    print("# fp32 CPU oracle vs fp64: outputs %s | losses max %.2e | gradients max %.2e median %.2e" %
          (" ".join("%.1e" % mrel(out32[k], out64[k]) for k in OUTS),
           max(abs(float(l32[k]) - float(l64[k])) / abs(float(l64[k])) for k in LOSSES),
           max(rl2(g32[k], g64[k]) for k in names), st.median(rl2(g32[k], g64[k]) for k in names)))
    print("%-46s %4s | %-26s | %-26s | %9s %9s | %9s %9s %9s | %9s | worst" %
          ("rung", "MFMA", "out max-rel v64 (rad dif spe)", "out max-rel v32", "loss v64", "loss v32", "grad v64", "grad v32", "1-cos v32", "flips"))
    for name, fmt, cost in RUNGS:
        r = run(fmt)
        gr = r["grad"]
        worst = max(names, key=lambda k: gr[k][1])
        wl = max(LOSSES, key=lambda k: r["loss32"][k])
        print("%-46s %4d | %-26s | %-26s | %9.2e %9.2e | %9.2e %9.2e %9.2e | %9.2e | %s %s; loss %s" %
              (name, cost, " ".join("%.2e" % r["out64"][k] for k in OUTS), " ".join("%.2e" % r["out32"][k] for k in OUTS),
               max(r["loss64"].values()), max(r["loss32"].values()),
               max(v[0] for v in gr.values()), max(v[1] for v in gr.values()), max(v[2] for v in gr.values()),
               r["flips"], worst[0], worst[1], wl), flush=True)
        print("     losses vs fp64: " + " ".join("%s %.1e" % (k, r["loss64"][k]) for k in LOSSES))
        kp = sorted((k for k in names if k[0] == "dncnn" and k[1].endswith("weight")), key=lambda k: k[1])
        print("     KPCN weight gradients vs fp32 oracle, layer 0..8: diffuse " +
              " ".join("%.1e" % gr[k][1] for k in kp if "diffuse" in k[1]) + " | specular " +
              " ".join("%.1e" % gr[k][1] for k in kp if "specular" in k[1]), flush=True)
