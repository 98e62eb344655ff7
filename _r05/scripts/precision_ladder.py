"""Precision ladder of the BACKWARD GEMMs.

The benchmarked KPCN-Manifold step (8 patches of 128x128, S=8; eager, forward untouched: split-bf16, three MFMAs per
product) is run once per rung with the operands of the weight-gradient and data-gradient GEMMs rounded to the rung's
storage format BEFORE the library's split-bf16 kernels multiply them, and every parameter gradient is compared with an
fp64 CPU run of the same step (the yardstick) and with the fp32 CPU oracle (what tests/test_gpu_bench_config.py holds
the product to: relative L2 <= 2e-3, 1 - cos <= 2e-6 per tensor).

Why the emulation is faithful: a value rounded to bf16 (8 bits) or fp16 (11 bits) is EXACTLY representable as a bf16
hi + lo pair (16 bits), and the product of two such values is exact in fp32, so `hi*hi + hi*lo + lo*hi` of the existing
kernels differs from the rung's own MFMA sequence by the dropped lo*lo term only (2^-16 relative, an order of magnitude
below fp16's rounding), with the same fp32 accumulation.  Rounding and re-splitting are torch ops -- this is a
diagnostic script, not product code.

Operand formats:  full  = bf16 hi + lo (today, 16 bits)      bf16 = hi plane only (8 bits)
                  fp16  = one fp16 plane of x * 2^k (11 bits; k per tensor so that max|x| lands in [2^top, 2^(top+1)),
                          gradual underflow as the hardware conversion does)
Roles: wgrad = (dy, x), dgrad = (dy, W).  MFMAs per product: full x full 3, full x single 2, single x single 1.

   python3 scripts/precision_ladder.py [B] [--all-chains] [--top 14]
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
from wcmc_amd.support.interfaces import KPCNInterface
from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
from wcmc_amd.support.networks import PathNet
from wcmc_amd.synthetic import make_batch

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(argv[0]) if argv else 8
ALL_CHAINS = "--all-chains" in sys.argv
TOP = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 14

# ------------------------------------------------------------------ rounding of split tensors / weights
FMT = dict(wgrad_dy="full", wgrad_x="full", dgrad_dy="full", dgrad_w="full")
IN_BWD = [False, 0]          # inside _chainx_backward, ksize of the chain


def q_values(v, fmt):
    if fmt == "full":
        return v
    if fmt == "bf16":
        return v.bfloat16().float()
    assert fmt == "fp16"
    m = float(v.abs().max())
    if m == 0.0 or not math.isfinite(m):
        return v
    k = TOP - math.floor(math.log2(m))
    s = 2.0 ** k
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


_wgrad, _igemm, _packx, _bwd, _pair = ops.conv2d_wgrad_x_raw, ops.conv2d_x_raw, ops._pack_x, ops._chainx_backward, ops.conv1x1_pair_x_raw


def active():
    return IN_BWD[0] and (ALL_CHAINS or IN_BWD[1] == 5)


def wgrad(xs, xdims, dys, cout, ks, pad, *a, **kw):
    if active():
        n, cin, h, w = xdims
        ho, wo = h + 2 * pad - ks + 1, w + 2 * pad - ks + 1
        xs = q_split(xs, xdims, FMT["wgrad_x"])
        dys = q_split(dys, (n, cout, ho, wo), FMT["wgrad_dy"])
    return _wgrad(xs, xdims, dys, cout, ks, pad, *a, **kw)


def igemm(xs, dims, *a, **kw):
    if active():
        xs = q_split(xs, dims, FMT["dgrad_dy"])
    return _igemm(xs, dims, *a, **kw)


def pair(xs, dims, *a, **kw):          # the fused data gradient of PathNet.final's two layers (only its first GEMM's dy is rounded)
    if active():
        xs = q_split(xs, dims, FMT["dgrad_dy"])
    return _pair(xs, dims, *a, **kw)


def packx(weight, mode):
    if active() and mode == 1:
        weight = q_values(weight.detach(), FMT["dgrad_w"])
    return _packx(weight, mode)


def bwd(ctx, *a, **kw):
    IN_BWD[0], IN_BWD[1] = True, ctx.spec[0]
    wp1, ctx.wp1 = ctx.wp1, None          # pack the data-gradient weights in the backward, through packx above
    try:
        return _bwd(ctx, *a, **kw)
    finally:
        IN_BWD[0] = False
        ctx.wp1 = wp1


ops.conv2d_wgrad_x_raw, ops.conv2d_x_raw, ops._pack_x, ops._chainx_backward, ops.conv1x1_pair_x_raw = wgrad, igemm, packx, bwd, pair

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
ostep.train_step(o32, mk(o32), batch, cfg, perms)
t1 = time.time()
ostep.train_step(o64, mk(o64), {k: v.double() for k, v in batch.items()}, cfg, perms)
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

names = [(mn, k) for mn in o32 for k, _ in o32[mn].named_parameters()]
g32 = {(mn, k): p.grad for mn in o32 for k, p in o32[mn].named_parameters()}
g64 = {(mn, k): p.grad for mn in o64 for k, p in o64[mn].named_parameters()}


def run(fmt):
    FMT.update(fmt)
    for m in hm.values():
        m.zero_grad()
    torch.manual_seed(1234)          # same pairings (CPU generator, reference order)
    itf.preprocess(db)
    itf.train_batch(db)
    torch.cuda.synchronize()
    out = {}
    for mn in hm:
        for k, p in hm[mn].named_parameters():
            gh = p.grad.detach().cpu()
            out[(mn, k)] = (rl2(gh, g64[(mn, k)]), rl2(gh, g32[(mn, k)]), cosd(gh, g32[(mn, k)]))
    return out


RUNGS = [
    # name, formats, MFMAs per product (wgrad, dgrad)
    ("A  bf16x3 (today)", dict(wgrad_dy="full", wgrad_x="full", dgrad_dy="full", dgrad_w="full"), (3, 3)),
    ("B  dy bf16 hi-only, x/W full", dict(wgrad_dy="bf16", wgrad_x="full", dgrad_dy="bf16", dgrad_w="full"), (2, 2)),
    ("C  dy full, x/W bf16 hi-only", dict(wgrad_dy="full", wgrad_x="bf16", dgrad_dy="full", dgrad_w="bf16"), (2, 2)),
    ("C' wgrad only: x bf16 hi-only", dict(wgrad_dy="full", wgrad_x="bf16", dgrad_dy="full", dgrad_w="full"), (2, 3)),
    ("D  all single bf16", dict(wgrad_dy="bf16", wgrad_x="bf16", dgrad_dy="bf16", dgrad_w="bf16"), (1, 1)),
    ("H  wgrad single bf16 (dy hi x x hi), dgrad bf16x3", dict(wgrad_dy="bf16", wgrad_x="bf16", dgrad_dy="full", dgrad_w="full"), (1, 3)),
    ("I  wgrad single bf16, dgrad dy bf16 hi x W full", dict(wgrad_dy="bf16", wgrad_x="bf16", dgrad_dy="bf16", dgrad_w="full"), (1, 2)),
    ("F  dy fp16 (scaled), x/W full [=fp16 two-term]", dict(wgrad_dy="fp16", wgrad_x="full", dgrad_dy="fp16", dgrad_w="full"), (2, 2)),
    ("G  dy full, x/W fp16 [=fp16 two-term]", dict(wgrad_dy="full", wgrad_x="fp16", dgrad_dy="full", dgrad_w="fp16"), (2, 2)),
    ("E  all single fp16 (dy scaled per tensor)", dict(wgrad_dy="fp16", wgrad_x="fp16", dgrad_dy="fp16", dgrad_w="fp16"), (1, 1)),
    ("E1 wgrad single fp16, dgrad bf16x3", dict(wgrad_dy="fp16", wgrad_x="fp16", dgrad_dy="full", dgrad_w="full"), (1, 3)),
    ("E2 dgrad single fp16, wgrad bf16x3", dict(wgrad_dy="full", wgrad_x="full", dgrad_dy="fp16", dgrad_w="fp16"), (3, 1)),
    ("E3 wgrad single fp16, dgrad dy fp16 x W full", dict(wgrad_dy="fp16", wgrad_x="fp16", dgrad_dy="fp16", dgrad_w="full"), (1, 2)),
]
print("# backward-GEMM precision ladder: %s, B=%d, fp16 scale: max|x| -> [2^%d, 2^%d)" %
      ("ALL conv chains (KPCN 5x5, U-Net 3x3, PathNet 1x1)" if ALL_CHAINS else "KPCN 5x5 chains only", B, TOP, TOP + 1))
print("# per-tensor relative L2 of the parameter gradients over the 116 tensors; bar of test_gpu_bench_config: "
      "vs fp32 oracle <= 2e-3, 1-cos <= 2e-6")
print("# fp32 CPU oracle vs fp64: max %.2e median %.2e" %
      (max(rl2(g32[k], g64[k]) for k in names), st.median(rl2(g32[k], g64[k]) for k in names)))
print("%-50s %5s | %9s %9s | %9s %9s %9s | worst tensor (vs fp32 oracle)" %
      ("rung", "MFMA", "max v64", "med v64", "max v32", "med v32", "max 1-cos"))
for name, fmt, cost in RUNGS:
    r = run(fmt)
    worst = max(names, key=lambda k: r[k][1])
    print("%-50s %d / %d | %9.2e %9.2e | %9.2e %9.2e %9.2e | %s %s" %
          (name, cost[0], cost[1], max(v[0] for v in r.values()), st.median(v[0] for v in r.values()),
           max(v[1] for v in r.values()), st.median(v[1] for v in r.values()), max(v[2] for v in r.values()),
           worst[0], worst[1]), flush=True)
    if name.startswith(("A", "E ", "I ")):
        kp = sorted((k for k in names if k[0] == "dncnn" and k[1].endswith("weight")), key=lambda k: k[1])
        print("     KPCN weight gradients vs fp64, layer 0..8: diffuse " +
              " ".join("%.1e" % r[k][0] for k in kp if "diffuse" in k[1]) + " | specular " +
              " ".join("%.1e" % r[k][0] for k in kp if "specular" in k[1]))
