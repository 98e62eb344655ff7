#!/usr/bin/env python3
"""Headline benchmark: 128x128 MC patches/sec of one full KPCN-Manifold train step on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = ``KPCNInterface.preprocess`` + ``train_batch`` (support/interfaces.py:108-192) on config C3 of
BASELINE.json: KPCN(n_in=39) + 2 x PathNet(36->3) + FeatureMSE (w=0.1, m11r11, train_branches), 8 patches
of 128x128 with S=8 spp per GPU, inputs resident in HBM, fused clip+Adam, RCCL gradient all-reduce when
N > 1 (weak scaling: 8 patches per GPU).  Rank 0 prints ONE JSON line.

Extra objects on that line
  roofline           dominant kernel class (conv implicit-GEMM, MFMA-bound): algorithmic FLOP / launch time
                     measured with HIP events on the launch stream inside the timed region
  roofline_kernel_apply   the HBM-bound op the north star puts a >= 40 % target on
  cpu_baseline       the CPU oracle's train step on this box's host cores (rank 0, N == 1 only)
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA, dense (not the 2:1-sparse figure)
PEAK_HBM_GBS = 8000.0              # HBM3E spec (6.3 TB/s achievable)
_ROWS8_ENV = os.environ.get("WCMC_WGRAD_ROWS8")                 # which filter-row weight-gradient kernel the library launches:
_rows8 = lambda terms: (_ROWS8_ENV[:1] != "0") if _ROWS8_ENV else terms == 3      # eight waves for three-term launches, seven for one-term ones
_ROWS8_XE = 0 if os.environ.get("WCMC_WGRAD_ROWS8_XE", "1")[:1] == "0" else 1
PROFILE_ROUND = "r06"              # profiles/<round>_pmc_summary.json, <round>_bench_kernel_stats.csv: the evidence of THIS binary


def rocprof_names(wgrad_terms):
    """profiler class -> the rocprofv3 name of the ONE kernel its launches run (tests/test_cpu_host.py checks every name
    against the committed kernel stats).  Template arguments of conv_halo64: <cout tiles, weight stages, pixel tiles per wave,
    debug, halo stride, planes of x multiplied, planes of W multiplied, fp16 operands>; of conv_wgrad_rows8: <debug, dealing of the left-over tiles, planes multiplied>."""
    return {"conv_halo7": "wcmc::conv_halo_bf16x3_kernel<7, 8, 16, 0, 2, 2>",
            "conv_halo64_pt4": "wcmc::conv_halo64_bf16x3_kernel<7, 3, 4, 0, 80, 2, 2, 0>",
            "conv_halo64_pt3": "wcmc::conv_halo64_bf16x3_kernel<7, 3, 3, 0, 80, 2, 2, 0>",
            "conv_halo64_cs32": "wcmc::conv_halo64_bf16x3_kernel<7, 2, 3, 0, 160, 2, 2, 0>",
            "conv_halo64_pt4_x2": "wcmc::conv_halo64_bf16x3_kernel<7, 3, 4, 0, 80, 1, 2, 0>",
            "conv_halo64_pt3_x2": "wcmc::conv_halo64_bf16x3_kernel<7, 3, 3, 0, 80, 1, 2, 0>",
            "conv_halo64_pt4_x1": "wcmc::conv_halo64_bf16x3_kernel<7, 3, 4, 0, 80, 1, 1, 0>",
            "conv_halo64_pt3_x1": "wcmc::conv_halo64_bf16x3_kernel<7, 3, 3, 0, 80, 1, 1, 0>",
            "conv_halo64_pt4_h1": "wcmc::conv_halo64_bf16x3_kernel<7, 3, 4, 0, 80, 1, 1, 1>",
            "conv_halo64_pt3_h1": "wcmc::conv_halo64_bf16x3_kernel<7, 3, 3, 0, 80, 1, 1, 1>",
            "conv_wgrad_rows": ("wcmc::conv_wgrad_rows8_bf16x3_kernel<0, %d, %d>" % (_ROWS8_XE if wgrad_terms == 3 else 1, 1 if wgrad_terms == 1 else 2)) if _rows8(wgrad_terms)
                               else "wcmc::conv_wgrad_rows_bf16x3_kernel<5, 7, 7, 0, %d>" % (1 if wgrad_terms == 1 else 2),
            # the U-Net's 3x3 layers: <planes of x multiplied, stages per tap, K groups, debug>; the two-term class runs two instances
            # (64- and 128-channel slabs of the hi plane)
            "conv_halo3": "wcmc::conv_halo3_bf16x3_kernel<2, 2, 2, 0>",
            "conv_halo3_x2": "wcmc::conv_halo3_bf16x3_kernel<1, 4, 2, 0> and <1, 2, 2, 0>",
            "conv_pw": "wcmc::conv_pw_bf16x3_kernel<4|8, U, split> (the 1x1 PathNet layers)",
            # the fused PathNet chains (csrc/pathnet_fused.hip); the backward brackets include their small finish kernels
            "embed3_fwd": "wcmc::embed3_fwd_kernel", "embed3_bwd": "wcmc::embed3_bwd_kernel",
            "final2_fwd": "wcmc::final2_kernel<false>", "final2_bwd": "wcmc::final2_kernel<true>"}


# bf16 MFMAs issued per algorithmic multiply-add, by profiler class (forward 3; "_x2" data gradients 2; weight gradient: the mode's)
def mfma_terms(cls, wgrad_terms):
    return 1.0 if cls.endswith(("_x1", "_h1")) else 2.0 if cls.endswith("_x2") else float(wgrad_terms) if cls.startswith("conv_wgrad") else 3.0
B_PER_GPU, SPP, PATCH = 8, 8, 128
TWO_STREAM = True                 # the default form of the captured step (GraphedTrainStep(two_stream=...)); --one-graph: the other


class EventProfiler:
    """HIP-event pairs around op launches (wcmc_amd.ops._Timed).  Per class the summary takes the MEDIAN
    profiled step times the number of steps: one bracket in a few thousand straddles a host stall (the queue
    of an eagerly enqueued step running dry) and would otherwise add ~100 ms to its class."""

    def __init__(self):
        self.rows = []
        self.step = 0

    def next_step(self):
        self.step += 1

    def add(self, name, work, unit, e0, e1):
        self.rows.append((name, work, unit, e0, e1, self.step))

    def summary(self):
        per = {}
        for name, work, unit, e0, e1, st in self.rows:
            d = per.setdefault(name, {"unit": unit, "steps": {}})
            a = d["steps"].setdefault(st, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += work
        out = {}
        for name, d in per.items():
            steps = sorted(d["steps"].values(), key=lambda a: a[1])
            med = steps[len(steps) // 2]
            n = len(steps)
            out[name] = {"launches": med[0] * n, "ms": med[1] * n, "work": med[2] * n, "unit": d["unit"]}
        return out


# Seed of the initial weights.  PathNet.final ends in a ReLU on top of ReLU activations (support/networks.py:23-24): at
# initialisation every output channel of a P-buffer is positive almost everywhere or almost nowhere, a coin per channel, and a
# channel that starts dead gets no gradient.  Seed 0 draws a specular PathNet with NO live channel (3 % of its entries positive)
# and a diffuse one with one of three; seed 1 draws live channels in both, and on the round-6 synthetic scenes
# (wcmc_amd/synthetic.py: ``scene=True``) both manifold terms then keep moving for 200 steps (profiles/r06_pbuffer_death.txt).
WEIGHT_SEED = 1


def build_interface(device, group, rng="device", weight_norm=True, seed=None):
    """weight_norm: the PathNets' parametrisation -- True = upstream sbmc's ConvChain default, which
    ``support/networks.py:18-24`` does not switch off (w = g * v / ||v|| per layer); False = plain weights (rounds 1-4)."""
    seed = WEIGHT_SEED if seed is None else seed
    from wcmc_amd import KPCN
    from wcmc_amd.optim import FusedClipAdam
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    torch.manual_seed(seed)                                         # train_kpcn.py:346-348 seeds the generator the same way (with 0)
    models = {"dncnn": KPCN(39), "backbone_diffuse": PathNet(ic=36, outc=3, weight_norm=weight_norm),
              "backbone_specular": PathNet(ic=36, outc=3, weight_norm=weight_norm)}
    for k in models:
        models[k] = models[k].to(device)
    optims = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-4) for k, m in models.items()}
    loss_funcs = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(),
                  "l_recon": torch.nn.L1Loss(), "l_test": RelativeMSE(),
                  "l_manif": FeatureMSE(non_local=True, rng=rng)}
    itf = KPCNInterface(models, optims, loss_funcs, types.SimpleNamespace(model_name="bench"),
                        use_llpm_buf=True, manif_learn=True, w_manif=0.1, train_branches=True,
                        disentanglement_option="m11r11")
    itf.fused_optim = FusedClipAdam(models, optims, process_group=group)
    itf.iters = 1          # not iteration 1: skip the debug PNG dump (interfaces.py:130-137)
    itf.to_train_mode()
    return itf


def kernel_apply_probe(device, iters=24, nsets=4):
    """Back-to-back launches of the kernel-apply op through the C ABI at the step's shapes ((8,441,92,92) logits per
    branch) into preallocated buffers, HIP events around the whole train so that neither host launch gaps nor the
    allocator are inside the measurement.  COLD-cache: the launches rotate over `nsets` buffer sets (4 x (121 MB logits
    + 121 MB d_logits) = 0.97 GB, well beyond the 256 MiB Infinity Cache whose hits FETCH_SIZE would count), so no
    launch finds its logits on the die."""
    from wcmc_amd import ops
    from wcmc_amd._lib import check, lib
    n, k2, h = B_PER_GPU, 441, PATCH - 36
    sets = []
    for _ in range(nsets):
        sets.append(dict(logits=ops.nhwc_empty(n, k2, h, h, device).normal_(), dlog=ops.nhwc_empty(n, k2, h, h, device),
                         data=torch.rand(n, 3, h, h, device=device), g=torch.randn(n, 3, h, h, device=device),
                         res=torch.empty(n, 3, h, h, device=device), lse=torch.empty(n * h * h, device=device)))
    P, V, S = ops._ptr, ops._v, ops._stream

    def fwd(b):
        check(lib().wcmc_kernel_apply_fwd(*V(b["logits"]), P(b["data"]), *b["data"].stride(), P(b["res"]), *b["res"].stride(),
                                          P(b["lse"]), n, 3, h, h, 21, S()), "kernel_apply_fwd")

    def bwd(b):
        check(lib().wcmc_kernel_apply_bwd(*V(b["logits"]), P(b["data"]), *b["data"].stride(), P(b["res"]), *b["res"].stride(),
                                          P(b["g"]), *b["g"].stride(), P(b["lse"]), *V(b["dlog"]), P(None), n, 3, h, h, 21, S()),
              "kernel_apply_bwd")

    out = {}
    px = n * h * h
    for name, fn, nbytes in (("fwd", fwd, 4.0 * px * (k2 + 6)), ("bwd", bwd, 4.0 * px * (2 * k2 + 9))):
        for b in sets:
            fn(b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            fn(sets[i % nsets])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[name] = {"kernel": "kernel_apply_" + name, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                     "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
                     "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": nbytes,
                     "sample": "%d back-to-back launches rotating over %d buffer sets (%.2f GB: cold Infinity Cache), "
                               "logits (%d,441,%d,%d)" % (iters, nsets, nsets * 2 * 4.0 * px * 444 / 1e9, n, h, h)}
    return out


def measured_peaks(device):
    """What THIS box sustains on library kernels, printed beside the spec constants the roofline fractions use ("take
    gfx950 numbers measured on the box ... state which was used"): a 1 GiB device copy and read-only sum (HBM), and an
    8192^3 bf16 GEMM through torch (hipBLASLt) -- the matrix pipe under a dense load at the clock it settles to."""
    def timeit(fn, n):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    n = 256 * 1024 * 1024
    x = torch.empty(n, device=device).normal_()
    y = torch.empty_like(x)
    t_copy, t_read = timeit(lambda: y.copy_(x), 10), timeit(lambda: x.sum(), 10)
    del y
    a = torch.randn(8192, 8192, device=device, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=device, dtype=torch.bfloat16)
    t_mm = timeit(lambda: torch.matmul(a, b), 10)
    return {"hbm_copy_GBs": round(2 * n * 4 / t_copy / 1e9, 0), "hbm_read_GBs": round(n * 4 / t_read / 1e9, 0),
            "bf16_gemm_8192_TFLOPs": round(2 * 8192.0 ** 3 / t_mm / 1e12, 0),
            "used_for_frac": {"hbm_GBs": PEAK_HBM_GBS, "bf16_mfma_TFLOPs": PEAK_BF16_MFMA_TFLOPS, "fp32_mfma_TFLOPs": PEAK_FP32_MFMA_TFLOPS,
                              "source": "MI355X_MICROARCH.md spec peaks (the fractions are against these, not against the measured values)"}}


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/<round>_pmc_summary.json):
    (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950 correction of MI355X_MICROARCH.md section HBM."""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_pmc_summary.json")
    if not os.path.isfile(path):
        return {}
    with open(path) as f:
        d = json.load(f)
    pick = {}
    for k, v in d.items():
        for tag, key in (("conv_halo_bf16x3_kernel<7, 8, 16", "conv_halo7"), ("conv_halo64_bf16x3_kernel<7, 3, 4, 0, 80, 2, 2, 0>", "conv_halo64_pt4"),
                         ("conv_halo64_bf16x3_kernel<7, 3, 3, 0, 80, 2, 2, 0>", "conv_halo64_pt3"), ("conv_halo64_bf16x3_kernel<7, 2, 3", "conv_halo64_cs32"),
                         ("conv_halo64_bf16x3_kernel<7, 3, 4, 0, 80, 1, 2, 0>", "conv_halo64_pt4_x2"), ("conv_halo64_bf16x3_kernel<7, 3, 3, 0, 80, 1, 2, 0>", "conv_halo64_pt3_x2"),
                         ("conv_halo64_bf16x3_kernel<7, 3, 3, 0, 80, 1, 1, 0>", "conv_halo64_pt3_x1"), ("conv_halo64_bf16x3_kernel<7, 3, 3, 0, 80, 1, 1, 1>", "conv_halo64_pt3_h1"),
                         ("conv_halo64_bf16x3_kernel<7, 3, 4, 0, 80, 1, 1, 0>", "conv_halo64_pt4_x1"), ("conv_halo64_bf16x3_kernel<7, 3, 4, 0, 80, 1, 1, 1>", "conv_halo64_pt4_h1"),
                         ("conv_wgrad_rows8_bf16x3_kernel<0, 1, 1>" if _rows8(1) else "conv_wgrad_rows_bf16x3_kernel<5, 7, 7, 0, 1>", "conv_wgrad_rows"),
                         ("conv_pw_bf16x3_kernel<4, 16, true, 0>", "conv_pw"),
                         ("embed3_fwd_kernel", "embed3_fwd"), ("embed3_bwd_kernel", "embed3_bwd"),
                         ("final2_kernel<false>", "final2_fwd"), ("final2_kernel<true>", "final2_bwd"),
                         ("conv_halo3_bf16x3_kernel<2, 2, 2, 0>", "conv_halo3"), ("conv_halo3_bf16x3_kernel<1, 2, 2, 0>", "conv_halo3_x2"),
                         ("kernel_apply_strip_kernel<false", "kernel_apply_fwd"), ("kernel_apply_strip_kernel<true", "kernel_apply_bwd")):
            if tag in k and v.get("hbm_bytes_per_launch_corrected"):
                shape = ("64x128x128 64->64 1x1 hidden layer (PathNet embedding): 536.9 MB algorithmic" if key == "conv_pw" else
                         "64x128x128: PathNet.embedding 36->64->64->64 (+ spp mean) / PathNet.final 64+64->128->3, the benchmark's shape"
                         if key.startswith(("embed3", "final2")) else
                         "8x96x96 100->441 5x5 (the KPCN output layer, 92x92 outputs)" if key.endswith(("_x1", "_h1")) else
                         "8x100x100 100->100 5x5 (KPCN layer, 96x96 outputs: 12x16 tiles)" if key.startswith("conv_halo64_pt3") else
                         "8x128x128 64->64 3x3, pad 1 (a U-Net layer of the 128^2 level)" if key.startswith("conv_halo3") else
                         "8x116x116 100->100 5x5 (KPCN mid layer; its data gradient: 116x116 outputs = five tile rows of 16 + three of 12)" if key.startswith("conv") else "logits (8,441,92,92)")
                pick[key] = {"hbm_bytes_per_launch": v["hbm_bytes_per_launch_corrected"], "shape": shape,
                             "source": "profiles/%s_pmc_summary.json" % PROFILE_ROUND}
    return pick


def c2_leg(device, steps, warmup):
    """BASELINE configs[1] (BASELINE.md section 3 promises its number beside C3's): KPCN-Vanilla, diffuse + specular
    (n_in = 34, no PathNet, no manifold loss), 128x128, batch 8 on one MI355X -- the same graphed step machinery, the
    library's default arithmetic.  Parity of exactly this step: tests/test_gpu_bench_config.py::test_c2_vanilla_..."""
    from wcmc_amd import KPCN, ops
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.optim import FusedClipAdam
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import RelativeMSE
    from wcmc_amd.synthetic import make_batch
    torch.manual_seed(0)
    models = {"dncnn": KPCN(34).to(device)}
    optims = {"optim_dncnn": torch.optim.Adam(models["dncnn"].parameters(), lr=1e-4)}
    lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(), "l_test": RelativeMSE()}
    itf = KPCNInterface(models, optims, lf, types.SimpleNamespace(model_name="bench_c2"), train_branches=True)
    itf.fused_optim = FusedClipAdam(models, optims)
    itf.iters = 1
    itf.to_train_mode()
    batch = make_batch(B_PER_GPU, SPP, PATCH, seed=0, device=device, use_llpm=False)
    graphed = GraphedTrainStep(itf, batch, two_stream=TWO_STREAM, defer_check=True)
    batch = graphed.static
    for _ in range(warmup):
        graphed(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graphed(batch)
    graphed.flush()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    last = {k: round(float(v), 6) for k, v in itf.last_loss_dict.items()}
    graphed.close()
    flops = 374.0e9 * B_PER_GPU                       # 3 x 124.7 GF per patch
    peak = PEAK_BF16_MFMA_TFLOPS if ops.PRECISION != "fp32" else PEAK_FP32_MFMA_TFLOPS
    return {"workload": "BASELINE configs[1]: KPCN-Vanilla diffuse+specular (n_in=34), 128x128, batch %d, 1 GPU" % B_PER_GPU,
            "value": round(B_PER_GPU * steps / el, 3), "unit": "patches/s", "ms_per_step": round(el / steps * 1e3, 3),
            "steps": steps, "warmup": warmup, "dtype": ops.PRECISION,
            "whole_step_mfma_frac": round(flops / (el / steps) / 1e12 / peak, 4),
            "losses_last_step": last}


def extra_leg(device, steps, warmup, precision=None, group=None, force_collective=False, overlap=False, weight_norm=True, two_stream=TWO_STREAM):
    """The benchmarked step once more in another configuration, graphed, same weights (seed 0) and batch: another arithmetic
    (`other_precisions`), or the default one with the MULTI-RANK tail on a one-rank RCCL group (`multi_rank_path`)."""
    from wcmc_amd import ops
    from wcmc_amd.graph import GraphedTrainStep
    from wcmc_amd.optim import FusedClipAdam
    from wcmc_amd.synthetic import make_batch
    old = ops.PRECISION
    if precision is not None:
        ops.set_precision(precision)
    try:
        itf = build_interface(device, None, rng="device", weight_norm=weight_norm)
        if force_collective:
            itf.fused_optim = FusedClipAdam(itf.models, itf.optims, process_group=group, force_collective=True,
                                            order=("dncnn", "backbone_diffuse", "backbone_specular") if overlap else None)
        batch = make_batch(B_PER_GPU, SPP, PATCH, seed=0, device=device)
        torch.manual_seed(1234)
        graphed = GraphedTrainStep(itf, batch, overlap_allreduce=overlap, two_stream=two_stream and not overlap, defer_check=True)
        batch = graphed.static
        for _ in range(warmup):
            graphed(batch)
        if force_collective:
            graphed.tail_events = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            graphed(batch)
        graphed.flush()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out = {"value": round(B_PER_GPU * steps / el, 3), "unit": "patches/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
               "warmup": warmup, "dtype": ops.PRECISION if ops.split_path() else "f32",
               "losses_last_step": {k: round(float(v), 6) for k, v in itf.last_loss_dict.items()}}
        if force_collective:
            assert graphed.tail_split and not graphed.tail_captured
            tails = [a.elapsed_time(b) for a, b in graphed.tail_events]
            out["tail_ms"] = round(sum(tails) / len(tails), 4)
        graphed.close()                                   # (one graphed step alive at a time: GraphedTrainStep.close)
        return out
    finally:
        ops.set_precision(old)


def cpu_baseline():
    """The oracle's step (same architecture, same losses) on the host cores: C3 shape at batch 1."""
    from oracle import step as ostep
    from oracle.models import KPCN as OKPCN
    from oracle.networks import PathNet as OPathNet
    from wcmc_amd.synthetic import make_batch
    torch.manual_seed(0)
    models = {"dncnn": OKPCN(39), "backbone_diffuse": OPathNet(36), "backbone_specular": OPathNet(36)}
    optims = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-4) for k, m in models.items()}
    cfg = dict(use_llpm_buf=True, manif_learn=True, train_branches=True, disentanglement_option="m11r11",
               w_manif=0.1)

    def one(b, h):
        batch = make_batch(b, SPP, h, seed=0, device="cpu")
        ho = h - 36
        perms = [ostep.draw_perms(b, SPP, ho, ho), ostep.draw_perms(b, SPP, ho, ho)]
        t0 = time.perf_counter()
        ostep.train_step(models, optims, batch, cfg, perms)
        return time.perf_counter() - t0

    # Pick the thread count on a small patch: one thread per visible CPU is NOT the fastest on a
    # many-core host (256 logical CPUs measured 40x slower than 8 threads), so grow while it helps.
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best_n, best_t = None, None
    for n in (8, 16, 32, 64, 128, 256):
        if n > avail and best_n is not None:
            break
        torch.set_num_threads(min(n, avail))
        t = one(1, 64)
        if best_t is not None and t > 0.9 * best_t:
            break
        best_n, best_t = min(n, avail), t
    torch.set_num_threads(best_n)
    one(1, PATCH)                                    # warm-up (allocator, oneDNN primitive caches)
    # bounded sample: steps at batch 1 until ~12 s of CPU work (at most 12 steps), the mean step is reported
    times = []
    while len(times) < 12 and sum(times) < 12.0:
        times.append(one(1, PATCH))
    t = sum(times) / len(times)
    # BASELINE configs[0] (the reference's own CPU-runnable case): KPCN-Vanilla, 64x64, batch 2, 3 steps after a warm-up
    vmodels = {"dncnn": OKPCN(34)}
    voptims = {"optim_dncnn": torch.optim.Adam(vmodels["dncnn"].parameters(), lr=1e-4)}
    vbatch = make_batch(2, SPP, 64, seed=0, device="cpu", use_llpm=False)
    vcfg = dict(use_llpm_buf=False, manif_learn=False, train_branches=True)
    c1 = []
    for i in range(4):
        t0 = time.perf_counter()
        ostep.train_step(vmodels, voptims, vbatch, vcfg, None)
        c1.append(time.perf_counter() - t0)
    c1_t = sum(c1[1:]) / 3
    return {"value": 1.0 / t, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_logical_cpus": os.cpu_count(), "host_cpus_in_affinity_mask": avail,
            "sample": "%d train steps of the PyTorch-CPU oracle after one warm-up, KPCN-Manifold C3 shape at batch 1 "
                      "(128x128, S=8), %.1f s in total, %.2f s per step; `cores` = torch intra-op threads, chosen by a "
                      "scaling probe (more threads are slower on this host)" % (len(times), sum(times), t),
            "c1": {"value": 2.0 / c1_t, "unit": "64x64 patches/s", "sample": "BASELINE configs[0]: KPCN-Vanilla (n_in=34, both "
                   "branches), 64x64, batch 2, 3 train steps after one warm-up, %.2f s per step" % c1_t}}


def self_launch(n, argv):
    """``python bench.py --gpus N`` without a launcher: start N fresh worker processes (one per GPU) through
    ``torch.distributed.run`` BEFORE this process makes any GPU call, pass rank 0's JSON line through, and exit with
    the workers' status.  (Never an exec of a process that touched the GPU; this parent never does.)"""
    import socket
    import subprocess
    # preflight, before any worker exists (counting devices does not initialise the GPU on this image)
    have = torch.cuda.device_count()
    if have < n and "--share-gpu" not in argv:
        sys.exit("bench.py: --gpus %d but this node shows %d GPU(s) (rocm-smi / HIP_VISIBLE_DEVICES); nothing was launched" % (n, have))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL fails with the legacy mode on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of one hipGraph")
    ap.add_argument("--cpu-rng", action="store_true",
                    help="draw the FeatureMSE pairings on the global CPU generator like the reference (+46 ms/step)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend of the gradient all-reduce: nccl = RCCL over xGMI (the measured "
                         "configuration); gloo only to smoke-test the multi-rank path where RCCL cannot run")
    ap.add_argument("--share-gpu", action="store_true",
                    help="smoke test on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo; RCCL refuses two ranks "
                         "on one device); the printed throughput is then meaningless")
    ap.add_argument("--sync-check", action="store_true",
                    help="read the non-finite-loss flags of every step before the next one is enqueued (one host sync per step) instead of one step later")
    ap.add_argument("--one-graph", action="store_true",
                    help="the step as ONE forked hipGraph (rounds 2-4) instead of two half-step graphs on two streams + a tail graph")
    ap.add_argument("--no-pathnet-weight-norm", action="store_true",
                    help="plain nn.Conv2d weights in the PathNets instead of upstream sbmc's weight-normalised layers (the default)")
    ap.add_argument("--precision", choices=("bf16x321h", "bf16x321o", "bf16x321", "bf16x3", "fp32"), default=None,
                    help="conv GEMM arithmetic: split-bf16 with 3 / 2 / 1 MFMAs per product in forward / data gradient / weight "
                         "gradient (default), 3 everywhere (rounds 1-2), or exact fp32 MFMA (roofline vs the 157.3 TF/s peak)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, sys.argv[1:])

    from wcmc_amd import distributed as wd
    from wcmc_amd import ops
    from wcmc_amd.synthetic import make_batch
    if args.precision:
        ops.set_precision(args.precision)
    if args.share_gpu:
        assert args.backend == "gloo", "--share-gpu needs --backend gloo"
        os.environ["LOCAL_RANK"] = "0"
    if not args.share_gpu and torch.cuda.device_count() < int(os.environ.get("WORLD_SIZE", "1")):
        sys.exit("bench.py: WORLD_SIZE=%s but this node shows %d GPU(s); refusing to start (one rank per GPU)" %
                 (os.environ.get("WORLD_SIZE"), torch.cuda.device_count()))
    try:
        rank, world, local = wd.init(args.backend)
    except Exception as err:            # RCCL / rendezvous failure: a non-zero exit with the reason, never a re-exec or a silent fallback
        print("bench.py: torch.distributed (%s) failed to initialise: %r" % (args.backend, err), file=sys.stderr, flush=True)
        sys.exit(3)
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or "
                 "without a launcher: bench.py starts its own workers)" % (args.gpus, world, args.gpus))
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    group = torch.distributed.group.WORLD if world > 1 else None
    itf = build_interface(device, group, rng="cpu" if args.cpu_rng else "device", weight_norm=not args.no_pathnet_weight_norm)
    if world > 1:
        for fl in itf.fused_optim.flats.values():
            torch.distributed.broadcast(fl.flat, 0)
    batch = make_batch(B_PER_GPU, SPP, PATCH, seed=wd.shard_seed(0, rank), device=device)
    torch.manual_seed(1234 + rank)      # FeatureMSE pairings (CPU generator, losses.py:35,50)

    def eager_step():
        itf.preprocess(batch)
        itf.train_batch(batch)

    prof = EventProfiler()
    stream_defaults = (ops.USE_SIDE_STREAM, ops.USE_BRANCH_STREAM)      # (the per-kernel profile below switches them off)

    def eager_profiled_step():
        prof.next_step()
        eager_step()

    if args.eager:
        ops.USE_SIDE_STREAM = False        # per-launch events need one stream (warm-up included: the rocprofv3
        ops.USE_BRANCH_STREAM = False
        step = eager_profiled_step         # averages of `bench.py --eager` are then single-stream durations too)
    else:
        from wcmc_amd.graph import capture_validated
        # every capture is timed (10 replays behind the device guard: nothing is updated) and re-made when it is more than 5 %
        # slower than the fastest capture of this configuration the process has seen; at least two are compared
        graphed = capture_validated(itf, batch, two_stream=TWO_STREAM and not args.one_graph, defer_check=not args.sync_check)
        # the synthetic batch lives IN the step's static buffers (inputs resident in HBM: a loader assembles the next batch into them
        # at the step boundary, support/loader.py) -- no hand-over copy; the pairings are drawn inside the graph (device keys); the
        # non-finite flags of step t are read after step t + 1 has been enqueued (--sync_check: before)
        batch = graphed.static
        step = lambda: graphed(batch)

    for _ in range(args.warmup):
        step()
    if args.eager:
        ops.USE_SIDE_STREAM = False        # per-launch events need one stream
        ops.USE_BRANCH_STREAM = False
        ops.set_profiler(prof)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if not args.eager:
        graphed.flush()                    # (the deferred non-finite check of the last step)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    my_elapsed = time.perf_counter() - t0
    elapsed = wd.max_over_ranks(my_elapsed, device)
    rank_ms = wd.gather_floats(my_elapsed / args.steps * 1e3, device)        # every rank's own ms per step (rank 0 reports min / max)
    comm = wd.time_allreduce(itf.fused_optim, group, device) if world > 1 else None
    # every rank's parameters after the timed steps, as an exact integer checksum (sum of the fp32 bit patterns): data-parallel
    # replicas that saw the same summed gradients must hold the same bits
    with torch.no_grad():
        digest = sum(int(p.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())
                     for m in itf.models.values() for p in m.parameters())
    digests = wd.gather_ints(digest, device)
    last_losses = {k: float(v) for k, v in itf.last_loss_dict.items()}       # of step warmup + steps, this rank
    ops.set_profiler(None)
    prof_elapsed = elapsed
    if not args.eager:
        # The timed region above replays one hipGraph per step (no per-kernel events inside a graph).
        # Per-kernel durations for the roofline come from the same step launched eagerly right after,
        # HIP events on the launch stream around every conv / kernel-apply launch (same kernels, same shapes).
        itf.fused_optim.leave_grads = True
        itf.loss_funcs["l_manif"].static_perms = None
        itf.loss_funcs["l_manif"].check_finite = True
        nprof = max(2, min(5, args.steps))
        ops.USE_SIDE_STREAM = False        # one stream: every event pair brackets exactly one kernel class
        ops.USE_BRANCH_STREAM = False
        eager_step()
        ops.set_profiler(prof)
        torch.cuda.synchronize()
        pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prof_elapsed = 0.0
        for _ in range(nprof):
            # The eager host needs ~40 ms to enqueue a step, longer than the GPU needs to run it: keep the
            # stream busy with replays of the captured step meanwhile, so that the event pairs bracket
            # kernels that run back to back (and at the clocks of the timed region: behind a 100 ms spin
            # kernel the same launches measured 2-3x longer) instead of a GPU waiting for Python.
            prof.next_step()
            for _ in range(3):
                graphed._replay()
            pe0.record()
            eager_step()
            pe1.record()
            torch.cuda.synchronize()
            prof_elapsed += pe0.elapsed_time(pe1) * 1e-3
        ops.set_profiler(None)

    if rank == 0:
        summ = prof.summary()
        if os.environ.get("WCMC_BENCH_DEBUG"):
            for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                print("  [profile] %-22s %5d launches %9.3f ms total  avg %.4f ms" %
                      (k, d["launches"], d["ms"], d["ms"] / d["launches"]), file=sys.stderr)
            print("  [profile] eager region %.1f ms over the profiled steps" % (prof_elapsed * 1e3), file=sys.stderr)
        global_batch = B_PER_GPU * world
        value = global_batch * args.steps / elapsed

        def roof(name, bound):
            d = summ.get(name)
            if not d or d["ms"] <= 0:
                return None
            rate = d["work"] / (d["ms"] * 1e-3)
            if bound == "mfma":
                peak = PEAK_BF16_MFMA_TFLOPS if ops.split_path() else PEAK_FP32_MFMA_TFLOPS
                ach, unit = rate / 1e12, "TFLOP/s"
            else:
                ach, peak, unit = rate / 1e9, PEAK_HBM_GBS, "GB/s"
            extra = {}
            if bound == "mfma" and ops.split_path():
                # bf16 MFMAs per algorithmic multiply-add (+12 % cout and 5 % k padding on top): `frac` counts algorithmic
                # FLOPs once against the dense bf16 peak, so its ceiling is 1 / terms; for scale, the exact-fp32 MFMA peak is 157.3 TFLOP/s
                t = mfma_terms(name, ops.wgrad_terms())
                extra = {"mfma_flops_per_algorithmic_flop": t, "frac_of_issued_mfma_flops": round(ach * t / peak, 4),
                         "issued_mfma_TFLOPs": round(ach * t, 1),
                         "frac_of_fp32_mfma_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 3)}
            rocprof_name = rocprof_names(ops.wgrad_terms()).get(name, name + " (several kernels)")
            if ops.PRECISION == "fp32":
                rocprof_name = ("wcmc::conv_wgrad_kernel" if "wgrad" in name else "wcmc::conv_igemm_kernel") + \
                    " (exact fp32 MFMA; launches of class %s)" % name
            return {"kernel": rocprof_name, "class": name, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "traffic": None, "launches": d["launches"], **extra,
                    "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                    "share_of_profiled_region": round(d["ms"] / (prof_elapsed * 1e3), 4)}

        def family(label, keys):
            keys = [k for k in keys if k in summ and summ[k]["ms"] > 0]
            if not keys:
                return None
            work, ms = sum(summ[k]["work"] for k in keys), sum(summ[k]["ms"] for k in keys)
            ach = work / (ms * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS if ops.split_path() else PEAK_FP32_MFMA_TFLOPS
            extra = {}
            if ops.split_path():        # MFMA FLOPs the launches issue for their algorithmic ones (terms per product; padding not counted)
                extra = {"issued_mfma_TFLOPs": round(sum(summ[k]["work"] * mfma_terms(k, ops.wgrad_terms()) for k in keys) / (ms * 1e-3) / 1e12, 1)}
            return {"family": label, "classes": keys, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), **extra, "launches": sum(summ[k]["launches"] for k in keys),
                    "ms_per_profiled_step": round(ms / max(1, prof.step), 4),
                    "share_of_profiled_region": round(ms / (prof_elapsed * 1e3), 4)}

        # classes = kernels: conv_halo64_pt4 / _pt3 / _cs32 are the instances of conv_halo64_bf16x3_kernel<7,NB,PT> (KPCN 5x5 fwd +
        # dgrad: 16x16 tiles, 12x16 tiles, 12x16 with 32-channel slabs for the 441-channel data gradient; conv_halo7 = the 8x16
        # kernel they replace, WCMC_HALO64=0), conv_wgrad_rows is conv_wgrad_rows_bf16x3_kernel<5,7,7> for one-term launches, conv_wgrad_rows8_bf16x3_kernel for three-term ones (WCMC_WGRAD_ROWS8=0|1: one of them for both);
        # conv_igemm / conv_wgrad collect the other GEMM kernels
        conv_keys = [k for k in ("conv_halo64_pt3", "conv_halo64_pt4", "conv_halo64_pt3_x2", "conv_halo64_pt4_x2", "conv_halo64_pt3_x1", "conv_halo64_pt4_x1", "conv_halo64_pt3_h1", "conv_halo64_pt4_h1", "conv_halo64_cs32", "conv_halo7",
                                 "conv_halo3", "conv_halo3_x2", "conv_wgrad_rows", "conv_igemm", "conv_wgrad")
                     if k in summ]
        # the roofline kernel: the single kernel (one rocprof name) with the most time per step; the two catch-all classes
        # collect several kernels and are reported under roofline_other_conv (exact-fp32 mode: one kernel per class anyway)
        single = ([k for k in conv_keys if k not in ("conv_igemm", "conv_wgrad")] or conv_keys) if ops.split_path() else conv_keys
        dominant = max(single, key=lambda k: summ[k]["ms"]) if conv_keys else None
        ka = kernel_apply_probe(device)
        traffic = pmc_traffic()
        for nm in ("fwd", "bwd"):
            ka[nm]["traffic"] = traffic.get("kernel_apply_" + nm)
            d = summ.get("kernel_apply_" + nm)          # the same kernel inside the (eagerly launched, profiled) train step
            if d and d["ms"] > 0:
                gbs = d["work"] / (d["ms"] * 1e-3) / 1e9
                ka[nm]["in_step"] = {"achieved": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4), "launches": d["launches"],
                                     "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                                     "note": "HIP events around the launches of the profiled eager steps; its algorithmic bytes "
                                             "count logits + radiance + result (+ gradient) as SURVEY 8d does"}
        line = {
            "metric": "128x128 MC patches/sec (train step), KPCN-Manifold",
            "value": round(value, 3), "unit": "patches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ops.PRECISION if ops.split_path() else "f32",
            "data": "synthetic",
            # from the communicator, not from the command line: ranks of the process group whose backend is RCCL ("nccl" on ROCm)
            "rccl_ranks": (torch.distributed.get_world_size(group) if world > 1 and torch.distributed.get_backend(group) == "nccl"
                           else 1 if world == 1 and args.backend == "nccl" else 0),
            "collective_backend": ("rccl" if args.backend == "nccl" else args.backend + (" (smoke test, shared GPU)" if args.share_gpu else "")),
            "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
            "params_identical_across_ranks": (len(set(digests)) == 1) if world > 1 else None,
            # capture validation (wcmc_amd.graph.capture_validated), this rank: ms per replay of every capture that was made; the last one is the step that ran
            "capture_attempts": None if args.eager else graphed.capture_attempts,
            "capture_ms": None if args.eager else graphed.capture_ms,
            "capture_validated": None if args.eager else getattr(graphed, "capture_validated", None),
            # the two streams the halves replay on: picked once per process by a spin-kernel probe so that they sit on different
            # hardware queues (wcmc_amd.ops.concurrent_stream_pair)
            "stream_pair": (getattr(graphed.half_streams[0], "probe", None) if (not args.eager and graphed.two_stream) else None),
            "allreduce": comm,
            # loss_dict of the last timed step on rank 0 (seeded weights, inputs and pairings: reproducible run to run
            # with the same binary; a stream-ordering race in the captured step would show here)
            "losses_last_step": {k: round(v, 6) for k, v in last_losses.items()},
            "config": {"workload": "BASELINE configs[2]: KPCN-Manifold (KPCN n_in=39 + 2xPathNet 36->3 + "
                                   "FeatureMSE w=0.1 m11r11, train_branches), 128x128, S=8 spp, "
                                   "%d patches/GPU, global batch %d" % (B_PER_GPU, global_batch),
                       "global_batch": global_batch, "parallelism": "dp%d" % world,
                       "graph_form": None if args.eager else ("two half-step hipGraphs on two streams + tail graph" if graphed.two_stream
                                                               else "one forked hipGraph"),
                       "launch": ("eager" if args.eager else
                                  "one hipGraph replay per step, optimiser tail (finite check, loss sums, gradient gather, clip + Adam) captured in it"
                                  if graphed.tail_captured else
                                  "two hipGraph replays per step around three eager asynchronous all-reduces of the gradient buckets: graph A = forward, "
                                  "backward, gradient hand-over, guard flag; graph B = global guard, loss sums, scale -> clip -> Adam"
                                  if graphed.tail_split else
                                  "one hipGraph replay per step (forward + backward) + eager all-reduce of three buckets + eager clip + Adam"),
                       "feature_mse_rng": "cpu (reference stream)" if args.cpu_rng else "device",
                       # initial weights and synthetic patches: the seed that draws live output channels in BOTH P-buffers, scenes with
                       # object-scale contrast whose path descriptors carry the radiance (both manifold terms stay alive)
                       "weights_seed": WEIGHT_SEED, "synthetic_patches": "scene (wcmc_amd/synthetic.py, round 6)",
                       # PathNet layers as sbmc.modules.ConvChain builds them when support/networks.py:18-24 passes no weight_norm
                       # argument: w = g * v / ||v|| (wcmc_weight_norm_fwd / _bwd, one launch per PathNet and direction)
                       "pathnet_weight_norm": not args.no_pathnet_weight_norm,
                       "precision": (("conv GEMMs: split-bf16 operands (hi + lo planes), v_mfma_f32_16x16x32_bf16, fp32 accumulate; per "
                                      "product 3 MFMAs in the forward (hi*hi + hi*lo + lo*hi)%s, 2 in the data gradients that have a "
                                      "two-term instance (dy_hi x (W_hi + W_lo): KPCN 5x5, U-Net 3x3; the fused 1x1 chains' too), 1 in the "
                                      "weight gradients (dy_hi x x_hi) -- the rungs of profiles/r03_precision_ladder.txt (backward) and "
                                      "profiles/r04_forward_ladder.txt (forward) that hold every parity bar.  roofline counts algorithmic "
                                      "FLOPs once against the dense bf16 MFMA peak, so frac <= 1 / (MFMAs per product); everything else fp32") %
                                     (" except the two un-gated KPCN output layers (100 -> 441 logits), which run ONE (%s: no "
                                      "ReLU behind them, so no gate can flip; every HIDDEN layer needs >= 16-bit operands to hold the "
                                      "gradient bars)" % ("x_hi x W_hi, bf16" if ops.PRECISION == "bf16x321o" else "fp16(x) x fp16(W)")
                                      if ops.PRECISION in ("bf16x321o", "bf16x321h") else
                                      "; outputs and losses are bit-identical to the all-three-term mode (--precision bf16x3)"))
                       if ops.reduced_backward() else
                       ("conv GEMMs: split-bf16 operands (hi+lo), 3 x v_mfma_f32_16x16x32_bf16 per product in every GEMM, fp32 "
                        "accumulate (frac <= 1/3); everything else fp32") if ops.PRECISION == "bf16x3"
                       else "fp32 MFMA (v_mfma_f32_16x16x4_f32), fp32 accumulate"},
            # the whole step against the matrix peak: 549.6 GF algorithmic per patch / step time / dense peak of the mode
            "whole_step": {"algorithmic_tflop_per_step": round(549.6e9 * B_PER_GPU / 1e12, 3),
                           "achieved_tflops": round(549.6e9 * global_batch * args.steps / elapsed / 1e12 / world, 1),
                           "frac_of_mfma_peak": round(549.6e9 * global_batch * args.steps / elapsed / 1e12 / world /
                                                      (PEAK_BF16_MFMA_TFLOPS if ops.split_path() else PEAK_FP32_MFMA_TFLOPS), 4)},
            "roofline": dict(roof(dominant, "mfma"), traffic=traffic.get(dominant)) if dominant else None,
            "roofline_other_conv": [dict(roof(k, "mfma"), traffic=traffic.get(k)) for k in conv_keys if k != dominant],
            # families: every instance of one kernel template (and, for the U-Net, what is left of it in the catch-all class) summed --
            # algorithmic FLOPs of all their launches / their HIP-event time -- so that the figure does not depend on how the template
            # arguments split the time
            "roofline_family": [f for f in (family("conv_halo64 (KPCN 5x5 forward + data gradient, every instance)",
                                                   [k for k in conv_keys if k.startswith("conv_halo64")]),
                                            family("U-Net 3x3 forward + data gradient (conv_halo3 instances + class conv_igemm)",
                                                   [k for k in conv_keys if k.startswith("conv_halo3") or k == "conv_igemm"]),
                                            family("weight gradients (conv_wgrad_rows + class conv_wgrad)",
                                                   [k for k in conv_keys if k.startswith("conv_wgrad")])) if f],
            "roofline_pointwise": (dict(roof("conv_pw", "hbm"), traffic=traffic.get("conv_pw")) if roof("conv_pw", "hbm") else None),
            # PathNet.embedding / PathNet.final as one launch per direction (hidden activations never leave the CU): HBM-bound by
            # construction, algorithmic bytes = input + output (+ gradients) once
            "roofline_pathnet_fused": [dict(roof(k, "hbm"), traffic=traffic.get(k))
                                       for k in ("embed3_fwd", "embed3_bwd", "final2_fwd", "final2_bwd") if roof(k, "hbm")],
            "roofline_kernel_apply": ka,
        }
        line["measured_peaks"] = measured_peaks(device)
        # the practical matrix peak of THIS box: what a dense 8192^3 bf16 hipBLASLt GEMM sustains (clock under MFMA load, not the 2.5 PF
        # of the data sheet); `issued_vs_measured_gemm` = MFMA FLOPs a conv kernel issues per second (terms per product, padding not
        # counted) / that rate -- how far the kernel is from what the matrix pipe delivers here, where `frac` is against the spec peak
        gemm = line["measured_peaks"].get("bf16_gemm_8192_TFLOPs")
        if gemm and ops.split_path() and ops.reduced_backward() and prof.step:
            # MFMA FLOPs ONE step issues: the conv classes' algorithmic FLOPs x MFMAs per product (HIP-event classes of the profiled
            # steps), + the fused 1x1 PathNet chains by their shapes (forward 3, data gradient 2, weight gradient 1 MFMAs per product;
            # their recomputation in the backward and every padding NOT counted) -- against the step time and the measured GEMM rate
            conv = sum(summ[k]["work"] * mfma_terms(k, ops.wgrad_terms()) for k in conv_keys) / prof.step
            px = B_PER_GPU * SPP * PATCH * PATCH
            emb, fin = 36 * 64 + 64 * 64 + 64 * 64, 128 * 128 + 128 * 3
            fused = 2 * 2.0 * px * (emb * 3 + (64 * 64 * 2) * 2 + emb * 1 + fin * 6)
            issued = (conv + fused) / 1e12
            line["whole_step"]["issued_mfma_tflop_per_step"] = round(issued, 3)
            line["whole_step"]["issued_mfma_TFLOPs"] = round(issued / (elapsed / args.steps), 1)
            line["whole_step"]["issued_vs_measured_gemm"] = round(issued / (elapsed / args.steps) / gemm, 3)
        if gemm:
            for obj in [line["roofline"]] + line["roofline_other_conv"] + line["roofline_family"]:
                if obj and obj.get("issued_mfma_TFLOPs"):
                    obj["issued_vs_measured_gemm"] = round(obj["issued_mfma_TFLOPs"] / gemm, 3)
        if world == 1 and not args.eager:
            ops.USE_SIDE_STREAM, ops.USE_BRANCH_STREAM = stream_defaults
            itf.fused_optim.leave_grads = False            # (back to the captured buffers: the long segment replays the graph)
            # a longer segment of the SAME graphed step right behind the official one (the driver's 20 steps are 0.27 s of a 20 s
            # process: a 5-second utilisation sampler sees nothing of them)
            nlong = max(100, 5 * args.steps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nlong):
                graphed(batch)
            graphed.flush()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            line["value_long"] = {"value": round(B_PER_GPU * nlong / el, 3), "unit": "patches/s", "steps": nlong,
                                  "ms_per_step": round(el / nlong * 1e3, 3), "seconds": round(el, 3),
                                  "note": "the same graphed step, %d more timed steps behind the official ones" % nlong}
            graphed.close()
            line["c2"] = c2_leg(device, args.steps, args.warmup)
            if args.precision is None:
                # the other arithmetics the library ships, same box, same process
                line["other_precisions"] = {m: extra_leg(device, n, 3, precision=m)
                                            for m, n in (("bf16x321", args.steps), ("bf16x321o", args.steps), ("bf16x3", args.steps),
                                                         ("fp32", max(3, args.steps // 4)))}
                # ... and the other PathNet parametrisation (plain weights when the headline is weight-normalised, and vice versa)
                line["other_parametrisation"] = dict(extra_leg(device, args.steps, 3, weight_norm=bool(args.no_pathnet_weight_norm)),
                                                     pathnet_weight_norm=bool(args.no_pathnet_weight_norm))
            if args.backend == "nccl":
                # The step exactly as rank k of N runs it -- graph A (forward, backward, gradient gather, guard flag), three eager
                # asynchronous RCCL all-reduces of the gradient buckets (46.8 MB), graph B (global guard, sums, scale -> clip ->
                # Adam) -- on a ONE-rank RCCL group: the same-box baseline of the first multi-GPU run.  What N ranks add to it is
                # the wire time of the buckets (the `allreduce` object of an N-rank line) and rank skew; what it shows here is
                # what the split tail itself costs against the single captured graph.  No scaling curve has been measured.
                try:
                    if not torch.distributed.is_initialized():
                        torch.distributed.init_process_group("nccl", store=torch.distributed.HashStore(), rank=0, world_size=1)
                    mr = extra_leg(device, args.steps, args.warmup, group=torch.distributed.group.WORLD, force_collective=True)
                    mr["what"] = ("graph A | 3 async RCCL all-reduces (one-rank group) | graph B; tail_ms = HIP events around the "
                                  "all-reduces + graph B")
                    mr["rccl_ranks"] = torch.distributed.get_world_size()
                    mr["predicted_weak_scaling_ceiling"] = round(line["ms_per_step"] / mr["ms_per_step"], 4)
                    mr["ceiling_note"] = ("single-graph step time / multi-rank-path step time on this box: an upper bound of the "
                                          "N-rank efficiency before any wire time or skew; no scaling curve has been measured")
                    # the same with the backward cut at the P-buffers (three graphs: the dncnn bucket is on the wire while the PathNets'
                    # backward runs) -- what the overlap COSTS on one rank (its benefit needs a wire to hide)
                    ov = extra_leg(device, args.steps, args.warmup, group=torch.distributed.group.WORLD, force_collective=True, overlap=True)
                    mr["overlap_allreduce"] = {"value": ov["value"], "ms_per_step": ov["ms_per_step"], "tail_ms": ov["tail_ms"],
                                               "losses_last_step": ov["losses_last_step"],
                                               "what": "graph A1 (... dncnn backward) | async all-reduce of the dncnn bucket || graph A2 (PathNet "
                                                       "backward) | all-reduces of the PathNet buckets | graph B; GraphedTrainStep(overlap_allreduce=True), "
                                                       "off by default"}
                    line["multi_rank_path"] = mr
                    torch.distributed.destroy_process_group()
                except Exception as err:                       # (reported, never fatal for the headline)
                    line["multi_rank_path"] = {"error": repr(err)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
