"""What is a kernel family worth AT THE STEP LEVEL?  The captured two-stream step with one family's launches REMOVED (timing only:
their outputs stay uninitialised, the losses are garbage, the device guard skips the update) against the full step.  The
difference is the step time that family is responsible for once the two halves overlap -- which is not its share of the
eagerly summed kernel time: what a latency-bound kernel leaves idle, the other half's kernels use.

    python3 scripts/marginal_step_value.py > profiles/r06_marginal_step_value.txt
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wcmc_amd import _lib, ops
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch

real = _lib.lib()


class Proxy:
    """The ctypes library with some entry points turned into no-ops (return 0)."""
    def __init__(self, skip):
        self.skip = skip

    def __getattr__(self, name):
        fn = getattr(real, name)
        rule = self.skip.get(name)
        if rule is None:
            return fn
        def wrapped(*a):
            return 0 if rule(a) else fn(*a)
        return wrapped


ALWAYS = lambda a: True
families = {
    "nothing removed": {},
    "KPCN 5x5 forward + data gradient": {"wcmc_conv2d_igemm_bf16x3": lambda a: a[13] == 5, "wcmc_conv2d_out_f16": ALWAYS},
    "U-Net 3x3 forward + data gradient": {"wcmc_conv2d_igemm_bf16x3": lambda a: a[13] == 3},
    "KPCN 5x5 weight gradient (GEMM + reduction)": {"wcmc_conv2d_wgrad_bf16x3": lambda a: a[7] == 5},
    "U-Net 3x3 weight gradient (GEMM; reductions stay)": {"wcmc_conv2d_wgrad_bf16x3": lambda a: a[7] == 3},
    "slab reductions of the small layers (reduce_multi)": {"wcmc_conv2d_wgrad_reduce_multi": ALWAYS},
    "fused PathNet chains (embed3, final2; fwd + bwd)": {n: ALWAYS for n in _lib.SIGNATURES if ("embed3" in n or "final2" in n) and "supported" not in n and "bytes" not in n and "elems" not in n},
    "kernel apply (fwd + bwd)": {"wcmc_kernel_apply_fwd": ALWAYS, "wcmc_kernel_apply_bwd": ALWAYS},
    "FeatureMSE (fwd + bwd)": {n: ALWAYS for n in _lib.SIGNATURES if n.startswith("wcmc_feature_mse")},
    "weight packing": {"wcmc_conv2d_pack_chain_bf16x3": ALWAYS, "wcmc_conv2d_pack_weight_bf16x3": ALWAYS},
    "all conv GEMMs (5x5, 3x3: fwd, dgrad, wgrad)": {"wcmc_conv2d_igemm_bf16x3": lambda a: a[13] >= 3, "wcmc_conv2d_out_f16": ALWAYS,
                                                    "wcmc_conv2d_wgrad_bf16x3": lambda a: a[7] >= 3},
}
only = sys.argv[1:] or list(families)
dev = torch.device("cuda", 0)
print("# step time of the captured two-stream step (ms, 30 replays after 5) with one family's launches removed; one MI355X")
base = None
for name in only:
    skip = families[name]
    _lib._lib = Proxy(skip) if skip else real
    itf = bench.build_interface(dev, None, rng="device")
    itf.loss_funcs["l_manif"].check_finite = False         # (garbage in, garbage out: timing only)
    batch = make_batch(bench.B_PER_GPU, bench.SPP, bench.PATCH, seed=0, device=dev)
    torch.manual_seed(1234)
    step = GraphedTrainStep(itf, batch, two_stream=True, defer_check=True)
    step._check = lambda *a, **k: None                     # (garbage losses: no host check)
    b = step.static
    ts = []
    for rep in range(3):
        for _ in range(5):
            step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step(b)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    t = min(ts)
    if base is None:
        base = t
    print("%-55s %7.3f ms   (%+.3f ms, %+.1f %%)" % (name, t, t - base, (t - base) / base * 100), flush=True)
    step._pending = None
    step.close()
    del step, itf
    _lib._lib = real
