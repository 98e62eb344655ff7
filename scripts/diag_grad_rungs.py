"""Which GEMM role carries the gradient error of a badly conditioned draw?

The benchmarked step's parameter gradients against the fp32 CPU oracle (tests/test_gpu_bench_config.py::parity_report), seed by seed,
with the backward rungs of ONE chain family widened at a time (ops.TERMS_BY_KS: filter size -> (weight-gradient, data-gradient) MFMAs
per product; ks 5 = the KPCN chains, ks 3 = the U-Net, ks 1 = PathNet.embedding / final, which run layer by layer when their fused
kernels are switched off), beside the all-three-term mode and exact fp32 MFMA.

    python3 scripts/diag_grad_rungs.py [seeds, e.g. 2,0] > profiles/r06_grad_rungs.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

torch.set_num_threads(min(16, os.cpu_count() or 1))

if __name__ == "__main__":
    seeds = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "2,0").split(",")]
    import test_gpu_bench_config as t
    from wcmc_amd import ops
    default = ops.MODES[0]
    fine = len(sys.argv) > 2 and sys.argv[2] == "fine"
    variants = [("1x1 chains layer by layer", default, {}, False),
                ("final: wgrad 3", default, {(1, 128): (3, 2)}, False),
                ("final: dgrad 3", default, {(1, 128): (1, 3)}, False),
                ("final: both 3", default, {(1, 128): (3, 3)}, False),
                ("embedding: wgrad 3", default, {(1, 36): (3, 2)}, False),
                ("embedding: dgrad 3", default, {(1, 36): (1, 3)}, False),
                ("embedding: both 3", default, {(1, 36): (3, 3)}, False),
                ("1x1 layer by layer, wgrad 3", default, {1: (3, 2)}, False),
                ("1x1 layer by layer, both 3", default, {1: (3, 3)}, False)] if fine else [("default", default, {}, True),
                ("U-Net wgrad 3", default, {3: (3, 2)}, True),
                ("U-Net dgrad 3", default, {3: (1, 3)}, True),
                ("U-Net both 3", default, {3: (3, 3)}, True),
                ("KPCN dgrad 3", default, {5: (1, 3)}, True),
                ("KPCN wgrad 3", default, {5: (3, 2)}, True),
                ("1x1 chains layer by layer", default, {}, False),
                ("1x1 layer by layer, both 3", default, {1: (3, 3)}, False),
                ("PathNet all 3 (1x1 + U-Net)", default, {1: (3, 3), 3: (3, 3)}, False),
                ("bf16x3", "bf16x3", {}, True),
                ("fp32", "fp32", {}, True)]
    print("# worst parameter-gradient tensors of the benchmarked step (two steps, weight-normalised PathNets) against the fp32 CPU oracle")
    for seed in seeds:
        for name, mode, terms, fuse in variants:
            ops.set_precision(mode)
            os.environ["WCMC_PRECISION"] = mode
            ops.TERMS_BY_KS = dict(terms)
            fe, ff = ops.FUSE_EMBED, ops.FUSE_FINAL
            ops.FUSE_EMBED, ops.FUSE_FINAL = (fe and fuse), (ff and fuse)
            try:
                # (parity_report asserts the default switches; the fused-chain switches are part of what is varied here)
                report, _ = t.parity_report("device", True, seed=seed, pin_defaults=False, bars=(1.0, 1.0))
            finally:
                ops.FUSE_EMBED, ops.FUSE_FINAL = fe, ff
                ops.TERMS_BY_KS = {}
            grads = [r for r in report if " grad " in r[0]]
            per = {}
            for r in grads:
                fam = "dncnn" if " dncnn " in r[0] else ("diffuse PathNet" if "backbone_diffuse" in r[0] else "specular PathNet")
                per[fam] = max(per.get(fam, (0.0, "")), (r[1], r[0]))
            e = max(grads, key=lambda r: r[1])
            print("seed %d  %-32s worst %.3e (1 - cos %.2e) %s" % (seed, name, e[1], e[2], e[0]))
            print("        " + "   ".join("%s %.2e" % (k, v[0]) for k, v in sorted(per.items())), flush=True)
