"""Timing-only ablations of the strip kernel-apply forward (debug library: make -C wcmc_amd/csrc debug).
   WCMC_DEBUG_LIB=1 python3 scripts/time_ka_abl.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
os.environ["WCMC_KA_TILE"] = "0"
for ab, name in (("0", "full"), ("1", "stream only (1/7 of the arithmetic)"), ("2", "arithmetic only (no logits stream)"), ("0", "full")):
    os.environ["WCMC_DEBUG_ABLATE"] = ab
    r = bench.kernel_apply_probe(torch.device("cuda", 0), iters=48)
    print(name, r["fwd"]["avg_launch_ms"], flush=True)
