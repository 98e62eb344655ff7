"""Kernel-apply probe of bench.py (cold Infinity Cache: rotating buffer sets), strip kernel variants vs the tile kernel.
   python3 scripts/time_ka.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
for tile in ("1", "0", "0", "1"):
    os.environ["WCMC_KA_TILE"] = tile
    r = bench.kernel_apply_probe(torch.device("cuda", 0), iters=48)
    print("tile " if tile == "1" else "strip", {k: (v["avg_launch_ms"], v["frac"]) for k, v in r.items()}, flush=True)
