import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
for i in range(3):
    r = bench.kernel_apply_probe(torch.device("cuda", 0), iters=50)
    print({k: (v["avg_launch_ms"], v["frac"]) for k, v in r.items()})
