"""bench.py's sequence of graphs in one process -- main step, C2 step, then the precision legs -- repeated, each timed: which of
them run at the rate of serialised halves?   python3 scripts/graph_lottery2.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    out = []
    out.append(("c2", bench.c2_leg(dev, 20, 3)["ms_per_step"]))
    for m in ("bf16x321", "bf16x321h", "bf16x321o", "bf16x3"):
        out.append((m, bench.extra_leg(dev, 20, 3, precision=m)["ms_per_step"]))
    print("round %d: " % r + "  ".join("%s %.2f" % kv for kv in out), flush=True)
