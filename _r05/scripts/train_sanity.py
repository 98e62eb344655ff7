"""Sanity: 120 graphed steps on one fixed synthetic batch -- the losses must stay finite and fall.
   python3 scripts/train_sanity.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch

dev = torch.device("cuda", 0)
itf = bench.build_interface(dev, None)
for g in itf.optims.values():
    g.param_groups[0]["lr"] = 1e-4
batch = make_batch(8, 8, 128, seed=0, device=dev)
torch.manual_seed(1)
step = GraphedTrainStep(itf, batch)
prev = None
for i in range(120):
    step(batch)
    if i % 20 == 19:
        cur = {k: float(v) / 20 for k, v in itf.m_losses.items()}
        print(i + 1, {k: round(v, 5) for k, v in cur.items()}, flush=True)
        for k in itf.m_losses:
            itf.m_losses[k].zero_()
        if prev is not None:
            assert cur["m_l_total"] < prev["m_l_total"] * 1.05, "loss is not falling"
        prev = cur
print("ok")
