"""How far is fp32 itself from exact arithmetic on the benchmarked step?  One KPCN-Manifold step at the bench shape
(B patches of 128x128, S=8) three ways -- CPU oracle in fp64 (the yardstick), CPU oracle in fp32 (what the reference
computes), HIP path (split-bf16, graph-free) -- and the per-tensor relative L2 of the two fp32 gradients against fp64.
   python3 scripts/diag_grad_floor.py [B]"""
import copy, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(min(32, os.cpu_count() or 1))
from oracle import step as ostep
from oracle.models import KPCN as OKPCN
from oracle.networks import PathNet as OPathNet
from wcmc_amd import KPCN
from wcmc_amd.support.interfaces import KPCNInterface
from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
from wcmc_amd.support.networks import PathNet
from wcmc_amd.synthetic import make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
    hm[k].load_state_dict(o32[k].state_dict()); hm[k].to("cuda")
batch = make_batch(B, 8, 128, seed=40, device="cpu")
cfg = dict(use_llpm_buf=True, manif_learn=True, train_branches=True, disentanglement_option="m11r11", w_manif=0.1)
torch.manual_seed(1234)
perms = [ostep.draw_perms(B, 8, 92, 92), ostep.draw_perms(B, 8, 92, 92)]
mk = lambda ms: {"optim_" + k: torch.optim.SGD(m.parameters(), lr=0.0) for k, m in ms.items()}
ostep.train_step(o32, mk(o32), batch, cfg, perms)
ostep.train_step(o64, mk(o64), {k: v.double() for k, v in batch.items()}, cfg, perms)
lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
      "l_test": RelativeMSE(), "l_manif": FeatureMSE(non_local=True)}
itf = KPCNInterface(hm, mk(hm), lf, types.SimpleNamespace(model_name="d"), use_llpm_buf=True, manif_learn=True,
                    w_manif=0.1, train_branches=True)
itf.iters = 1; itf.to_train_mode()
torch.manual_seed(1234)
db = {k: v.to("cuda") for k, v in batch.items()}
itf.preprocess(db); itf.train_batch(db)
rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
rows = []
for mn in o32:
    for (k, p32), (_, p64), (_, ph) in zip(o32[mn].named_parameters(), o64[mn].named_parameters(), hm[mn].named_parameters()):
        rows.append((mn + " " + k, rl2(p32.grad, p64.grad), rl2(ph.grad.cpu(), p64.grad), rl2(ph.grad.cpu(), p32.grad)))
print("%-74s %10s %10s %10s" % ("tensor (B=%d)" % B, "o32-o64", "hip-o64", "hip-o32"))
for r in sorted(rows, key=lambda r: -r[2])[:12]:
    print("%-74s %10.2e %10.2e %10.2e" % r)
import statistics as st
print("median over %d tensors: oracle fp32 vs fp64 %.2e, HIP vs fp64 %.2e, HIP vs oracle fp32 %.2e" %
      (len(rows), st.median(r[1] for r in rows), st.median(r[2] for r in rows), st.median(r[3] for r in rows)))
print("max: oracle fp32 vs fp64 %.2e, HIP vs fp64 %.2e, HIP vs oracle fp32 %.2e" %
      (max(r[1] for r in rows), max(r[2] for r in rows), max(r[3] for r in rows)))
