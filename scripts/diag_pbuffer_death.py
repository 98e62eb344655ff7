"""Why does the specular P-buffer of the benchmark recipe die?  Per step: share of the P-buffer's entries that
are > 0, its mean, the manifold terms -- for the recipe as it is and for variants of the synthetic specular target.
    python3 scripts/diag_pbuffer_death.py [steps] [seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wcmc_amd import synthetic
from wcmc_amd.synthetic import make_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seeds = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
dev = torch.device("cuda", 0)
for variant in (os.environ.get("VARIANTS", "recipe_r5,scene").split(",")):
    for seed in seeds:
        itf = bench.build_interface(dev, None, rng="device", seed=seed)
        torch.manual_seed(1234 + seed)
        batches = [make_batch(bench.B_PER_GPU, bench.SPP, bench.PATCH, seed=100 * seed + i, device=dev, scene=(variant != "recipe_r5")) for i in range(4)]
        rows = []
        for st in range(steps):
            b = batches[st % len(batches)]
            itf.preprocess(b)
            itf.train_batch(b)
            if st < 6 or st % 20 == 19:
                with torch.no_grad():
                    ps = itf.models["backbone_specular"](b)
                    pd = itf.models["backbone_diffuse"](b)
                ld = {k: float(v) for k, v in itf.last_loss_dict.items()}
                rows.append("step %3d  P_spec > 0: %.3f mean %.3e max %.3e | P_diff > 0: %.3f mean %.3e | l_manif d %.3e s %.3e | l_spec %.4f" % (
                    st, (ps > 0).float().mean().item(), ps.mean().item(), ps.max().item(), (pd > 0).float().mean().item(), pd.mean().item(),
                    ld.get("l_manif_diffuse", float("nan")), ld.get("l_manif_specular", float("nan")), ld.get("l_specular", float("nan"))))
        print("== %s, seed %d" % (variant, seed))
        print("\n".join(rows), flush=True)
        tgt = batches[0]["target_specular"]
        print("   target_specular: mean %.3f std %.3f (per-image spatial std %.3f)" % (tgt.mean().item(), tgt.std().item(), tgt.std(dim=(2, 3)).mean().item()))
        tgt = batches[0]["target_diffuse"]
        print("   target_diffuse:  mean %.3f std %.3f (per-image spatial std %.3f)" % (tgt.mean().item(), tgt.std().item(), tgt.std(dim=(2, 3)).mean().item()))
        del itf
