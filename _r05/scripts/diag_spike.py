"""Why does l_manif_diffuse of the default-mode trajectory spike for ONE step (step 194 of scripts/train_trajectory.py: 4.3e-4 against
5.7e-5 before and after, other arithmetics flat)?  Replays the run to that step, then evaluates the SAME step (same weights, batch,
pairings) eagerly with the fused PathNet chains on / off and in bf16x3.
   python3 scripts/diag_spike.py [STEP]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wcmc_amd import ops
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch

STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 194
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 16
MODE = sys.argv[3] if len(sys.argv) > 3 else ops.MODES[0]
ops.set_precision(MODE)
dev = torch.device("cuda", 0)
itf = bench.build_interface(dev, None, rng="device")
batches = [make_batch(8, 8, 128, seed=500 + i, device=dev) for i in range(NB)]
step = GraphedTrainStep(itf, batches[0])
torch.manual_seed(1234)
fo = itf.fused_optim
for i in range(STEP - 1):
    step(batches[i % NB])
print("step %d (graph): %s" % (STEP - 1, {k: round(float(v), 7) for k, v in step.losses.items()}))
snap = {n: fl.flat.clone() for n, fl in fo.flats.items()}
step(batches[(STEP - 1) % NB])
print("step %d (graph): %s" % (STEP, {k: round(float(v), 7) for k, v in step.losses.items()}))
perms = [(a.clone(), b.clone()) for a, b in step.perms]
after = {n: fl.flat.clone() for n, fl in fo.flats.items()}
fm = itf.loss_funcs["l_manif"]


def eager(tag):
    for n, fl in fo.flats.items():
        fl.flat.copy_(snap[n])
    fm.static_perms, fm._static_i = perms, 0
    b = {k: v.clone() for k, v in batches[(STEP - 1) % NB].items()}
    loss = itf._forward_backward(b)
    torch.cuda.synchronize()
    print("%-28s %s" % (tag, {k: round(float(v), 7) for k, v in loss.items()}), flush=True)
    o = itf.last_out
    print("      max |diffuse out| %.4g  max |specular out| %.4g  max diffuse buffer %.4g  max target_diffuse %.4g" %
          (float(o["diffuse"].abs().max()), float(o["specular"].abs().max()), float(b["kpcn_diffuse_buffer"].abs().max()), float(b["target_diffuse"].abs().max())))
    return itf.last_out


eager("eager, mode of the run (%s)" % MODE)
ops.set_precision(ops.MODES[0])
ops.FUSE_EMBED = False
eager("eager, embed unfused")
ops.FUSE_FINAL = False
eager("eager, both unfused")
ops.FUSE_EMBED = ops.FUSE_FINAL = True
ops.set_precision("bf16x3")
eager("eager, bf16x3")
ops.set_precision("fp32")
eager("eager, fp32")
ops.set_precision(MODE)
# same weights, the graph again (is the spike reproducible on replay?)
for n, fl in fo.flats.items():
    fl.flat.copy_(snap[n])
for t, (a, b) in zip(step.perms, perms):
    pass
step.graph.replay()
torch.cuda.synchronize()
print("graph replayed on the snapshot: %s" % {k: round(float(v), 7) for k, v in step.losses.items()})
