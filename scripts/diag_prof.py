"""Why do the bench's per-launch igemm events read longer than rocprof's durations?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wcmc_amd import ops
from wcmc_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
itf = bench.build_interface(dev, None)
batch = make_batch(8, 8, 128, seed=0, device=dev)
ops.USE_SIDE_STREAM = False
def step():
    itf.preprocess(batch); itf.train_batch(batch)
for _ in range(3): step()
def timed(label, n=3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    print("%-28s gpu %.1f ms/step  host+gpu %.1f ms/step" % (label, e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n))
timed("eager, no profiler")
prof = bench.EventProfiler(); ops.set_profiler(prof)
timed("eager, op profiler")
s = prof.summary()
for k, d in sorted(s.items(), key=lambda kv: -kv[1]["ms"])[:8]:
    print("   %-22s %5d launches %8.2f ms/step  avg %.3f ms" % (k, d["launches"], d["ms"] / 3, d["ms"] / d["launches"]))
ops.set_profiler(None)
# --- as bench.py: captured step, then the eager step behind replays
from wcmc_amd.graph import GraphedTrainStep
graphed = GraphedTrainStep(itf, batch)
for _ in range(3): graphed(batch)
itf.fused_optim.leave_grads = True
itf.loss_funcs["l_manif"].static_perms = None
itf.loss_funcs["l_manif"].check_finite = True
ops.USE_SIDE_STREAM = False
step()
for mode in ("replays", "none"):
    prof = bench.EventProfiler(); ops.set_profiler(prof)
    torch.cuda.synchronize()
    for _ in range(3):
        if mode == "replays":
            for _ in range(3): graphed.graph.replay()
        step()
        torch.cuda.synchronize()
    ops.set_profiler(None)
    s = prof.summary()
    print("after GraphedTrainStep, eager step behind", mode)
    for k, d in sorted(s.items(), key=lambda kv: -kv[1]["ms"])[:4]:
        print("   %-22s %5d launches %8.2f ms/step  avg %.3f ms" % (k, d["launches"], d["ms"] / 3, d["ms"] / d["launches"]))
