"""Same-box A/B of a switch of the captured step: alternating captures with the switch off / on, best of 3 timings each.
    python3 scripts/ab_step_switch.py FUSE_FINAL [rounds]          a Python-level switch: ops.<NAME> = False / True
    python3 scripts/ab_step_switch.py ENV:WCMC_HALO3 [rounds]      a kernel switch of the DEBUG library (read per launch at capture time):
                                                                   the environment variable = "0" / "1"; loads libwcmc_hip_debug.so"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
name_ = sys.argv[1]
if name_.startswith("ENV:"):
    os.environ["WCMC_DEBUG_LIB"] = "1"
import bench
from wcmc_amd import ops
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch

name = sys.argv[1]
ENV = name.startswith("ENV:")
if ENV:
    os.environ["WCMC_DEBUG_LIB"] = "1"
    name = name[4:]
VALS = ("0", "1")
if "=" in name:                                            # ENV:NAME=off,on
    name, v = name.split("=")
    VALS = tuple(v.split(","))
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
res = {False: [], True: []}
# (order off-on-on-off per pair of rounds: the chip warms up over the first minute of a process and a fixed order would charge the
# drift to whichever arm comes second)
for r in range(rounds):
    for val in ((False, True) if r % 2 == 0 else (True, False)):
        if ENV:
            os.environ[name] = VALS[1 if val else 0]
        else:
            setattr(ops, name, val)
        itf = bench.build_interface(dev, None, rng="device")
        batch = make_batch(bench.B_PER_GPU, bench.SPP, bench.PATCH, seed=0, device=dev)
        torch.manual_seed(1234)
        if ENV and VALS != ("0", "1"):
            itf.loss_funcs["l_manif"].check_finite = False   # (timing-only ablations: garbage losses)
        if os.environ.get("AB_SERIAL") == "1":                # one stream: the halves in series (what a kernel is worth alone on the chip)
            ops.USE_BRANCH_STREAM = False
        step = GraphedTrainStep(itf, batch, two_stream=os.environ.get("AB_SERIAL") != "1", defer_check=True)
        if ENV and VALS != ("0", "1"):
            step._check = lambda *a, **k: None
        b = step.static
        ts = []
        for rep in range(3):
            for _ in range(5):
                step(b)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                step(b)
            step.flush()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 40 * 1e3)
        res[val].append(min(ts))
        print("round %d  %s = %-5s  %.3f ms per step" % (r, name, val, min(ts)), flush=True)
        step._pending = None
        step.close()
        del step, itf
print("%s: off %.3f ms (mean of %d), on %.3f ms" % (name, sum(res[False]) / len(res[False]), rounds, sum(res[True]) / len(res[True])))
