"""The graphed benchmark step of THIS process's environment (A/B of WCMC_* / HIP runtime switches between processes):
   WCMC_SIDE_STREAM=1 python3 scripts/time_step_env.py label        -> five rounds of 30 steps, first four l_total values"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
batch = make_batch(8, 8, 128, seed=0, device=dev)
itf = bench.build_interface(dev, None, rng="device")
st = GraphedTrainStep(itf, batch, defer_check="defer" in sys.argv)
torch.manual_seed(3)
ls = []
for _ in range(4):
    st(batch); ls.append(float(st.losses["l_total"]))
v = []
for rnd in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): st(st.static if "static" in sys.argv else batch)      # ("static": no hand-over copy -- the caller fills the graph's own input buffers)
    st.flush(); torch.cuda.synchronize(); v.append((time.perf_counter() - t0) / 30 * 1e3)
print(sys.argv[1:], " ".join("%.3f" % x for x in v), "ms -> %.1f patches/s" % (8e3 / sorted(v)[2]), [repr(x) for x in ls])
