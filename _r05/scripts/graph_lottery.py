"""Does a graphed step built LATER in a process run as fast as the first one?  Builds, times and closes the benchmark step
again and again (bench.py's extra legs are the 3rd .. 9th graphs of their process; some of them run at the rate of
serialised halves).   python3 scripts/graph_lottery.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
batch = make_batch(8, 8, 128, seed=0, device=dev)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    itf = bench.build_interface(dev, None, rng="device")
    st = GraphedTrainStep(itf, batch)
    for _ in range(5): st(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(25): st(batch)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 25 * 1e3
    print("graph %2d: %.3f ms per step" % (i, ms), flush=True)
    st.close()
    del st, itf
