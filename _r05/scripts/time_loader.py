"""Staged patch loader (wcmc_amd/support/loader.py): images/s, patches/s and PCIe GB/s with raw renderer output in host
memory, alone and feeding the graphed train step.   python3 scripts/time_loader.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import make_golden as mg
import bench
from wcmc_amd.support.loader import PatchLoader
dev = torch.device("cuda", 0)
if os.environ.get("WCMC_SWITCH"):                      # experiment: the interpreter's thread switch interval (default 5 ms)
    sys.setswitchinterval(float(os.environ["WCMC_SWITCH"]))
WORKERS = int(os.environ.get("WCMC_LOADER_WORKERS", "2"))
H = W = 512; S = 8
images = [{"raw": mg.raw_samples(H, W, S, 10 + i), "gt": np.random.rand(H, W, 9).astype(np.float32), "prob": None} for i in range(2)]
reader = lambda i: images[i % 2]
n_img = 6
loader = PatchLoader(reader, range(n_img), dev, batch_size=8, patch_size=128, workers=WORKERS)
print("reader / staging workers: %d, switch interval %.4f s" % (WORKERS, sys.getswitchinterval()))
for _ in loader: pass                                   # warm-up (pinned allocations)
torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
for b in loader: nb += 1
torch.cuda.synchronize(); t = time.perf_counter() - t0
gb = n_img * (images[0]["raw"].nbytes + images[0]["gt"].nbytes) / 1e9
print("loader alone: %d images of %dx%dx%d spp (%.2f GB raw) -> %d batches of 8 in %.3f s = %.0f patches/s, %.1f GB/s over PCIe"
      % (n_img, H, W, S, gb, nb, t, nb * 8 / t, gb / t))
itf = bench.build_interface(dev, None, rng="device")
from wcmc_amd.graph import GraphedTrainStep
first = next(iter(loader))
step = GraphedTrainStep(itf, first)
if os.environ.get("WCMC_LOADER_PACE", "1") != "0":
    step.after_enqueue = loader.kick                    # the producer assembles the next batch while this step runs on the GPU
for b in loader: step(b)                                # warm-up
# where the difference to resident inputs goes: (a) the same step on one resident batch, (b) on that batch while the loader
# runs beside it (its batches are drawn and dropped), (c) fed by the loader
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(96): step(first)
torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / 96
def epoch(run_step, feed):
    """One pass over the loader; the clock starts when the FIRST batch is there (an epoch's start-up -- thread start, the first
    image read, staged over PCIe and preprocessed: ~50 ms, once per epoch whatever its length -- is reported on its own)."""
    torch.cuda.synchronize(); s0 = time.perf_counter()
    it = iter(loader); b = next(it)
    torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
    while b is not None:
        run_step(b if feed else first); nb += 1
        b = next(it, None)
    if hasattr(run_step, "flush"): run_step.flush()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    return (t1 - t0) / nb, nb, t0 - s0
tb, nb, start = epoch(step, False)
print("graphed step on a resident batch: %.2f ms; the same with the loader running beside it: %.2f ms (epoch start-up %.0f ms, not included)"
      % (ta * 1e3, tb * 1e3, start * 1e3))
t, nb, start = epoch(step, True)
print("loader feeding the graphed KPCN-Manifold step: %d steps at %.3f ms = %.1f patches/s = %.3f of the resident step (+ %.0f ms of start-up per epoch)"
      % (nb, t * 1e3, 8 / t, ta / t, start * 1e3))
step2 = GraphedTrainStep(itf, first, defer_check=True)
if os.environ.get("WCMC_LOADER_PACE", "1") != "0":
    step2.after_enqueue = loader.kick
for b in loader: step2(b)
step2.flush()
t, nb, start = epoch(step2, True)
print("the same with defer_check=True (non-finite check of step t after step t + 1 is enqueued): %.1f patches/s = %.3f of the resident step" % (8 / t, ta / t))
