"""Is the captured step power-limited?  (a) the step's time with the model's weights as drawn and with ALL parameters zero (the
same kernels and launches on operands that toggle nothing: MI355X_MICROARCH.md, DVFS give-back -- zero-filled inputs ran +19 %);
(b) rocm-smi's clock / power samples while the step replays back to back.
    python3 scripts/probe_power.py"""
import os, sys, time, subprocess, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wcmc_amd.graph import GraphedTrainStep
from wcmc_amd.synthetic import make_batch

dev = torch.device("cuda", 0)
samples = []
stop = False

def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
            keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "Power", "power", "Temperature (Sensor junction)", "mclk"))]
            samples.append(" | ".join(keep))
        except Exception as e:
            samples.append("rocm-smi failed: %r" % e)
        time.sleep(0.3)

for zero in (False, True, False, True):
    itf = bench.build_interface(dev, None, rng="device")
    itf.loss_funcs["l_manif"].check_finite = False
    if zero:
        with torch.no_grad():
            for m in itf.models.values():
                for p in m.parameters():
                    p.zero_()
    batch = make_batch(bench.B_PER_GPU, bench.SPP, bench.PATCH, seed=0, device=dev)
    if zero:
        batch = {k: torch.zeros_like(v) for k, v in batch.items()}
    torch.manual_seed(1234)
    step = GraphedTrainStep(itf, batch, two_stream=True, defer_check=True)
    step._check = lambda *a, **k: None
    b = step.static
    for _ in range(20):
        step(b)
    torch.cuda.synchronize()
    samples.clear()
    stop = False
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        step(b)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n * 1e3
    stop = True
    th.join()
    print("%-28s %.3f ms per step over %d replays" % ("ALL-ZERO weights and inputs" if zero else "weights and inputs as drawn", el, n), flush=True)
    for s_ in samples[1:6]:
        print("     " + s_)
    step._pending = None
    step.close()
    del step, itf
