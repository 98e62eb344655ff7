"""What this MI355X sustains on plain streams (context for the roofline fractions in DESIGN.md section 4).
   python3 scripts/hbm_probe.py"""
import torch
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (128, 537, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda").normal_(); y = torch.empty_like(x)
    t_copy = timeit(lambda: y.copy_(x))
    t_read = timeit(lambda: x.sum())
    t_write = timeit(lambda: y.fill_(1.0))
    print("%5d MB: copy %.2f TB/s (read+write bytes)   read-only sum %.2f TB/s   write-only fill %.2f TB/s" %
          (mb, 2 * n * 4 / t_copy / 1e12, n * 4 / t_read / 1e12, n * 4 / t_write / 1e12))
