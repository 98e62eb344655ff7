"""Per-layer event timing of the conv launches of one eager training step (single stream).
   python3 scripts/profile_layers.py"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wcmc_amd import ops
from wcmc_amd.synthetic import make_batch

dev = torch.device("cuda", 0)
ops.USE_SIDE_STREAM = False
ops.USE_BRANCH_STREAM = False
real = ops.lib()
rows = []

class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if name == "wcmc_conv2d_igemm_bf16x3":
            def wrapped(*a):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); rc = fn(*a); e1.record()
                n, h, w, cin, cout, ks, pad = a[1], a[2], a[3], a[4], a[12], a[13], a[14]
                rows.append(("igemm", (n, h, cin, cout, ks, pad), e0, e1, 2.0 * n * min((h + 2 * pad - ks + 1) ** 2, h * h) * cin * cout * ks * ks))
                return rc
            return wrapped
        if name == "wcmc_conv2d_wgrad_bf16x3":
            def wrapped(*a):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); rc = fn(*a); e1.record()
                n, h, w, cin, cout, ks, pad = a[1], a[2], a[3], a[4], a[6], a[7], a[8]
                rows.append(("wgrad", (n, h, cin, cout, ks, pad), e0, e1, 2.0 * n * (h + 2 * pad - ks + 1) ** 2 * cin * cout * ks * ks))
                return rc
            return wrapped
        return fn

ops.lib = lambda: Proxy()
itf = bench.build_interface(dev, None)
batch = make_batch(8, 8, 128, seed=0, device=dev)
for it in range(3):
    rows.clear()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); itf.preprocess(batch); itf.train_batch(batch); t1.record()
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, key, e0, e1, fl in rows:
    d = agg.setdefault((kind, key), [0, 0.0, 0.0])
    d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e3; d[2] += fl
tot = {"igemm": 0.0, "wgrad": 0.0}
print("%-6s %-32s %5s %10s %9s" % ("kind", "(n,h,cin,cout,ks,pad)", "calls", "total us", "TF/s"))
for (kind, key), (c, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot[kind] += us
    print("%-6s %-32s %5d %10.1f %9.1f" % (kind, key, c, us, fl / us / 1e6))
print("eager step %.1f ms; igemm %.2f ms, wgrad %.2f ms" % (t0.elapsed_time(t1), tot["igemm"] / 1e3, tot["wgrad"] / 1e3))
