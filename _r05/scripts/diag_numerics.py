"""GPU diagnostic: print the actual relative errors of each op against fp64 references."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from wcmc_amd import ops as o
from oracle import modules as om
torch.set_num_threads(16)
DEV = "cuda"
def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()
for (n, cin, h, w, cout, ks, pad, act) in [(2, 100, 24, 24, 100, 5, 0, "relu"), (2, 39, 24, 24, 100, 5, 0, "linear"),
                                           (1, 100, 16, 16, 441, 5, 0, "linear"), (2, 64, 16, 16, 64, 3, 1, "relu"),
                                           (2, 36, 16, 16, 64, 1, 0, "relu")]:
    x = gen(n, cin, h, w, seed=2); wt = gen(cout, cin, ks, ks, seed=3, scale=(2.0 / (cin * ks * ks)) ** 0.5 * 1.7)
    b = gen(cout, seed=4, scale=0.2)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, wt, b))
    yr = om._activation(F.conv2d(xr, wr, br, padding=pad), act)
    gy = gen(*yr.shape, seed=5); yr.backward(gy.double())
    x32, w32, b32 = (t.clone().requires_grad_(True) for t in (x, wt, b))
    y32 = om._activation(F.conv2d(x32, w32, b32, padding=pad), act); y32.backward(gy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, wt, b))
    y = o.conv_chain(xd, ks, pad, [act], [wd, bd]); y.backward(gy.to(DEV))
    print("conv", (n, cin, h, w, cout, ks, pad, act))
    print("   hip : fwd %.2e dx %.2e dw %.2e db %.2e" % (rel(y, yr), rel(xd.grad, xr.grad), rel(wd.grad, wr.grad), rel(bd.grad, br.grad)))
    print("   cpu32: fwd %.2e dx %.2e dw %.2e db %.2e" % (rel(y32, yr), rel(x32.grad, xr.grad), rel(w32.grad, wr.grad), rel(b32.grad, br.grad)))
for (n, c, h, w, k) in [(2, 3, 28, 28, 21), (1, 3, 20, 20, 5)]:
    data = gen(n, c, h, w, seed=30) + 0.5
    for sc in (1.0, 3.0, 10.0):
        logits = gen(n, k * k, h, w, seed=31, scale=sc)
        dr, lr = data.double().requires_grad_(True), logits.double().requires_grad_(True)
        outr = om.kernel_apply(dr, lr); g = gen(*outr.shape, seed=32); outr.backward(g.double())
        dd, ld = data.to(DEV).requires_grad_(True), logits.to(DEV).requires_grad_(True)
        out = o.kernel_apply(dd, ld); out.backward(g.to(DEV))
        print("kernel_apply k=%d scale %.0f: fwd %.2e dlogits %.2e ddata %.2e" % (k, sc, rel(out, outr), rel(ld.grad, lr.grad), rel(dd.grad, dr.grad)))
