import sys, os; sys.argv=['x']
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from wcmc_amd import ops as o
from wcmc_amd.ops import _ptr, _stream, lib, check
dev='cuda'
def case(n, cin, h, cout, ks):
    ho = h - ks + 1
    xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev)))
    dys = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, ho, ho, device=dev)))
    nbytes = lib().wcmc_conv2d_wgrad_bf16x3_workspace_bytes(n, ho, ho, cout, cin, ks)
    ws = torch.zeros((nbytes + 3) // 4, device=dev); dw = torch.empty(cout, cin, ks, ks, device=dev); db = torch.empty(cout, device=dev)
    args = (_ptr(xs), n, h, h, cin, _ptr(dys), cout, ks, 0, _ptr(dw), _ptr(db), _ptr(ws), ws.numel() * 4)
    for _ in range(12): check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 1, _stream()), "w")
    torch.cuda.synchronize()
    R = n * ho; rps = -(-R // 51); S = -(-R // rps); SLAB = S * 25 * 112 * 112
    st = ws.cpu().numpy()[SLAB:].view(np.uint64)[:280 * 4].reshape(280, 4).astype(np.float64)
    st = st[st[:, 2] > 0]
    mhz = st[:, 0] / st[:, 1] * 100.0
    print("n %2d out %3d: blocks %3d stages %3d  loop %.1f us  clock %.0f MHz (min %.0f max %.0f)  cycles/stage %.0f" %
          (n, ho, len(st), st[0, 2], st[:, 1].mean() / 100.0, mhz.mean(), mhz.min(), mhz.max(), (st[:, 0] / st[:, 2]).mean()))
for n, h in ((8, 100), (16, 100), (2, 128), (8, 128), (8, 68)):
    case(n, 100, h, 100, 5)
