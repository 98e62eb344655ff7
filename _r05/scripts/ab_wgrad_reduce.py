"""Weight gradients of the benchmark's layer shapes from the library in use, saved for a bit-for-bit comparison of two
builds, and the time of the slab reduction (phase 2 of the call):
   python3 scripts/ab_wgrad_reduce.py out.pt ; WCMC_LIB_AB=libwcmc_hip_prev.so python3 scripts/ab_wgrad_reduce.py prev.pt
   python3 scripts/ab_wgrad_reduce.py --compare out.pt prev.pt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
TERMS = int(os.environ.get("WCMC_WGRAD_TERMS", "1"))      # bf16 MFMAs per product of the weight gradient: 1 (hi planes only) or 3
if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        print("%-28s dw %s db %s   finish %6.1f us vs %6.1f us" % (k, torch.equal(a[k][0], b[k][0]), torch.equal(a[k][1], b[k][1]), a[k][2], b[k][2]))
    sys.exit(0)
from wcmc_amd import ops as o
from wcmc_amd.ops import _ptr, _stream, lib, check
dev = "cuda"
LAYERS = [(8, 100, 100, 5, 0, 124), (8, 100, 441, 5, 0, 96), (8, 39, 100, 5, 0, 128), (16, 64, 64, 3, 1, 128), (16, 128, 128, 3, 1, 64),
          (16, 256, 256, 3, 1, 32), (16, 192, 64, 3, 1, 128), (16, 67, 64, 3, 1, 128), (64, 36, 64, 1, 0, 128), (64, 128, 3, 1, 0, 128),
          (3, 100, 100, 5, 0, 37), (2, 20, 24, 3, 1, 19)]
out = {}
for n, cin, cout, ks, pad, h in LAYERS:
    torch.manual_seed(h + cin)
    ho = h + 2 * pad - ks + 1
    xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev)))
    dys = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, ho, ho, device=dev)))
    nbytes = lib().wcmc_conv2d_wgrad_bf16x3_workspace_bytes(n, ho, ho, cout, cin, ks)
    ws = torch.empty((nbytes + 3) // 4, device=dev)
    dw = torch.empty(cout, cin, ks, ks, device=dev); db = torch.empty(cout, device=dev)
    call = lambda ph: check(lib().wcmc_conv2d_wgrad_bf16x3(_ptr(xs), n, h, h, cin, _ptr(dys), cout, ks, pad, _ptr(dw), _ptr(db), _ptr(ws),
                                                            ws.numel() * 4, ph, None, TERMS, _stream()), "wgrad")
    call(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): call(2)
    e1.record(); torch.cuda.synchronize()
    out["%dx%d^2 %d->%d %dx%d" % (n, h, cin, cout, ks, ks)] = (dw.cpu(), db.cpu(), e0.elapsed_time(e1) / 20 * 1e3)
torch.save(out, sys.argv[1])
