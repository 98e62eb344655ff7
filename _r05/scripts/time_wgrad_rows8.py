"""Eight-wave vs seven-wave filter-row weight gradient of the KPCN 5x5 layers (WCMC_WGRAD_ROWS8) and the priority
hand-over point of the eight-wave kernel (WCMC_WGRAD_ROWS8_PRIO), interleaved inside one process, with the bit-identity
of the results:  [PRIOS=0,8,9,10] [XES=0,1] python3 scripts/time_wgrad_rows8.py [h ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
TERMS = int(os.environ.get("WCMC_WGRAD_TERMS", "1"))      # bf16 MFMAs per product of the weight gradient: 1 (hi planes only) or 3
from wcmc_amd import ops as o
from wcmc_amd.ops import _ptr, _stream, lib, check
dev = "cuda"
hs = [int(a) for a in sys.argv[1:]] or [124, 116, 108, 100]
prios = os.environ.get("PRIOS", "").split(",") if os.environ.get("PRIOS") else [None]
xes = os.environ.get("XES", "").split(",") if os.environ.get("XES") else [None]     # WCMC_WGRAD_ROWS8_XE values to compare
cfgs = [("0", None, None)] + [("1", q, x) for q in prios for x in xes]
n, cin, cout, ks = 8, 100, 100, 5
for h in hs:
    ho = h - ks + 1
    torch.manual_seed(h)
    xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev)))
    dys = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, ho, ho, device=dev)))
    nbytes = lib().wcmc_conv2d_wgrad_bf16x3_workspace_bytes(n, ho, ho, cout, cin, ks)
    ws = torch.empty((nbytes + 3) // 4, device=dev)
    outs = {}
    def run(cfg, phase=1):
        os.environ["WCMC_WGRAD_ROWS8"] = cfg[0]
        if cfg[1] is not None: os.environ["WCMC_WGRAD_ROWS8_PRIO"] = cfg[1]
        if cfg[2] is not None: os.environ["WCMC_WGRAD_ROWS8_XE"] = cfg[2]
        dw = torch.empty(cout, cin, ks, ks, device=dev); db = torch.empty(cout, device=dev)
        check(lib().wcmc_conv2d_wgrad_bf16x3(_ptr(xs), n, h, h, cin, _ptr(dys), cout, ks, 0, _ptr(dw), _ptr(db), _ptr(ws),
                                              ws.numel() * 4, phase, None, TERMS, _stream()), "wgrad")
        return dw, db
    def once(sw, reps=10):
        run(sw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): run(sw)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    outs = [run(c, 0) for c in cfgs]
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0][0], q[0]) and torch.equal(outs[0][1], q[1]) for q in outs[1:])
    res = {c: [] for c in cfgs}
    for rnd in range(8):
        for c in cfgs:
            res[c].append(once(c))
    print("h=%d  rows7 %.1f us  rows8 %s  (the split-K kernel alone: phase 1 of the call)  bit-identical: %s"
          % (h, np.median(res[cfgs[0]]), "  ".join("%s%s%.1f us" % ("" if c[1] is None else "prio=%s " % c[1], "" if c[2] is None else "xe=%s " % c[2], np.median(res[c])) for c in cfgs[1:]), same))
