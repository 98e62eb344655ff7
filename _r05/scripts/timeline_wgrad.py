"""Wall-clock timeline and phase cycles of the filter-row weight-gradient kernel (debug build, WCMC_DEBUG_ABLATE=16):
   make -C wcmc_amd/csrc debug; WCMC_DEBUG_LIB=1 WCMC_DEBUG_ABLATE=16 python3 scripts/timeline_wgrad.py [h ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
TERMS = int(os.environ.get("WCMC_WGRAD_TERMS", "1"))      # bf16 MFMAs per product of the weight gradient: 1 (hi planes only) or 3
from wcmc_amd import ops as o
from wcmc_amd.ops import _ptr, _stream, lib, check
assert os.environ.get("WCMC_DEBUG_ABLATE") == "16"
NWV = 7 if os.environ.get("WCMC_WGRAD_ROWS8", "1")[:1] == "0" else 8     # waves per block of the instance that runs
dev = "cuda"
n, cin, cout, ks = 8, 100, 100, 5
for h in [int(a) for a in sys.argv[1:]] or [124, 108]:
    ho = h - ks + 1
    xs = o.split_raw(o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev)))
    dys = o.split_raw(o.to_nhwc_raw(torch.randn(n, cout, ho, ho, device=dev)))
    nbytes = lib().wcmc_conv2d_wgrad_bf16x3_workspace_bytes(n, ho, ho, cout, cin, ks)
    ws = torch.zeros((nbytes + 3) // 4, device=dev); dw = torch.empty(cout, cin, ks, ks, device=dev); db = torch.empty(cout, device=dev)
    args = (_ptr(xs), n, h, h, cin, _ptr(dys), cout, ks, 0, _ptr(dw), _ptr(db), _ptr(ws), ws.numel() * 4)
    for _ in range(3):
        check(lib().wcmc_conv2d_wgrad_bf16x3(*args, 1, None, TERMS, _stream()), "wgrad")
    torch.cuda.synchronize()
    # slabs: S x 25 x 112 x 112 floats; the stamps sit behind them
    S = (nbytes // 4 - 0) // (25 * 112 * 112)
    raw = ws.cpu().numpy()
    for S_try in range(S, 0, -1):
        st = raw[S_try * 25 * 112 * 112:].view(np.uint64)
        nb = ((S_try + 7) // 8) * 8 * 5
        rec = st[: nb * NWV * 8].reshape(nb, NWV, 8)
        live = rec[:, 0, 7] > 0
        if live.sum() == S_try * 5 and rec[live][:, 0, 7].max() < 1000:
            break
    rec = rec[live].astype(np.float64)
    rt = rec[:, :, 0:4] * 0.01
    rt -= rt[:, :, 0].min()
    print("h=%d: %d blocks (S=%d), %d stages each; launch span %.1f us" % (h, len(rec), S_try, rec[0, 0, 7], rt[:, :, 3].max()))
    print("  entry %.1f..%.1f  first stage issued +%.1f  stage loop %.1f (%.1f..%.1f)  slab write %.1f  exit %.1f..%.1f" %
          (rt[:, :, 0].min(), rt[:, :, 0].max(), (rt[:, :, 1] - rt[:, :, 0]).mean(), (rt[:, :, 2] - rt[:, :, 1]).mean(),
           (rt[:, :, 2] - rt[:, :, 1]).min(), (rt[:, :, 2] - rt[:, :, 1]).max(), (rt[:, :, 3] - rt[:, :, 2]).mean(),
           rt[:, :, 3].min(), rt[:, :, 3].max()))
    cyc = rec[:, :, 4:7]
    tot = cyc.sum(axis=2)
    print("  stage loop cycles per stage: %.0f = wait (DMA + barrier) %.0f + issue %.0f + MFMA/fragment reads %.0f   (%d MFMAs x 2 k-steps x 16 = %d)" %
          (tot.mean() / rec[0, 0, 7], cyc[:, :, 0].mean() / rec[0, 0, 7], cyc[:, :, 1].mean() / rec[0, 0, 7], cyc[:, :, 2].mean() / rec[0, 0, 7],
           105 if NWV == 7 else 93, 3360 if NWV == 7 else 2976))
    for w in range(NWV):
        print("    wave %d: wait %.0f issue %.0f mfma %.0f" % (w, cyc[:, w, 0].mean() / rec[0, 0, 7], cyc[:, w, 1].mean() / rec[0, 0, 7], cyc[:, w, 2].mean() / rec[0, 0, 7]))
