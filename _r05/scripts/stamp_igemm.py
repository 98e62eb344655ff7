"""Phase shares of the split-bf16 implicit-GEMM main loop from in-kernel stamps (WCMC_DEBUG_ABLATE=64).
   make -C wcmc_amd/csrc debug; WCMC_DEBUG_LIB=1 WCMC_DEBUG_ABLATE=64 python3 scripts/stamp_igemm.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wcmc_amd import ops as o
assert os.environ.get("WCMC_DEBUG_ABLATE") == "64"
dev = "cuda"
n, cin, h, cout, ks = 8, 100, 116, 100, 5
x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
b = torch.zeros(cout, device=dev)
xs = o.split_raw(x); wp = o._pack_x(w, 0)
for _ in range(3):
    y, part = o.conv2d_x_raw(xs, (n, cin, h, h), wp, b, cout, ks, 0, "relu", out_split=True, colsum=True)
torch.cuda.synchronize()
ho = h - ks + 1
halo = os.environ.get("WCMC_IGEMM_HALO", "1") != "0"
if halo:      # the shipped 8x16 tiling: 4 waves, 32-channel slabs (32 + 32 + 40 of the 104 padded channels)
    tiles, nw, nstage = n * ((ho + 7) // 8) * ((ho + 15) // 16), 4, 2 * ((ks * ks * 32 + 31) // 32) + ((ks * ks * 40 + 31) // 32)
else:
    tiles, nw, nstage = (n * ho * ho + 127) // 128, 4, (ks * ks * 104 + 31) // 32
st = part.cpu().numpy().view(np.uint64)[: tiles * nw * 8].reshape(tiles, nw, 8).astype(np.float64)
names = ["load issue", "frag reads+wait", "mfma issue", "vmcnt+lds store", "barrier", "slab boundary"]
if halo:
    names = ["stage barrier", "halo DMA issue (slab ends)", "mfma + fragment reads (drained)", "stage tail (+ slab boundaries)",
             "wait for own weight DMA (vmcnt)", "weight DMA issue outside the MFMA stream"]
NB = 6 if halo else 5
tot = st[:, :, :NB].sum(axis=2)
print("tiles", tiles, "stages/tile", nstage)
print("cycles per tile (s_memtime ticks = 100 MHz?): mean %.0f min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
for i, nm in enumerate(names[:NB]):
    print("  %-34s %5.1f %%   (per stage %.0f ticks)" % (nm, 100 * st[:, :, i].sum() / tot.sum(), st[:, :, i].mean() / nstage))
for wv in range(nw):
    print("  wave %d:" % wv, " ".join("%5.1f" % (100 * st[:, wv, i].sum() / tot[:, wv].sum()) for i in range(NB)))
