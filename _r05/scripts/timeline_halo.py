"""Wall-clock timeline of the halo-resident 5x5 kernel's workgroups (s_memrealtime stamps of the debug build):
   make -C wcmc_amd/csrc debug; WCMC_DEBUG_LIB=1 WCMC_DEBUG_ABLATE=64 python3 scripts/timeline_halo.py [h ...]
Per workgroup: entry -> stage loop -> end of loop -> exit, and the CU it ran on; prints the round structure of a launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wcmc_amd import ops as o
assert os.environ.get("WCMC_DEBUG_ABLATE") == "64"
dev = "cuda"
unet = "--unet" in sys.argv                      # a U-Net 3x3 layer (64 -> 64, pad 1) instead of a KPCN 5x5 layer
n, cin, cout, ks = (8, 64, 64, 3) if unet else (8, 100, 100, 5)
pad = 1 if unet else 0
for h in [int(a) for a in sys.argv[1:] if a != "--unet"] or [124, 116, 100]:
    x = o.to_nhwc_raw(torch.randn(n, cin, h, h, device=dev))
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    xs = o.split_raw(x); wp = o._pack_x(w, 0)
    for _ in range(3):
        y, part = o.conv2d_x_raw(xs, (n, cin, h, h), wp, b, cout, ks, pad, "relu", out_split=True, colsum=True)
    torch.cuda.synchronize()
    ho = h + 2 * pad - ks + 1
    th = 8 if unet else 16 if os.environ.get("WCMC_HALO64", "1") != "0" else 8      # conv_halo64: 16x16 tiles; conv_halo<7,8,16>: 8x16
    tiles, nw = n * ((ho + th - 1) // th) * ((ho + 15) // 16), (1 if unet else 4)     # (64 couts: room for one wave's record per tile)
    st = part.cpu().numpy().view(np.uint64)[: tiles * nw * 14].reshape(tiles, nw, 14)
    rt = st[:, :, 6:13].astype(np.float64) * 0.01          # us: entry, loop start, loop end, E0, E1, E2, exit
    t0 = rt[:, :, 0].min()
    rt -= t0
    hw = st[:, 0, 13]
    cu = ((hw >> 32) & 0xf) * 1024 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)
    ent, lp0, lp1, ex = rt[:, :, 0].min(1), rt[:, :, 1].max(1), rt[:, :, 2].max(1), rt[:, :, 6].max(1)
    e0, e1, e2 = rt[:, :, 3].max(1), rt[:, :, 4].max(1), rt[:, :, 5].max(1)
    print("h=%d: %d tiles on %d CUs, launch span %.1f us (entry of the first workgroup -> exit of the last)" %
          (h, tiles, len(np.unique(cu)), ex.max()))
    first = ent < 5.0
    for nm, m in (("round 1 (entry < 5 us)", first), ("later", ~first)):
        if m.sum() == 0:
            continue
        print("  %-22s %4d tiles: entry %6.1f..%6.1f  prologue %5.1f  loop %5.1f  epilogue %5.1f  exit %6.1f..%6.1f (mean %.1f)" %
              (nm, m.sum(), ent[m].min(), ent[m].max(), (lp0 - ent)[m].mean(), (lp1 - lp0)[m].mean(), (ex - lp1)[m].mean(),
               ex[m].min(), ex[m].max(), ex[m].mean()))
        if th == 8:
            print("      epilogue: drain+barrier %.1f  act/split -> LDS %.1f  store issue %.1f  store drain %.1f" %
                  ((e0 - lp1)[m].mean(), (e1 - e0)[m].mean(), (e2 - e1)[m].mean(), (ex - e2)[m].mean()))
        else:
            print("      epilogue: operand loads + drain + barrier %.1f  first half: act/split -> LDS %.1f, store issue %.1f  second half + store drain %.1f" %
                  ((e0 - lp1)[m].mean(), (e1 - e0)[m].mean(), (e2 - e1)[m].mean(), (ex - e2)[m].mean()))
    if th == 8:      # the 8x16 kernel's s_memtime phase sums of the stage loop (cycles; the stamps themselves cost ~100 each)
        ph = st[:, :, 0:6].astype(np.float64)
        names = ["stage barrier", "halo DMA issue (slab ends)", "MFMAs + fragment reads", "stage tail (+ slab boundaries)",
                 "wait for the wave's own weight DMA", "weight DMA issue"]
        nstage = (ks * ks * (cin if unet else 32) + 31) // 32 if unet else 82
        print("  stage loop, cycles per stage (%d stages): " % nstage + ", ".join("%s %.0f" % (nm, ph[:, :, i].mean() / nstage) for i, nm in enumerate(names))
              + "; MFMA issue alone: %d" % ((4 if unet else 7) * 2 * 3 * 16))
    # occupancy over time: how many workgroups are inside their stage loop
    grid = np.arange(0, ex.max(), 5.0)
    inloop = [(int(((lp0 <= t) & (lp1 > t)).sum()), int(((ent <= t) & (ex > t)).sum())) for t in grid]
    print("  t (us): in-loop/resident  " + "  ".join("%d:%d/%d" % (t, a, r) for t, (a, r) in zip(grid, inloop)))
    xcc, se = (hw >> 32) & 0xf, (hw >> 13) & 7
    print("  stage-loop time by XCD: " + "  ".join("%d: %.1f (%.1f..%.1f)" % (x, (lp1 - lp0)[xcc == x].mean(), (lp1 - lp0)[xcc == x].min(),
                                                                            (lp1 - lp0)[xcc == x].max()) for x in np.unique(xcc)))
    x0 = xcc == np.unique(xcc)[0]
    print("  ... inside the first XCD by shader engine: " + "  ".join("%d: %.1f" % (e, (lp1 - lp0)[x0 & (se == e)].mean()) for e in np.unique(se[x0])))
    ucu, inv = np.unique(cu, return_inverse=True)
    percu = np.array([(lp1 - lp0)[inv == i].mean() for i in range(len(ucu))])
    print("  per-CU mean stage-loop time: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" %
          (percu.min(), np.percentile(percu, 10), np.median(percu), np.percentile(percu, 90), percu.max()))
    # does the tile's position matter?  (tile index -> image row / column of the tile)
    tcol = np.arange(tiles) % ((ho + 15) // 16)
    print("  stage-loop time by tile column: " + "  ".join("%.1f" % (lp1 - lp0)[tcol == c].mean() for c in np.unique(tcol)))
    # which workgroups share a CU?  (block index b <-> tile: the kernel's XCD-aware remap)
    nbk = tiles; qq, rr = nbk >> 3, nbk & 7
    blk_of_tile = np.zeros(tiles, dtype=np.int64)
    for bidx in range(nbk):
        x, kk = bidx & 7, bidx >> 3
        blk_of_tile[(x * (qq + 1) if x < rr else rr * (qq + 1) + (x - rr) * qq) + kk] = bidx
    pairs = []
    for i in range(min(len(ucu), 6)):
        t = np.nonzero(inv == i)[0]
        pairs.append(" ".join("b%d(l%d):%.0f" % (blk_of_tile[j], blk_of_tile[j] >> 3, (lp1 - lp0)[j]) for j in t))
    print("  workgroups per CU (block, local index in its XCD, stage-loop us): " + " | ".join(pairs))
    per = np.bincount(np.unique(cu, return_inverse=True)[1])
    print("  tiles per CU: min %d max %d" % (per.min(), per.max()))
