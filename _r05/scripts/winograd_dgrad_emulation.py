"""Winograd F(2x2, 5x5) for the KPCN DATA GRADIENT, emulated before any kernel is written.

The data gradient of a valid 5x5 convolution is a full 5x5 correlation of dy (zero-padded by 4) with the flipped, channel-swapped
filter: the same GEMM shape as the forward.  Winograd F(2x2, 5x5) computes each 2x2 output tile from a 6x6 input tile with 36
products per (cin, cout) pair instead of 100: 2.8x fewer MFMAs on 2.1 ms of the step -- IF the transformed operands survive the
rounding a kernel would apply.  The default mode's data gradient multiplies dy_hi (bf16: 8 significant bits) by W_hi + W_lo (16
bits): two MFMAs per product.  This script forms, on the CPU in fp64 / fp32, for one 100 -> 100 layer at a reduced spatial size:

  exact      fp64 direct correlation
  direct-2   the shipped rung: dy rounded to bf16, W to 16 bits (hi + lo), fp32 accumulation
  wino-fp32  Winograd with fp32 transforms and NO operand rounding: the transform's own conditioning
  wino-2     transformed dy rounded to bf16 (one plane), transformed W to 16 bits: the two-MFMA kernel one would build (0.72 MFMAs / product)
  wino-3     transformed dy to 16 bits (hi + lo) as well: three MFMAs per product (1.08 / product: no saving left; for scale)
  wino-2s    wino-2 with per-tile-position scaling of the transformed dy (rows of B^T have very different gains)

and prints the relative L2 error of the data gradient against `exact`.  Interpolation points 0, +-1, +-2 (scaled), infinity --
the standard choice for six points; `--points` takes others.  Decision rule (from profiles/r03_precision_ladder.txt): the shipped
rung sits at ~2e-3 per GEMM on i.i.d. operands and holds the network-level bar with ~2x margin; a candidate must not be worse.

    python3 scripts/winograd_dgrad_emulation.py > profiles/r05_winograd_dgrad.txt        (CPU only, ~1 min)
"""
import sys

import numpy as np
import torch

torch.manual_seed(0)
np.set_printoptions(precision=4, suppress=True)


def cook_toom(m, r, pts):
    """Winograd matrices AT (m x n), G (n x r), BT (n x n), n = m + r - 1, for points `pts` (n - 1 finite ones + infinity):
    y = AT [(G g) . (BT d)]  (1-D correlation of d (n) with g (r), m outputs)."""
    n = m + r - 1
    a = np.array(pts, dtype=np.float64)
    assert len(a) == n - 1
    # evaluation matrices of polynomials of degree < k at the points (+ infinity row: leading coefficient)
    def V(k):
        M = np.zeros((n, k))
        for i in range(n - 1):
            M[i] = a[i] ** np.arange(k)
        M[n - 1, k - 1] = 1.0
        return M
    # Toom-Cook: polynomial product of degree n - 1 from n evaluations; correlation form by transposition
    Vn = V(n)
    AT = V(m).T                                   # m x n
    G = V(r)                                      # n x r
    BT = np.linalg.inv(Vn).T                      # n x n  (the interpolation, transposed)
    # scale G rows / BT rows so that BT is the "nice" side (any diagonal D: G <- D G, BT <- D^-1 BT keeps the identity)
    return AT, G, BT


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def split16(x):
    hi = bf16(x)
    return hi + bf16(x - hi)


def wino_corr(d, g, AT, G, BT, round_d=None, round_g=None, scale_d=False):
    """Full 2-D correlation out[b, co, y, x] = sum_{ci, u, v} d[b, ci, y + u, x + v] g[co, ci, u, v] with F(2x2, 5x5) tiles.
    d: (B, Ci, H, W) with H, W = 2 * T + 4; g: (Co, Ci, 5, 5).  fp32 arithmetic; round_* applied to the TRANSFORMED operands."""
    B, Ci, H, W = d.shape
    Co = g.shape[0]
    Ty, Tx = (H - 4) // 2, (W - 4) // 2
    ATt, Gt, BTt = (torch.tensor(M, dtype=torch.float32) for M in (AT, G, BT))
    U = torch.einsum("ik,ockl,jl->ocij", Gt, g, Gt)                                  # (Co, Ci, 6, 6)
    tiles = d.unfold(2, 6, 2).unfold(3, 6, 2)                                         # (B, Ci, Ty, Tx, 6, 6)
    Vt = torch.einsum("ik,bctxkl,jl->bctxij", BTt, tiles, BTt)                        # transformed input tiles
    if scale_d:                                                                       # per (i, j) position: a power-of-two gain
        s = Vt.abs().amax(dim=(0, 1, 2, 3), keepdim=True).clamp_min(1e-30)
        s = torch.exp2(torch.floor(torch.log2(s)))
        Vt, U = Vt / s, U * s.reshape(1, 1, 6, 6)
    if round_d is not None:
        Vt = round_d(Vt)
    if round_g is not None:
        U = round_g(U)
    M = torch.einsum("ocij,bctxij->botxij", U, Vt)                                    # 36 products per tile and channel pair
    Y = torch.einsum("mi,botxij,nj->botxmn", ATt, M, ATt)                             # (B, Co, Ty, Tx, 2, 2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, Co, 2 * Ty, 2 * Tx)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


if __name__ == "__main__":
    pts = [0.0, 1.0, -1.0, 2.0, -2.0]
    for arg in sys.argv[1:]:
        if arg.startswith("--points="):
            pts = [float(x) for x in arg.split("=")[1].split(",")]
    AT, G, BT = cook_toom(2, 5, pts)
    # self-check of the matrices in fp64 on a 1-D example
    dd, gg = np.random.rand(6), np.random.rand(5)
    want = np.array([np.dot(dd[i:i + 5], gg) for i in range(2)])
    got = AT @ ((G @ gg) * (BT @ dd))
    assert np.allclose(want, got, rtol=1e-9), (want, got)
    B, C, H = 2, 100, 44                                       # dy of a 100 -> 100 layer: (B, 100, 40, 40) padded by 4 -> 48; here 44 -> 40
    dy = torch.randn(B, C, H - 4, H - 4) * torch.rand(B, C, 1, 1)
    wt = (torch.rand(C, C, 5, 5) * 2 - 1) * (6.0 / (2 * C * 25)) ** 0.5 * 1.7
    gflip = wt.flip(2, 3).transpose(0, 1).contiguous()        # the data gradient's filter: (Ci_layer, Co_layer, 5, 5) flipped
    dpad = torch.nn.functional.pad(dy, (4, 4, 4, 4))
    exact = torch.nn.functional.conv2d(dpad.double(), gflip.double())
    rows = []
    direct2 = torch.nn.functional.conv2d(bf16(dpad), split16(gflip))
    rows.append(("direct-2   dy_hi x (W_hi + W_lo): the shipped rung, 2 MFMAs per product", rel(direct2, exact)))
    direct3 = torch.nn.functional.conv2d(split16(dpad), split16(gflip))
    rows.append(("direct-3   16-bit operands, 3 MFMAs per product", rel(direct3, exact)))
    rows.append(("wino-fp32  F(2x2,5x5), fp32 transforms, no operand rounding (0.36 products per product)", rel(wino_corr(dpad, gflip, AT, G, BT), exact)))
    rows.append(("wino-2     transformed dy -> bf16, transformed W -> 16 bits (0.72 MFMAs per product)", rel(wino_corr(dpad, gflip, AT, G, BT, bf16, split16), exact)))
    rows.append(("wino-2s    the same with a power-of-two gain per tile position", rel(wino_corr(dpad, gflip, AT, G, BT, bf16, split16, scale_d=True), exact)))
    rows.append(("wino-3     transformed dy -> 16 bits too (1.08 MFMAs per product: no saving)", rel(wino_corr(dpad, gflip, AT, G, BT, split16, split16), exact)))
    print("# Winograd F(2x2, 5x5) for the KPCN data gradient, emulated on the CPU (scripts/winograd_dgrad_emulation.py); points %s + infinity" % pts)
    print("# one 100 -> 100 5x5 layer, dy (2, 100, 40, 40) with per-channel scales, xavier weights; relative L2 error of dx against fp64")
    print("# transform gains: max |BT| = %.3g, max |G| = %.3g, max |AT| = %.3g" % (np.abs(BT).max(), np.abs(G).max(), np.abs(AT).max()))
    for name, e in rows:
        print("%-100s %.3e" % (name, e))
    for alt in ([0.0, 1.0, -1.0, 0.5, -0.5], [0.0, 1.0, -1.0, 2.0, -0.5]):
        A2, G2, B2 = cook_toom(2, 5, alt)
        print("%-100s %.3e" % ("wino-2     with points %s" % alt, rel(wino_corr(dpad, gflip, A2, G2, B2, bf16, split16), exact)))
