"""Parity of every HIP op (called through the C ABI) against the CPU oracle.

Tolerance: the north star asks for 1e-3 relative in fp32; these tests hold each op to 2e-5 of the
tensor's scale (max |a-b| / max |b|) against an fp64 reference unless stated (measured: 1e-7..3e-6,
scripts/diag_numerics.py), on seeded inputs.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import DEBUG_LIB, needs_debug_lib      # noqa: E402
from oracle import losses as ol            # noqa: E402
from oracle import modules as om           # noqa: E402
from oracle.step import assemble_input     # noqa: E402

DEV = "cuda"


def ops():
    from wcmc_amd import ops as _ops
    return _ops


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def assert_close(a, b, tol=2e-5, what=""):
    assert tuple(a.shape) == tuple(b.shape), (what, a.shape, b.shape)
    e = rel_err(a, b)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def test_library_loaded_is_in_tree():
    from wcmc_amd._lib import LIB_PATH, lib
    assert lib().wcmc_abi_version() == 2
    assert os.path.isfile(LIB_PATH)
    with open("/proc/self/maps") as f:
        assert "libwcmc_hip.so" in f.read()


@pytest.mark.parametrize("shape", [(2, 5, 7, 9), (1, 36, 16, 70), (3, 100, 5, 4)])
def test_layout_roundtrip(shape):
    o = ops()
    x = gen(*shape, seed=1).to(DEV)
    y = o.to_nhwc_raw(x)
    assert o.is_nhwc_view(y) and torch.equal(y.cpu(), x.cpu())
    assert torch.equal(o.from_nhwc_raw(y).cpu(), x.cpu())
    xs = x[:, :, 1:, 2:]                                    # strided source
    assert torch.equal(o.to_nhwc_raw(xs).cpu(), xs.cpu())


@pytest.mark.parametrize("shape", [(3, 36, 5, 70), (2, 64, 4, 64), (1, 7, 3, 9)])
def test_split_from_channel_first_equals_convert_then_split(shape):
    """wcmc_split_from_nchw (transpose + split in one pass, what PathNet does with `paths`) == to_nhwc + split, bit for
    bit, also from a strided source; and the attached split is what conv_chain_spp_mean consumes."""
    o = ops()
    x = gen(*shape, seed=65).to(DEV)
    assert torch.equal(o.split_from_nchw_raw(x), o.split_raw(o.to_nhwc_raw(x)))
    xs = gen(shape[0], shape[1] + 3, shape[2] + 2, shape[3] + 5, seed=66).to(DEV)[:, 2:-1, 1:-1, 3:-2]
    assert torch.equal(o.split_from_nchw_raw(xs), o.split_raw(o.to_nhwc_raw(xs)))


CONV_CASES = [
    # N, Cin, H, W, Cout, ks, pad, act
    (2, 39, 20, 20, 100, 5, 0, "relu"),
    (1, 100, 14, 17, 100, 5, 0, "relu"),
    (1, 100, 12, 12, 441, 5, 0, "linear"),
    (2, 34, 11, 13, 100, 5, 0, "relu"),
    (2, 64, 16, 16, 128, 3, 1, "relu"),
    (1, 384, 8, 8, 128, 3, 1, "relu"),
    (2, 192, 8, 12, 64, 3, 1, "leaky_relu"),
    (3, 36, 9, 10, 64, 1, 0, "relu"),
    (2, 128, 6, 6, 3, 1, 0, "relu"),
    (1, 7, 6, 5, 5, 3, 1, "linear"),
    (1, 256, 8, 8, 256, 3, 1, "relu"),
    # several 16x16 halo tiles per image with ragged edges; N*Ho >= 64 selects the filter-row weight-gradient kernel
    (2, 100, 40, 37, 100, 5, 0, "relu"),       # Wo = 33: one 64-pixel chunk per row, second k-step nearly empty
    (1, 100, 70, 75, 441, 5, 0, "linear"),     # Wo = 71: two chunks per row, four cout blocks
    (2, 100, 36, 36, 100, 5, 4, "linear"),     # full correlation (the data-gradient geometry), padded halo
    (1, 128, 37, 21, 128, 3, 1, "relu"),       # 3x3, 128-channel slabs (PXS 288), weights-in-registers candidates
    # filter-row weight-gradient instances (KS, cout tiles, cin tiles per block)
    (2, 39, 40, 37, 100, 5, 0, "relu"),        # (5,7,3)
    (2, 64, 40, 37, 64, 3, 1, "relu"),         # 3x3 below 256 input channels: one-tap kernel
    (1, 256, 70, 35, 128, 3, 1, "leaky_relu"), # (3,8,8), two cin blocks
    (1, 384, 64, 18, 128, 3, 1, "relu"),       # (3,8,8), three cin blocks
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_chain_single_layer_fwd_bwd(case, precision):
    from conftest import gtol, otol, ptol
    tol = ptol(precision, 2e-5, 1e-4)        # split-bf16 operands: ~2^-17 per operand
    gt = gtol(precision, 2e-5, 1e-4)         # gradients: the default mode rounds dy (and x in the weight gradient) to bf16
    n, cin, h, w, cout, ks, pad, act = case
    o = ops()
    x = gen(n, cin, h, w, seed=2)
    wt = gen(cout, cin, ks, ks, seed=3, scale=(2.0 / (cin * ks * ks)) ** 0.5 * 1.7)
    b = gen(cout, seed=4, scale=0.2)
    # oracle in fp64
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, wt, b))
    pre = F.conv2d(xr, wr, br, padding=pad)
    yr = om._activation(pre, act)
    # no upstream gradient where the pre-activation is within rounding of the activation's kink: there the two
    # implementations may legitimately pick different slopes (a whole dy * w row of difference per unit)
    gy = gen(*yr.shape, seed=5) * (pre.detach().abs() > 1e-4).float()
    yr.backward(gy.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, wt, b))
    y = o.conv_chain(xd, ks, pad, [act], [wd, bd])
    assert o.is_nhwc_view(y)
    y.backward(gy.to(DEV))
    assert_close(y, yr, tol=otol(precision, ks, act, cout, tol), what="conv fwd")
    assert_close(xd.grad, xr.grad, tol=gt, what="conv dgrad")
    assert_close(wd.grad, wr.grad, tol=gt, what="conv wgrad")
    assert_close(bd.grad, br.grad, tol=tol, what="conv bias grad")


def test_conv_chain_deep_matches_oracle_chain(precision):
    from conftest import gtol, ptol
    tol = ptol(precision, 2e-5, 2e-4)
    gt = gtol(precision, 2e-5, 2e-4)
    torch.manual_seed(11)
    ref = om.ConvChain(13, 25, ksize=5, width=20, depth=4, pad=False, output_type="linear", weight_norm=False).double()
    from wcmc_amd.modules import ConvChain
    mod = ConvChain(13, 25, ksize=5, width=20, depth=4, pad=False, output_type="linear", weight_norm=False)
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod.to(DEV)
    x = gen(2, 13, 30, 28, seed=12)
    from conftest import FlipCounter
    xr = x.double().requires_grad_(True)
    xd = x.to(DEV).requires_grad_(True)
    with FlipCounter() as fc:
        yr = ref(xr)
        y = mod(xd)
    g = gen(*yr.shape, seed=13)
    yr.backward(g.double())
    y.backward(g.to(DEV))
    assert_close(y, yr, tol=tol, what="chain fwd")
    fc.check(xd.grad, xr.grad, gt, what="chain dx", l2=2e-2)                 # 20-channel test chain: ~10k units / layer
    for (k, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        fc.check(p.grad, q.grad, gt, what="chain grad " + k, l2=2e-2)


def test_conv_wgrad_is_bitwise_reproducible():
    o = ops()
    x = o.to_nhwc_raw(gen(2, 100, 24, 24, seed=20).to(DEV))
    dy = o.to_nhwc_raw(gen(2, 100, 20, 20, seed=21).to(DEV))
    a = o.conv2d_wgrad_raw(x, dy, 5, 0, (100, 100, 5, 5))
    b = o.conv2d_wgrad_raw(x, dy, 5, 0, (100, 100, 5, 5))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    xs, dys = o.split_raw(x), o.split_raw(dy)
    a = o.conv2d_wgrad_x_raw(xs, (2, 100, 24, 24), dys, 100, 5, 0, (100, 100, 5, 5))
    b = o.conv2d_wgrad_x_raw(xs, (2, 100, 24, 24), dys, 100, 5, 0, (100, 100, 5, 5))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # filter-row kernel (N*Ho >= 64): fixed slab order as well
    xs = o.split_raw(o.to_nhwc_raw(gen(4, 100, 24, 24, seed=22).to(DEV)))
    dys = o.split_raw(o.to_nhwc_raw(gen(4, 100, 20, 20, seed=23).to(DEV)))
    a = o.conv2d_wgrad_x_raw(xs, (4, 100, 24, 24), dys, 100, 5, 0, (100, 100, 5, 5))
    b = o.conv2d_wgrad_x_raw(xs, (4, 100, 24, 24), dys, 100, 5, 0, (100, 100, 5, 5))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("geom", [(8, 44, 44, 0), (4, 37, 70, 0), (3, 30, 101, 2), (16, 8, 8, 0), (8, 124, 124, 0),
                                  (2, 40, 40, 0, 100, 441), (2, 36, 52, 1, 224, 112), (2, 24, 24, 0, 210, 212)])
@needs_debug_lib
def test_eight_wave_filter_row_kernel_equals_seven_wave(geom, monkeypatch):
    """conv_wgrad_rows8_bf16x3_kernel (245 accumulator tiles dealt over eight waves, priority hand-over inside a stage)
    writes the slabs of conv_wgrad_rows_bf16x3_kernel<5, 7, 7> bit for bit: ragged chunks (Wo % 64 in {40, 2, 37, 4}),
    short last splits, padding, and every hand-over point."""
    o = ops()
    n, h, w, pad = geom[:4]
    cin, cout = geom[4:] if len(geom) > 4 else (100, 100)     # (several cout / cin blocks of 7 tiles: the 441-cout layer, 14 x 14 tiles)
    ho, wo = h + 2 * pad - 4, w + 2 * pad - 4
    xs = o.split_raw(o.to_nhwc_raw(gen(n, cin, h, w, seed=90).to(DEV)))
    dys = o.split_raw(o.to_nhwc_raw(gen(n, cout, ho, wo, seed=91).to(DEV)))
    monkeypatch.setenv("WCMC_WGRAD_ROWS8", "0")
    want = o.conv2d_wgrad_x_raw(xs, (n, cin, h, w), dys, cout, 5, pad, (cout, cin, 5, 5), terms=3)
    want1 = o.conv2d_wgrad_x_raw(xs, (n, cin, h, w), dys, cout, 5, pad, (cout, cin, 5, 5), terms=1)    # the one-plane instances alike
    for prio, xe in (("0", "1"), ("1", "1"), ("7", "0"), ("9", "1"), ("13", "0"), ("8", "1"), ("8", "0")):
        monkeypatch.setenv("WCMC_WGRAD_ROWS8", "1")
        monkeypatch.setenv("WCMC_WGRAD_ROWS8_PRIO", prio)
        monkeypatch.setenv("WCMC_WGRAD_ROWS8_XE", xe)           # both dealings of the 21 left-over tiles
        got = o.conv2d_wgrad_x_raw(xs, (n, cin, h, w), dys, cout, 5, pad, (cout, cin, 5, 5), terms=3)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (prio, xe)
        if xe == "1":
            got = o.conv2d_wgrad_x_raw(xs, (n, cin, h, w), dys, cout, 5, pad, (cout, cin, 5, 5), terms=1)
            assert torch.equal(got[0], want1[0]) and torch.equal(got[1], want1[1]), (prio, "one plane")
    assert want[0].abs().max().item() > 0
    if len(geom) > 4:                                   # the new block shapes against fp64 as well
        ref = torch.nn.grad.conv2d_weight(gen(n, cin, h, w, seed=90).double(), (cout, cin, 5, 5), gen(n, cout, ho, wo, seed=91).double(),
                                          padding=pad)
        assert_close(want[0], ref, tol=2e-5, what="dw of %d -> %d" % (cin, cout))


@pytest.mark.parametrize("case", [(2, 100, 40, 37, 100, 5, 0), (3, 36, 9, 10, 64, 1, 0), (2, 64, 16, 16, 128, 3, 1)])
def test_gate_from_bit_mask_equals_gate_from_tensor(case):
    """The data-gradient GEMM gated by the (hi > 0) bit mask of the forward launch == gated by the activation
    tensor itself, bit for bit (halo and streaming kernels, relu and leaky_relu)."""
    o = ops()
    n, cin, h, w, cout, ks, pad = case
    ho, wo = h + 2 * pad - ks + 1, w + 2 * pad - ks + 1
    xs = o.split_raw(o.to_nhwc_raw(gen(n, cin, h, w, seed=70).to(DEV)))
    wt = gen(cout, cin, ks, ks, seed=71, scale=0.1).to(DEV)
    b = gen(cout, seed=72, scale=0.3).to(DEV)
    y, mask = o.conv2d_x_raw(xs, (n, cin, h, w), o._pack_x(wt, 0), b, cout, ks, pad, "relu", out_split=True, mask_out=True)
    yd = o.unsplit_debug(y, n, cout, ho, wo)
    bits = torch.from_numpy(np.unpackbits(mask.cpu().numpy().reshape(n * ho * wo, -1), axis=1, bitorder="little"))
    want = (o.unsplit_debug(y, n, cout, ho, wo).permute(0, 2, 3, 1).reshape(n * ho * wo, cout) > 0).cpu()
    # (a value whose hi plane rounds to zero is positive only through its lo plane: not produced by these inputs)
    assert torch.equal(bits[:, :cout].bool(), want) and float((yd > 0).float().mean()) > 0.2
    dys = o.split_raw(o.to_nhwc_raw(gen(n, 48, ho, wo, seed=73).to(DEV)))
    w2t = o._pack_x(gen(48, cout, ks, ks, seed=74, scale=0.1).to(DEV), 1)
    for act in ("relu", "leaky_relu"):
        a = o.conv2d_x_raw(dys, (n, 48, ho, wo), w2t, None, cout, ks, ks - 1 - pad, "linear", out_split=True,
                           gate=y, gate_act=act) if pad == ks // 2 or ks == 1 else None
        if a is None:      # valid (unpadded) forward: the data gradient of the NEXT layer has y's geometry only when
            continue       # that layer is "same"-padded or 1x1; the 5x5 valid case is covered through the chains
        bb = o.conv2d_x_raw(dys, (n, 48, ho, wo), w2t, None, cout, ks, ks - 1 - pad, "linear", out_split=True,
                            gate_mask=mask, gate_act=act)
        assert torch.equal(a, bb), act


HALO5_CASES = [
    # N, Cin, H, W, Cout, pad, split output -- ragged 16x16 / 12x16 tiles, the three slab structures (104 = 6 x 16 + 8,
    # 40 = 16 + 16 + 8, 448 = 28 x 16), four cout blocks, the fp32-view output of the last KPCN layer
    (2, 100, 40, 37, 100, 0, True),
    (1, 100, 70, 75, 441, 0, False),
    (2, 39, 29, 52, 100, 0, True),
    (1, 441, 21, 26, 100, 4, True),
    (3, 100, 20, 33, 39, 4, True),
    # output heights that neither 16 nor 12 divides: tile rows of 16 followed by tile rows of 12 in ONE launch (launch_xhalo64, p.rows16):
    # 28 = 16 + 12, 44 = 2 x 16 + 12, 100 = 4 x 16 + 3 x 12 (KPCN), 124 = 7 x 16 + 12 (KPCN's first layer), 28 with a padded halo,
    # 108 = 6 x 16 + 12 (12 divides it: mixed all the same)
    (1, 100, 32, 37, 100, 0, True),
    (2, 100, 48, 20, 100, 0, True),
    (1, 100, 104, 33, 100, 0, False),
    (1, 39, 128, 20, 100, 0, True),
    (1, 100, 24, 21, 100, 4, True),
    (1, 100, 112, 20, 100, 0, True),
    # output widths that 16 does not divide: tile columns of 16 + a strip of TRANSPOSED 12-wide workgroups in the same launch (p.stripX):
    # 28 x 28 = (16 + 12)^2, 32 x 44, 100 x 100 (KPCN: 4 + 3 columns, 4 + 3 tile rows), 28 x 40 with a padded halo (one column of 16, two of 12)
    (1, 100, 32, 32, 100, 0, True),
    (2, 100, 36, 48, 100, 0, True),
    (1, 100, 104, 104, 100, 0, False),
    (1, 100, 24, 36, 100, 4, True),
    # ... through the narrower cout blocks (NT = 4: 39 couts; NT = 1: 16 couts) and three images
    (3, 100, 32, 32, 39, 0, True),
    (2, 40, 32, 48, 16, 0, True),
]


@pytest.mark.parametrize("case", HALO5_CASES)
def test_conv5x5_kernel_variants_agree_with_fp64(case, monkeypatch):
    """The KPCN 5x5 forward / data-gradient launches through every variant of their kernel -- 12x16 tiles (what small
    launches get), 16x16 tiles only, without the priority alternation, the 96-byte halo stride is a process-wide plan
    (not switched here), and the 8x16 kernel they replaced -- against an fp64 convolution, gated by a bit mask and with
    the column sums (the consumer's bias gradient) where the output is split."""
    o = ops()
    n, cin, h, w, cout, pad, split = case
    ks = 5
    ho, wo = h + 2 * pad - ks + 1, w + 2 * pad - ks + 1
    x = gen(n, cin, h, w, seed=80)
    wt = gen(cout, cin, ks, ks, seed=81, scale=(2.0 / (cin * ks * ks)) ** 0.5 * 1.7)
    b = gen(cout, seed=82, scale=0.2)
    keep = (gen(n, cout, ho, wo, seed=83) > -0.3)                     # the gate: ~65 % of the outputs pass
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=pad)) * keep.double()
    xs = o.split_raw(o.to_nhwc_raw(x.to(DEV)))
    cp = (cout + 7) // 8 * 8
    bits = torch.zeros(n, ho, wo, cp, dtype=torch.bool)
    bits[..., :cout] = keep.permute(0, 2, 3, 1)
    mask = torch.from_numpy(np.packbits(bits.numpy().reshape(-1, cp), axis=1, bitorder="little").reshape(-1)).to(DEV)
    name = "conv_halo64 (the tile height the plan picks)"
    wp = o._pack_x(wt.to(DEV), 0)                                  # (the packing follows the kernel's slab plan)
    if split:
        y, part = o.conv2d_x_raw(xs, (n, cin, h, w), wp, b.to(DEV), cout, ks, pad, "relu", out_split=True,
                                 gate_mask=mask, gate_act="relu", colsum=True)
        yd = o.unsplit_debug(y, n, cout, ho, wo)
        db = o.colsum_finish_raw(part, (n, cout, ho, wo))
        assert_close(db, ref.sum(dim=(0, 2, 3)), tol=2e-5, what=name + ": column sums")
    else:
        yd = o.conv2d_x_raw(xs, (n, cin, h, w), wp, b.to(DEV), cout, ks, pad, "relu", out_split=False)
        yd = yd * keep.to(DEV)
    assert_close(yd, ref, tol=2e-5, what=name)


@needs_debug_lib
@pytest.mark.parametrize("case", [(1, 100, 32, 37, 100, 0), (2, 100, 48, 20, 100, 0), (1, 100, 104, 33, 100, 0), (1, 39, 128, 20, 100, 0), (1, 100, 24, 21, 100, 4),
                                  (8, 100, 120, 120, 100, 0), (2, 100, 96, 96, 441, 0), (1, 100, 112, 20, 100, 0), (1, 100, 124, 36, 100, 0),
                                  (1, 100, 32, 32, 100, 0), (2, 100, 36, 48, 100, 0), (1, 100, 104, 104, 100, 0), (1, 39, 128, 128, 100, 0), (1, 100, 24, 36, 100, 4),
                                  (2, 100, 112, 112, 100, 0), (3, 100, 32, 32, 39, 0), (2, 40, 32, 48, 16, 0), (1, 100, 44, 60, 441, 0)])
def test_halo64_mixed_tile_heights_variant_is_bit_identical_to_the_pure_tilings(case, monkeypatch):
    """Tile rows of 16 and of 12 pixels in one launch (p.rows16), and tile columns of 16 with a strip of transposed 12-wide workgroups
    (p.stripX), against the pure 12x16 / 16x16 tilings (WCMC_HALO64_MIX=0, WCMC_HALO64_STRIP=0, debug build): a pixel's products are summed in the same order whatever tile it sits in, so outputs, gate masks and results of the
    data-gradient orientation are equal BIT FOR BIT; the per-tile column sums are different partitions of the same rows."""
    o = ops()
    n, cin, h, w, cout, pad = case
    ks = 5
    ho, wo = h + 2 * pad - ks + 1, w + 2 * pad - ks + 1
    xs = o.split_raw(o.to_nhwc_raw(gen(n, cin, h, w, seed=420).to(DEV)))
    wt = gen(cout, cin, ks, ks, seed=421, scale=(2.0 / (cin * ks * ks)) ** 0.5 * 1.7).to(DEV)
    b = gen(cout, seed=422, scale=0.2).to(DEV)
    got = {}
    for sw in ("2", "0"):
        monkeypatch.setenv("WCMC_HALO64_MIX", sw)
        monkeypatch.setenv("WCMC_HALO64_STRIP", "1" if sw == "2" else "0")      # (the transposed 12-wide strip of the widths likewise)
        y, part, mask = o.conv2d_x_raw(xs, (n, cin, h, w), o._pack_x(wt, 0), b, cout, ks, pad, "relu", out_split=True, colsum=True, mask_out=True)
        yf = o.conv2d_x_raw(xs, (n, cin, h, w), o._pack_x(wt, 3), b, cout, ks, pad, "linear", out_split=False, terms=2)
        got[sw] = (y.clone(), mask.clone(), yf.clone(), o.colsum_finish_raw(part, (n, cout, ho, wo)))
    a, bb = got["2"], got["0"]
    assert torch.equal(a[0], bb[0]) and torch.equal(a[1], bb[1]) and torch.equal(a[2], bb[2])
    assert rel_err(a[3], bb[3]) < 2e-6


@needs_debug_lib
@pytest.mark.parametrize("case", [(2, 100, 40, 37, 100, 0), (1, 100, 30, 30, 441, 0), (2, 36, 20, 33, 100, 0), (2, 100, 36, 36, 100, 4), (1, 97, 21, 26, 100, 4),
                                  (8, 100, 116, 116, 100, 0)])
def test_halo64_last_slab_variant_four_channel_k_order_equals_eight(case, monkeypatch):
    """conv_halo64_bf16x3_kernel, input channels = slabs + at most four (KPCN's 100): the last slab in the four-channel K order (eight
    taps per stage, x_last4) against the eight-channel order it replaced (WCMC_HALO64_L4=0, debug build) -- three-term forward, two-term
    data-gradient orientation, the one-MFMA and the fp16 output-layer forms.  Same products, another order of summation."""
    from wcmc_amd._lib import lib
    o = ops()
    n, cin, h, w, cout, pad = case
    ks = 5
    x = gen(n, cin, h, w, seed=410)
    wt = gen(cout, cin, ks, ks, seed=411, scale=(2.0 / (cin * ks * ks)) ** 0.5 * 1.7).to(DEV)
    b = gen(cout, seed=412, scale=0.2).to(DEV)
    xs = o.split_raw(o.to_nhwc_raw(x.to(DEV)))
    got = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("WCMC_HALO64_L4", sw)
        dims = (n, cin, h, w)
        y3 = o.conv2d_x_raw(xs, dims, o._pack_x(wt, 0), b, cout, ks, pad, "relu", out_split=False)
        y2 = o.conv2d_x_raw(xs, dims, o._pack_x(wt, 3), b, cout, ks, pad, "linear", out_split=False, terms=2)
        y1 = o.conv2d_x_raw(xs, dims, o._pack_x(wt, 3), b, cout, ks, pad, "linear", out_split=False, terms=1)
        yh = o.conv2d_out_f16_raw(xs, dims, o._pack_x(wt, 4), b, cout, ks, pad) if lib().wcmc_conv2d_out_f16_supported(cin, cout, ks) else y1
        got[sw] = [t.clone() for t in (y3, y2, y1, yh)]
    for a, bb, what in zip(got["1"], got["0"], ("three terms", "two terms", "one term", "fp16")):
        assert rel_err(a, bb) < 2e-6, what
    ref = F.relu(F.conv2d(x.double(), wt.double().cpu(), b.double().cpu(), padding=pad))
    assert_close(got["1"][0], ref, tol=2e-5, what="three-term forward against fp64")


@needs_debug_lib
@pytest.mark.parametrize("case", [(2, 64, 40, 37, 64), (1, 128, 37, 21, 128), (3, 192, 19, 33, 64), (1, 384, 16, 18, 128), (2, 256, 9, 16, 256),
                                  (1, 64, 8, 16, 64), (1, 64, 1, 1, 64), (2, 64, 33, 47, 128), (8, 64, 128, 128, 64)])
def test_unet3x3_kernel_variant_equals_the_kernel_it_replaced(case, monkeypatch):
    """conv_halo3_bf16x3_kernel (round 6: K split over two wave groups, swizzled halo) against conv_halo_bf16x3_kernel (WCMC_HALO3=0,
    debug build) on the U-Net's launches: forward with bias + ReLU + gate mask out, two-term data gradient gated by that mask with
    column sums, and the fp32-view output.  Same products, another summation order (the two K groups meet once): 2e-6 of the
    tensor's scale in fp32, one unit of the lo plane in split outputs; masks equal wherever the value is not within rounding of zero."""
    o = ops()
    n, cin, h, w, cout = case
    dims = (n, cin, h, w)
    xs = o.split_raw(o.to_nhwc_raw(gen(n, cin, h, w, seed=400).to(DEV)))
    wt = gen(cout, cin, 3, 3, seed=401, scale=(2.0 / (cin * 9)) ** 0.5 * 1.7).to(DEV)
    b = gen(cout, seed=402, scale=0.2).to(DEV)
    dys = o.split_raw(o.to_nhwc_raw(gen(n, cout, h, w, seed=403).to(DEV)))
    wp0, wp2 = o._pack_x(wt, 0), o._pack_x(wt, 2)
    cp = (cin + 7) // 8 * 8
    keep = torch.zeros(n, h, w, cp, dtype=torch.bool)
    keep[..., :cin] = (gen(n, cin, h, w, seed=404) > -0.3).permute(0, 2, 3, 1)
    gmask = torch.from_numpy(np.packbits(keep.numpy().reshape(-1, cp), axis=1, bitorder="little").reshape(-1)).to(DEV)
    got = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("WCMC_HALO3", sw)
        y, mask = o.conv2d_x_raw(xs, dims, wp0, b, cout, 3, 1, "relu", out_split=True, mask_out=True)
        yf = o.conv2d_x_raw(xs, dims, wp0, b, cout, 3, 1, "leaky_relu", out_split=False)
        dx, part = o.conv2d_x_raw(dys, (n, cout, h, w), wp2, None, cin, 3, 1, "linear", out_split=True, gate_mask=gmask, gate_act="relu",
                                    colsum=True, terms=2)
        got[sw] = (o.unsplit_debug(y, n, cout, h, w), mask.clone(), yf.clone(),
                   o.unsplit_debug(dx, n, cin, h, w), o.colsum_finish_raw(part, (n, cin, h, w)))
    a, bb = got["1"], got["0"]
    # (a split output resolves 2^-17 of an element: two fp32 sums an ulp apart may round its lo plane apart)
    for i, what, tol in ((0, "forward", 1.6e-5), (2, "forward, fp32 view", 2e-6), (3, "data gradient", 1.6e-5), (4, "column sums (of the split values, with cancellation)", 1e-5)):
        assert rel_err(a[i], bb[i]) < tol, what
    bits = lambda m: np.unpackbits(m.cpu().numpy(), bitorder="little")
    differ = int((bits(a[1]) != bits(bb[1])).sum())
    near_zero = int((a[0].abs() < 1e-5 * float(a[0].abs().max())).sum() - (a[0] == 0).sum())
    assert differ <= max(near_zero, 0), (differ, near_zero)


@pytest.mark.parametrize("case", [(16, 36, 64, 64, 64, 1), (12, 128, 48, 40, 128, 1), (16, 128, 64, 64, 3, 1), (4, 128, 40, 37, 128, 3)])
def test_weight_gradient_with_many_slabs_against_fp64(case, monkeypatch):
    """Weight and bias gradients at sizes where the split-K plan has MANY slabs (the unit cases above have one or a few):
    the 1x1 layers' slab reduction in groups, the filter-row kernel's KS = 1 instance (128 -> 128) and its 128-channel 3x3
    one, against an fp64 reduction; and twice, bit for bit (fixed order of additions)."""
    o = ops()
    n, cin, h, w, cout, ks = case
    pad = ks // 2
    x = gen(n, cin, h, w, seed=90)
    dy = gen(n, cout, h, w, seed=91)
    xs = o.split_raw(o.to_nhwc_raw(x.to(DEV)))
    dys = o.split_raw(o.to_nhwc_raw(dy.to(DEV)))
    want = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, ks, ks), dy.double(), padding=pad)
    got = {}
    for name, env in (("shipped plan", {}),) + ((("one-tap kernel", {"WCMC_WGRAD_ROWS": "0"}),) if DEBUG_LIB else ()):
        for k in ("WCMC_WGRAD_ROWS",):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dw, db = o.conv2d_wgrad_x_raw(xs, (n, cin, h, w), dys, cout, ks, pad, (cout, cin, ks, ks), terms=3)
        dw2, db2 = o.conv2d_wgrad_x_raw(xs, (n, cin, h, w), dys, cout, ks, pad, (cout, cin, ks, ks), terms=3)
        assert torch.equal(dw, dw2) and torch.equal(db, db2), name
        assert_close(dw, want, tol=2e-5, what=name + ": dw")
        assert_close(db, dy.double().sum(dim=(0, 2, 3)), tol=2e-5, what=name + ": db")
        got[name] = dw
    assert not DEBUG_LIB or rel_err(got["shipped plan"], got["one-tap kernel"]) < 1e-5


def _bf16_round(t):
    return t.bfloat16().float()


WGRAD_TERM_CASES = [
    # N, Cin, H, W, Cout, ks, pad -- one per weight-gradient kernel instance the plan can pick
    (8, 100, 44, 44, 100, 5, 0),     # conv_wgrad_rows8 (eight waves), one chunk per row
    (2, 100, 30, 101, 100, 5, 2),    # conv_wgrad_rows8, two chunks per row (Wo = 101), padding
    (2, 100, 40, 40, 441, 5, 0),     # conv_wgrad_rows8, four cout blocks
    (2, 39, 40, 37, 100, 5, 0),      # conv_wgrad_rows<5,7,3>
    (1, 256, 70, 35, 128, 3, 1),     # conv_wgrad_rows<3,8,8>, two cin blocks
    (2, 64, 40, 37, 64, 3, 1),       # conv_wgrad_rows<3,4,4> / one-tap kernel (TM = 4)
    (12, 128, 48, 40, 128, 1, 0),    # conv_wgrad_rows<1,8,8>
    (16, 36, 64, 64, 64, 1, 0),      # one-tap kernel, TM = 4, many slabs
    (2, 34, 11, 13, 100, 5, 0),      # one-tap kernel, TM = 7 (22 vectors per pixel over 4 threads: ragged)
    (3, 128, 9, 10, 3, 1, 0),        # one-tap kernel, 3 couts
]


@pytest.mark.parametrize("case", WGRAD_TERM_CASES)
def test_one_term_weight_gradient_is_the_three_term_one_on_bf16_operands(case, monkeypatch):
    """terms = 1 of wcmc_conv2d_wgrad_bf16x3 multiplies the hi planes only.  On operands that ARE bf16 numbers the lo planes
    are zero, the two dropped MFMAs add exact zeros, and every remaining MFMA and slab addition is the same: the result must
    equal the three-term kernel's BIT FOR BIT -- for every kernel instance (eight-wave / generic filter-row / one-tap), and
    both must sit on the fp64 value.  On general operands the one-term result is the exact gradient of the rounded operands:
    checked against fp64 on the rounded values at the same tolerance."""
    o = ops()
    n, cin, h, w, cout, ks, pad = case
    ho, wo = h + 2 * pad - ks + 1, w + 2 * pad - ks + 1
    x, dy = gen(n, cin, h, w, seed=190), gen(n, cout, ho, wo, seed=191)
    xb, dyb = _bf16_round(x), _bf16_round(dy)
    xs, dys = o.split_raw(o.to_nhwc_raw(xb.to(DEV))), o.split_raw(o.to_nhwc_raw(dyb.to(DEV)))
    want = torch.nn.grad.conv2d_weight(xb.double(), (cout, cin, ks, ks), dyb.double(), padding=pad)
    for env in ({},) + (({"WCMC_WGRAD_ROWS": "0"},) if DEBUG_LIB else ()):
        monkeypatch.delenv("WCMC_WGRAD_ROWS", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dw3, db3 = o.conv2d_wgrad_x_raw(xs, (n, cin, h, w), dys, cout, ks, pad, (cout, cin, ks, ks), terms=3)
        dw1, db1 = o.conv2d_wgrad_x_raw(xs, (n, cin, h, w), dys, cout, ks, pad, (cout, cin, ks, ks), terms=1)
        assert torch.equal(dw1, dw3) and torch.equal(db1, db3), env
        assert_close(dw1, want, tol=2e-5, what="dw on bf16 operands")
    # general operands: one term == the gradient of the ROUNDED operands; the bias gradient still sums hi + lo of dy
    xs, dys = o.split_raw(o.to_nhwc_raw(x.to(DEV))), o.split_raw(o.to_nhwc_raw(dy.to(DEV)))
    dw1, db1 = o.conv2d_wgrad_x_raw(xs, (n, cin, h, w), dys, cout, ks, pad, (cout, cin, ks, ks), terms=1)
    assert_close(dw1, want, tol=2e-5, what="dw of general operands = dw of their hi planes")
    assert_close(db1, dy.double().sum(dim=(0, 2, 3)), tol=2e-5, what="db")
    full = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, ks, ks), dy.double(), padding=pad)
    assert rel_err(dw1, full) < 1.5e-2          # (2^-9 per operand, uncorrelated from pixel to pixel)


TWO_TERM_FALLBACKS = ((2, 100, 30, 29, 39, 4, 5), (2, 120, 24, 24, 100, 4, 5), (2, 64, 20, 20, 32, 1, 3))


@pytest.mark.parametrize("case", ((8, 100, 44, 44, 100, 0, 5), (2, 100, 36, 36, 100, 4, 5), (1, 441, 40, 37, 100, 4, 5), (2, 232, 24, 24, 100, 4, 5),
                                  # the U-Net's 3x3 layers (conv_halo3_bf16x3_kernel<1, 2 | 4, 2> where couts come in 64s and cins in whole 64- / 128-channel
                                  # slabs, conv_halo_bf16x3_kernel<4 | 7, .., AP = 1> elsewhere): one 64- / 128-channel slab, two of 96,
                                  # three of 128, 12 cout tiles (NT = 7), a 40-channel slab, ragged tiles, the benchmark's shape
                                  (2, 64, 40, 37, 64, 1, 3), (1, 128, 37, 21, 128, 1, 3), (2, 192, 24, 24, 64, 1, 3), (1, 384, 33, 18, 128, 1, 3),
                                  (1, 256, 20, 36, 192, 1, 3), (1, 40, 20, 20, 64, 1, 3), (8, 64, 128, 128, 64, 1, 3),
                                  # the ONE-cout-tile 5x5 instance (conv_halo64<1, 3, PT, 0, 80, 1>: x_plan_k grants it to every two-term 5x5
                                  # launch with at most 16 output rows, not only to the sliced first-layer gradient of KPCN): 3 / 8 / 16
                                  # rows, 32 / 40 / 100 channels (Kp % 32 = 0 and 8), odd sizes with partial tiles
                                  (2, 32, 21, 23, 3, 4, 5), (1, 40, 19, 27, 8, 4, 5), (2, 100, 25, 21, 16, 4, 5), (1, 100, 17, 33, 3, 4, 5),
                                  (3, 32, 13, 40, 16, 4, 5)) + TWO_TERM_FALLBACKS)
def test_two_term_data_gradient_against_fp64(case):
    """terms = 2 of wcmc_conv2d_igemm_bf16x3: x (= dy in the data gradient) rounded to its hi plane, W exact to 16 bits,
    weights packed with mode 2 (32-channel slabs in the hi-plane-only halo).  Against fp64 on the rounded x at the kernel's
    usual tolerance, gated and ungated, split and fp32 outputs; and against the three-term launch on bf16 operands.  The last
    three cases have no two-term instance (5x5 with 39 couts: NT = 4; 120 channels: a 24-channel last slab; 3x3 with two cout
    tiles): packing and launch must fall back to the three-term plan TOGETHER (the result is then the exact data gradient)."""
    o = ops()
    n, cin, h, w, cout, pad, ks = case
    ho, wo = h + 2 * pad - ks + 1, w + 2 * pad - ks + 1
    x = gen(n, cin, h, w, seed=290)
    wt = gen(cin, cout, ks, ks, seed=291, scale=(2.0 / (cin * ks * ks)) ** 0.5 * 1.7)  # layer weight (Cout_l = cin, Cin_l = cout)
    xb = _bf16_round(x)
    xs, xbs = o.split_raw(o.to_nhwc_raw(x.to(DEV))), o.split_raw(o.to_nhwc_raw(xb.to(DEV)))
    wd = wt.to(DEV)
    wp3, wp2 = o._pack_x(wd, 1), o._pack_x(wd, 2)
    fallback = case in TWO_TERM_FALLBACKS
    if fallback:          # (the converse does not hold: 448 channels are 14 slabs of 32 in either plan)
        assert wp3.numel() == wp2.numel() and torch.equal(wp3, wp2)
    # the data-gradient GEMM: conv of x with the flipped, channel-swapped filter
    wf = wt.flip(2, 3).transpose(0, 1).contiguous()
    want = F.conv2d((x if fallback else xb).double(), wf.double(), padding=pad)
    y2 = o.conv2d_x_raw(xs, (n, cin, h, w), wp2, None, cout, ks, pad, "linear", out_split=False, terms=2)
    assert_close(y2, want, tol=2e-5, what="two-term dgrad (general x = its hi plane)")
    y2b = o.conv2d_x_raw(xbs, (n, cin, h, w), wp2, None, cout, ks, pad, "linear", out_split=False, terms=2)
    assert torch.equal(y2, y2b) != fallback           # the lo plane of x is never read
    y3b = o.conv2d_x_raw(xbs, (n, cin, h, w), wp3, None, cout, ks, pad, "linear", out_split=False, terms=3)
    assert rel_err(y2b, y3b) < 2e-6                   # same products, another K order
    gate = o.split_raw(o.to_nhwc_raw(gen(n, cout, ho, wo, seed=292).to(DEV)))
    ys, part = o.conv2d_x_raw(xs, (n, cin, h, w), wp2, None, cout, ks, pad, "linear", out_split=True, gate=gate, gate_act="relu",
                              colsum=True, terms=2)
    g = (o.unsplit_debug(gate, n, cout, ho, wo) > 0).double().cpu()
    assert_close(o.unsplit_debug(ys, n, cout, ho, wo), want * g, tol=2e-5, what="two-term dgrad, gated split output")
    assert_close(o.colsum_finish_raw(part, (n, cout, ho, wo)), (want * g).sum(dim=(0, 2, 3)), tol=2e-5, what="its column sums")
    full = F.conv2d(x.double(), wf.double(), padding=pad)
    assert rel_err(y2, full) < 1e-2


ONE_TERM_FALLBACKS = ((2, 100, 20, 21, 39, 0, 5), (2, 120, 20, 20, 100, 0, 5), (2, 64, 20, 20, 128, 1, 3))


@pytest.mark.parametrize("case", ((8, 100, 96, 96, 441, 0, 5), (1, 100, 40, 37, 441, 0, 5), (2, 100, 36, 36, 100, 4, 5), (1, 104, 30, 30, 112, 0, 5),
                                  (2, 72, 24, 27, 100, 0, 5)) + ONE_TERM_FALLBACKS)
def test_one_term_output_layer_forward_is_exact_on_the_rounded_operands(case):
    """terms = 1 of wcmc_conv2d_igemm_bf16x3 (the forward of an un-gated 5x5 output layer in the default mode "bf16x321o":
    ``conv_halo64_bf16x3_kernel<7, 3, PT, 0, 80, 1, 1>``): the hi planes of x and W only, ONE bf16 MFMA per product, weights packed
    with mode 3 (forward orientation in the hi-plane K order).  The arithmetic is pinned independently of any tolerance choice:
      * against fp64 on the bf16-ROUNDED operands at the kernels' usual 2e-5 (so the result IS the exact convolution of x_hi
        with W_hi, bias included);
      * on operands that ARE bf16 numbers the dropped MFMAs add exact zeros: equal to the two-term launch BIT FOR BIT and to the
        three-term launch up to its K order;
      * the lo planes are never read.
    The last three cases have no one-term instance (39 couts: NT = 4; a 24-channel last slab; 3x3): packing and launch fall back
    to the plan's instance TOGETHER and the result is the exact convolution of the unrounded operands."""
    o = ops()
    n, cin, h, w, cout, pad, ks = case
    x = gen(n, cin, h, w, seed=490)
    wt = gen(cout, cin, ks, ks, seed=491, scale=(2.0 / (cin * ks * ks)) ** 0.5 * 1.7)
    b = gen(cout, seed=492, scale=0.2)
    xb, wb = _bf16_round(x), _bf16_round(wt)
    xs, xbs = o.split_raw(o.to_nhwc_raw(x.to(DEV))), o.split_raw(o.to_nhwc_raw(xb.to(DEV)))
    wd, wbd, bd = wt.to(DEV), wb.to(DEV), b.to(DEV)
    wp1, wp1b, wp3b = o._pack_x(wd, 3), o._pack_x(wbd, 3), o._pack_x(wbd, 0)
    fallback = case in ONE_TERM_FALLBACKS
    run = lambda xs_, wp, t: o.conv2d_x_raw(xs_, (n, cin, h, w), wp, bd, cout, ks, pad, "linear", out_split=False, terms=t)
    y1 = run(xs, wp1, 1)
    if fallback:
        assert torch.equal(o._pack_x(wd, 0), wp1) or ks == 3      # (3x3: the hi-plane plan exists, the one-plane weight path does not)
        want = F.conv2d(x.double(), wt.double(), b.double(), padding=pad) if ks != 3 else F.conv2d(xb.double(), wt.double(), b.double(), padding=pad)
        assert_close(y1, want, tol=2e-5, what="no one-term instance: the plan's own arithmetic")
        return
    want = F.conv2d(xb.double(), wb.double(), b.double(), padding=pad)
    assert_close(y1, want, tol=2e-5, what="one-term forward = conv(x_hi, W_hi) + b")
    y1b = run(xbs, wp1b, 1)
    assert torch.equal(y1, y1b)                                   # the lo planes of x and W are never read
    y2b = run(xbs, wp1b, 2)
    assert torch.equal(y1b, y2b)                                  # W_lo = 0: the dropped MFMA adds exact zeros
    y3b = run(xbs, wp3b, 3)
    assert rel_err(y1b, y3b) < 2e-6                               # same products, another K order
    full = F.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    assert 1e-4 < rel_err(y1, full) < 1e-2                        # (2^-9 per operand on i.i.d. test data; the rung's cost in the
                                                                  # network is measured end to end: profiles/r04_forward_ladder.txt)


@pytest.mark.parametrize("case", ((8, 100, 96, 96, 441, 0, 5), (1, 100, 40, 37, 441, 0, 5), (2, 100, 36, 36, 100, 4, 5), (1, 104, 30, 30, 112, 0, 5),
                                  (2, 72, 24, 27, 100, 0, 5)))
def test_fp16_output_layer_forward_is_exact_on_the_rounded_operands(case):
    """wcmc_split_to_f16 + wcmc_conv2d_out_f16 (the forward of an un-gated 5x5 output layer in the default mode "bf16x321h":
    ``conv_halo64_bf16x3_kernel<7, 3, PT, 0, 80, 1, 1, 1>`` on ``v_mfma_f32_16x16x32_f16``): both operands rounded ONCE to fp16 --
    x from its split value hi + lo, W by the mode-4 pack -- one MFMA per product.  fp16 x fp16 is exact in fp32, so the result
    must equal fp64 on the operands rounded the same way at the kernels' usual 2e-5 (bias included); fp16 subnormals take part
    (weights below 6.1e-5 are NOT flushed); values beyond +-65504 saturate; and the library refuses shapes without an instance."""
    from wcmc_amd._lib import lib
    o = ops()
    n, cin, h, w, cout, pad, ks = case
    assert lib().wcmc_conv2d_out_f16_supported(cin, cout, ks) == 1
    x = torch.relu(gen(n, cin, h, w, seed=590) * 3.0)
    wt = gen(cout, cin, ks, ks, seed=591, scale=(2.0 / (cin * ks * ks)) ** 0.5 * 1.7)
    b = gen(cout, seed=592, scale=0.2)
    xs = o.split_raw(o.to_nhwc_raw(x.to(DEV)))
    y = o.conv2d_out_f16_raw(xs, (n, cin, h, w), o._pack_x(wt.to(DEV), 4), b.to(DEV), cout, ks, pad)
    x16 = o.unsplit_debug(xs, n, cin, h, w).half().double().cpu()          # the kernel's operand: fp16 of the split value
    want = F.conv2d(x16, wt.half().double(), b.double(), padding=pad)
    assert_close(y, want, tol=2e-5, what="fp16 output layer = conv(fp16(x), fp16(W)) + b")
    full = F.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    assert 1e-5 < rel_err(y, full) < 1e-3                                   # (2^-12 per operand on i.i.d. test data)
    if case[0] == 1 and cout == 112:
        # subnormal weights are multiplied, not flushed: 3e-6 is below fp16's smallest normal (6.1e-5)
        tiny = torch.full((cout, cin, ks, ks), 3e-6)
        ones = o.split_raw(o.to_nhwc_raw(torch.ones(n, cin, h, w, device=DEV)))
        yt = o.conv2d_out_f16_raw(ones, (n, cin, h, w), o._pack_x(tiny.to(DEV), 4), None, cout, ks, pad)
        np.testing.assert_allclose(yt[0, 0, 0, 0].item(), float(torch.tensor(3e-6).half()) * cin * ks * ks, rtol=1e-5)
        # saturation instead of inf
        big = o.split_raw(o.to_nhwc_raw(torch.full((n, cin, h, w), 1e6, device=DEV)))
        wone = torch.zeros(cout, cin, ks, ks); wone[:, 0, 0, 0] = 1.0
        yb = o.conv2d_out_f16_raw(big, (n, cin, h, w), o._pack_x(wone.to(DEV), 4), None, cout, ks, pad)
        assert torch.isfinite(yb).all() and yb[0, 0, 0, 0].item() == 65504.0
    assert lib().wcmc_conv2d_out_f16_supported(100, 39, 5) == 0 and lib().wcmc_conv2d_out_f16_supported(120, 100, 5) == 0
    assert lib().wcmc_conv2d_out_f16_supported(64, 128, 3) == 0


PW_CASES = [
    # N, H, W, widths of a 1x1 chain, output activation -- the PathNet chains (support/networks.py:22-27)
    (5, 37, 41, (36, 64, 64, 64), "linear"),     # embedding: 7585 pixels = 118 tiles + 33 (ragged last tile)
    (2, 33, 29, (128, 128, 3), "relu"),          # final: 128 -> 128 persistent, 128 -> 3 tiled; gradients 3 -> 128 -> 128
    (1, 3, 3, (64, 64, 64), "relu"),             # 9 pixels: less than one tile, most workgroups idle
]


def _pw_chain(o, case, seed):
    n, h, w, widths, out_act = case
    acts = ["relu"] * (len(widths) - 2) + [out_act]
    x = gen(n, widths[0], h, w, seed=seed)
    params = []
    for l in range(len(widths) - 1):
        params.append(gen(widths[l + 1], widths[l], 1, 1, seed=seed + 1 + 2 * l, scale=(2.0 / widths[l]) ** 0.5 * 1.7))
        params.append(gen(widths[l + 1], seed=seed + 2 + 2 * l, scale=0.2))
    return x, params, acts


@pytest.mark.parametrize("case", PW_CASES)
def test_pointwise_chain_matches_fp64_and_the_tiled_kernel_bitwise(case, monkeypatch, three_term_mode):
    """The persistent 1x1 kernel (LDS-DMA ring, weights in registers) against an fp64 chain, and bit for bit
    against the tiled streaming kernel it replaces (same MFMA sequence per output): outputs, data gradients,
    weight gradients (which consume the hidden activations it wrote) and bias gradients (its column sums are
    grouped differently: tolerance)."""
    o = ops()
    x, params, acts = _pw_chain(o, case, seed=80)
    pr = [t.double().requires_grad_(True) for t in params]
    xr = x.double().requires_grad_(True)
    hcur, pres = xr, []
    for l, a in enumerate(acts):
        pre = F.conv2d(hcur, pr[2 * l], pr[2 * l + 1])
        pres.append(pre)
        hcur = om._activation(pre, a)
    kink = torch.ones_like(pres[-1][:, :1])
    for pre in pres:       # no gradient through pixels with a pre-activation within rounding of a kink
        kink = kink * (pre.detach().abs().amin(dim=1, keepdim=True) > 1e-5).double()
    gy = gen(*hcur.shape, seed=99).double() * kink
    hcur.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    pd = [t.to(DEV).requires_grad_(True) for t in params]
    y = o.conv_chain(xd, 1, 0, acts, pd)
    y.backward(gy.float().to(DEV))
    got = [y.detach().clone(), xd.grad.clone()] + [t.grad.clone() for t in pd]
    names = ["fwd", "dx"] + ["dw%d" % (i // 2) if i % 2 == 0 else "db%d" % (i // 2) for i in range(len(params))]
    want = [hcur, xr.grad] + [t.grad for t in pr]
    for nm, a, r in zip(names, got, want):
        assert_close(a, r, tol=1e-4, what="pointwise " + nm)


def test_pointwise_kernel_repeats_bitwise_at_benchmark_size(monkeypatch):
    """Race screen for the persistent 1x1 kernel at full occupancy (1 M pixels, every ring stage reused
    thousands of times): run-to-run identical.  (Its equality with the tiled kernel it replaced was held here through round 4;
    that kernel is reachable in the debug build only now.)"""
    o = ops()
    n, h = 64, 128
    xs = o.split_raw(o.to_nhwc_raw(gen(n, 64, h, h, seed=40).to(DEV)))
    w = gen(64, 64, 1, 1, seed=41, scale=0.2).to(DEV)
    b = gen(64, seed=42, scale=0.1).to(DEV)
    wp, wpt = o._pack_x(w, 0), o._pack_x(w, 1)

    def run():
        y, mask = o.conv2d_x_raw(xs, (n, 64, h, h), wp, b, 64, 1, 0, "relu", out_split=True, mask_out=True)
        dx, part = o.conv2d_x_raw(y, (n, 64, h, h), wpt, None, 64, 1, 0, "linear", out_split=True, gate_act="relu",
                                  gate_mask=mask, colsum=True)
        yf = o.conv2d_x_raw(xs, (n, 64, h, h), wp, b, 64, 1, 0, "relu", out_split=False)
        return y.clone(), mask.clone(), dx.clone(), o.colsum_finish_raw(part, (n, 64, h, h)).clone(), yf.clone()

    ref = run()
    for _ in range(6):
        cur = run()
        for a, bb, what in zip(ref, cur, ("split out", "mask", "gated dgrad", "column sums", "fp32 out")):
            assert torch.equal(a, bb), "the pointwise kernel does not repeat bit for bit in " + what
    del ref, cur


def test_dma_fed_gemms_repeat_bitwise_at_benchmark_size():
    """Race screen for the LDS-DMA staged kernels (halo igemm weight stages, filter-row wgrad stages): their
    LDS hand-offs are ordered by counted waits + barriers, and a read that beats its DMA shows up as a
    run-to-run difference at full occupancy long before it shows up in a small parity case."""
    o = ops()
    n, c, h = 8, 100, 116
    xs = o.split_raw(o.to_nhwc_raw(gen(n, c, h, h, seed=30).to(DEV)))
    dys = o.split_raw(o.to_nhwc_raw(gen(n, 100, h - 4, h - 4, seed=31).to(DEV)))
    w = gen(100, c, 5, 5, seed=32, scale=0.02).to(DEV)
    b = gen(100, seed=33, scale=0.1).to(DEV)
    wp, wpt, wpt2 = o._pack_x(w, 0), o._pack_x(w, 1), o._pack_x(w, 2)
    first = None
    for _ in range(12):
        y = o.conv2d_x_raw(xs, (n, c, h, h), wp, b, 100, 5, 0, "relu", out_split=True)
        dx = o.conv2d_x_raw(dys, (n, 100, h - 4, h - 4), wpt, None, c, 5, 4, "linear", out_split=True, gate=xs, gate_act="relu")
        dx2 = o.conv2d_x_raw(dys, (n, 100, h - 4, h - 4), wpt2, None, c, 5, 4, "linear", out_split=True, gate=xs, gate_act="relu", terms=2)
        dw, db = o.conv2d_wgrad_x_raw(xs, (n, c, h, h), dys, 100, 5, 0, (100, c, 5, 5), terms=3)
        dw1, _ = o.conv2d_wgrad_x_raw(xs, (n, c, h, h), dys, 100, 5, 0, (100, c, 5, 5), terms=1)
        cur = (y.clone(), dx.clone(), dw.clone(), db.clone(), dx2.clone(), dw1.clone())
        if first is None:
            first = cur
        else:
            for a, bb, what in zip(first, cur, ("fwd", "dgrad", "wgrad", "bias grad", "two-term dgrad", "one-term wgrad")):
                assert torch.equal(a, bb), "run-to-run difference in " + what


@pytest.mark.parametrize("n", [1, 2, 5, 64, 1000, 541696, 4333568])
def test_random_permutation_is_a_bijection_and_seeded(n):
    """The sort-free device permutation behind FeatureMSE(rng='device'): every index exactly once, reproducible
    under torch.manual_seed, different for different seeds, and not the identity."""
    o = ops()
    torch.manual_seed(5)
    a = o.random_permutation(n, DEV)
    assert a.dtype == torch.int64 and a.numel() == n
    assert torch.equal(torch.sort(a).values, torch.arange(n, device=DEV))
    torch.manual_seed(5)
    assert torch.equal(o.random_permutation(n, DEV), a)
    if n >= 64:
        b = o.random_permutation(n, DEV)                    # next key of the same generator
        assert not torch.equal(a, b)
        fixed = float((a == torch.arange(n, device=DEV)).float().mean())
        assert fixed < 0.1                                  # E[fixed points] = 1 for a uniform permutation
        # neighbours are scattered: mean |pi(i+1) - pi(i)| ~ n/3 for a uniform permutation
        gap = float((a[1:] - a[:-1]).abs().double().mean())
        assert 0.2 * n < gap < 0.45 * n


@pytest.mark.parametrize("act", ["relu", "leaky_relu"])
def test_gated_split_equals_act_backward_then_split(act):
    """wcmc_split_gated_bf16 (a chain's output-activation backward folded into the split of dy) is bit-identical to the
    two launches it replaces, on a ragged channel count and a strided dy view."""
    o = ops()
    dyb = o.to_nhwc_raw(gen(2, 21, 9, 11, seed=60).to(DEV))
    dy = dyb[:, :19]                                   # strided channel view (19 of 21 channels)
    post = o.to_nhwc_raw((gen(2, 19, 9, 11, seed=61)).to(DEV))
    want = o.split_raw(o.act_backward_raw(dy, post, act))
    got = o.split_gated_raw(dy, post, act)
    assert torch.equal(want, got)


def test_split_roundtrip_is_near_fp32():
    o = ops()
    x = o.to_nhwc_raw((gen(2, 37, 9, 11, seed=22) * 100).to(DEV))
    back = o.unsplit_debug(o.split_raw(x), 2, 37, 9, 11)
    assert_close(back, x, tol=2.0 ** -16, what="hi + lo")
    raw = o.split_raw(x).view(torch.bfloat16).view(2, 9, 11, 2, 40)
    assert (raw[..., 37:] == 0).all()                      # pad channels are zero


def test_conv_rejects_bad_views():
    o = ops()
    x = torch.zeros(1, 8, 4, 4, device=DEV)           # NCHW contiguous, not an NHWC view
    wp = torch.zeros(16 * 32, device=DEV)
    with pytest.raises(RuntimeError, match="NHWC-view contract"):
        o.conv2d_raw(x, wp, None, 4, 1, 0, "linear")
    with pytest.raises(RuntimeError, match="no CPU path"):
        o.conv_chain(torch.zeros(1, 4, 4, 4), 1, 0, ["linear"], [torch.zeros(4, 4, 1, 1), torch.zeros(4)])


# ---------------------------------------------------------------------------- kernel apply
@pytest.mark.parametrize("shape", [(2, 3, 19, 23, 21), (1, 3, 8, 8, 21), (2, 3, 12, 9, 5), (1, 2, 30, 33, 7)])
def test_kernel_apply_fwd_bwd(shape):
    n, c, h, w, k = shape
    o = ops()
    data = gen(n, c, h, w, seed=30) + 0.5
    logits = gen(n, k * k, h, w, seed=31, scale=3.0)
    dr, lr = data.double().requires_grad_(True), logits.double().requires_grad_(True)
    outr = om.kernel_apply(dr, lr)
    g = gen(*outr.shape, seed=32)
    outr.backward(g.double())
    dd, ld = data.to(DEV).requires_grad_(True), logits.to(DEV).requires_grad_(True)
    out = o.kernel_apply(dd, ld)
    out.backward(g.to(DEV))
    assert_close(out, outr, what="kernel_apply fwd")
    assert_close(ld.grad, lr.grad, what="kernel_apply d_logits")
    assert_close(dd.grad, dr.grad, what="kernel_apply d_data")


@pytest.mark.parametrize("shape", [(2, 3, 37, 45), (1, 3, 5, 7), (3, 2, 16, 16), (1, 1, 92, 92), (2, 3, 21, 130),
                                   (70, 3, 92, 20)])       # (the last one: blocks that own rows of two strips / two images)
@needs_debug_lib
def test_kernel_apply_strip_equals_tile_kernel(shape, monkeypatch):
    """The persistent strip kernel (LDS-DMA ring per wave; the one the KPCN path runs: k = 21, C <= 3, no d_data) against
    the tile kernel it replaced (WCMC_KA_TILE=1), forward (result + log-sum-exp) and backward (d_logits), bit for bit:
    same lane / tap assignment, same arithmetic, same reduction order.  Shapes: ragged strips (w % 16 != 0), an image
    smaller than a strip, more strips than rows, several images per block."""
    o = ops()
    n, c, h, w = shape
    data = (gen(n, c, h, w, seed=35) + 0.5).to(DEV)
    logits = o.as_nhwc((gen(n, 441, h, w, seed=36, scale=3.0)).to(DEV))
    g = gen(n, c, h, w, seed=37).to(DEV)
    res = []
    for tile in ("1", "0"):
        monkeypatch.setenv("WCMC_KA_TILE", tile)
        ld = o.nhwc_empty(n, 441, h, w, DEV)
        ld.copy_(logits)
        ld.requires_grad_(True)
        assert o.is_nhwc_view(ld)
        out = o._KernelApply.apply(data, ld)
        out.backward(g)
        torch.cuda.synchronize()
        res.append((out.detach().clone(), ld.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]), "forward differs"
    assert torch.equal(res[0][1], res[1][1]), "d_logits differs"
    ref = om.kernel_apply(data.double().cpu(), logits.double().cpu())
    assert_close(res[1][0], ref, what="strip kernel fwd vs oracle")


EMBED_CASES = [(6, 3, 36, 20, 23), (8, 4, 36, 32, 32), (4, 2, 64, 16, 24), (2, 1, 8, 9, 7), (16, 8, 36, 64, 64)]


def _embed_params(cin, seed):
    ws = [gen(64, cin, 1, 1, seed=seed, scale=(2.0 / cin) ** 0.5 * 1.7), gen(64, 64, 1, 1, seed=seed + 1, scale=(2.0 / 64) ** 0.5 * 1.7),
          gen(64, 64, 1, 1, seed=seed + 2, scale=(2.0 / 64) ** 0.5 * 1.7)]
    bs = [gen(64, seed=seed + 3 + i, scale=0.2) for i in range(3)]
    return [t for pair in zip(ws, bs) for t in pair]


@pytest.mark.parametrize("case", EMBED_CASES)
def test_fused_embedding_chain_forward_is_bit_identical_and_backward_matches_its_fp64_emulation(case, monkeypatch):
    """``wcmc_embed3_fwd`` / ``_bwd`` (PathNet.embedding + spp mean as one launch per direction, hidden activations on chip /
    recomputed) against the layer-by-layer path: the forward output and its spp mean BIT FOR BIT (same MFMA sequence per
    output); the backward against an fp64 evaluation that rounds where the kernel rounds -- dy, dh1, dh0, x, h0, h1 to bf16
    as MFMA operands of the two-term data gradients and one-term weight gradients, exact sums for the bias gradients --
    and against the unfused backward of the same mode within the mode's gradient tolerance.  Ragged last tile, S = 1, 8
    input channels, a gradient that is a channel slice of a wider tensor, a missing gradient of the mean."""
    from conftest import rel_l2
    o = ops()
    assert o.reduced_backward()
    n, s, cin, h, w = case
    x = gen(n, cin, h, w, seed=300)
    params = _embed_params(cin, 310)
    gwide = gen(n, 128, h, w, seed=320)                    # g_y arrives as the first 64 channels of the concatenation's gradient
    gm = gen(n // s, 64, h, w, seed=321)
    res = {}
    acts = []
    for fused in (True, False):
        monkeypatch.setattr(o, "FUSE_EMBED", fused)
        monkeypatch.setattr(o, "DEBUG_ACTS", None if fused else acts)      # the layer-by-layer path hands out its hidden activations
        xd = o.presplit_shared(x.to(DEV)) if cin <= 64 and n * h <= 65535 else o.as_nhwc(x.to(DEV))
        ps = [t.to(DEV).requires_grad_(True) for t in params]
        y, m = o.conv_chain_spp_mean(xd, s, 1, 0, ["relu", "relu", "linear"], ps)
        gw = o.to_nhwc_raw(gwide.to(DEV))
        torch.autograd.backward([y, m], [gw[:, :64], gm.to(DEV)])
        res[fused] = (y.detach().clone(), m.detach().clone(), [t.grad.clone() for t in ps])
    monkeypatch.setattr(o, "DEBUG_ACTS", None)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    # fp64 emulation of the fused backward ON THE PRODUCT'S OWN hidden activations (the fused forward is bit-identical, so
    # these are the h0 / h1 it recomputes): a hidden unit within rounding of zero may sit on the other side in an fp64
    # forward, and in these small cases ONE such unit moves a layer-0 gradient by ~1e-2 (measured)
    bf = lambda t: t.float().bfloat16().double()
    W0, b0, W1, b1, W2, b2 = [t.double() for t in params]
    X = x.double().permute(0, 2, 3, 1).reshape(-1, cin)
    assert len(acts) == 2
    h0, h1 = (a.detach().cpu().double().permute(0, 2, 3, 1).reshape(-1, 64) for a in acts)
    dy = gwide[:, :64].double().permute(0, 2, 3, 1).reshape(-1, 64) + \
        (gm.double() / s).unsqueeze(1).expand(n // s, s, 64, h, w).reshape(n, 64, h, w).permute(0, 2, 3, 1).reshape(-1, 64)
    dyh = bf(dy)
    dh1 = (dyh @ W2.view(64, 64)) * (bf(h1) > 0)
    dh1h = bf(dh1)
    dh0 = (dh1h @ W1.view(64, 64)) * (bf(h0) > 0)
    want = [bf(dh0).t() @ bf(X), dh0.sum(0), dh1h.t() @ bf(h0), dh1.sum(0), dyh.t() @ bf(h1), dy.sum(0)]
    for got, wnt, unf, name in zip(res[True][2], want, res[False][2], ("dw0", "db0", "dw1", "db1", "dw2", "db2")):
        e = rel_l2(got.reshape(wnt.shape), wnt)
        # (5e-4: a value the kernel forms in fp32 and the emulation in fp64 now and then rounds to the other bf16 neighbour,
        # 2^-8 of that element; measured 1e-7 .. 2.1e-4)
        assert e <= 5e-4, "%s: relative L2 %.3e against the fp64 emulation" % (name, e)
        assert rel_l2(got, unf) <= 8e-3, name                # the unfused backward of the same mode (three-term 1x1 data gradients)
    # the gradient of the mean alone / of y alone
    monkeypatch.setattr(o, "FUSE_EMBED", True)
    for which in ("y", "m"):
        ps = [t.to(DEV).requires_grad_(True) for t in params]
        xd = o.presplit_shared(x.to(DEV))
        y, m = o.conv_chain_spp_mean(xd, s, 1, 0, ["relu", "relu", "linear"], ps)
        (y * gen(n, 64, h, w, seed=330).to(DEV)).sum().backward() if which == "y" else (m * gm.to(DEV)).sum().backward()
        assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in ps)
        if which == "m":
            np.testing.assert_allclose(ps[5].grad.cpu().numpy(), gm.double().sum(dim=(0, 2, 3)).numpy(), rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("case", [(2, 3, 8, 8, 3), (1, 1, 16, 12, 4), (2, 8, 16, 16, 3), (8, 8, 64, 64, 3), (3, 2, 8, 24, 1),
                                  # round 4: up to eight output channels (--pnet_out_size 6 of the reference's m10r01 / m11r01 runs)
                                  (2, 8, 16, 16, 6), (1, 2, 8, 24, 8), (2, 3, 8, 8, 5), (8, 8, 32, 32, 6)])
def test_fused_final_chain_forward_is_bit_identical_and_backward_matches_its_fp64_emulation(case, monkeypatch):
    """``wcmc_final2_fwd`` / ``_bwd`` (PathNet.final with the broadcast concatenation, one launch per direction: concatenation
    and hidden activation on chip / recomputed, d_prop summed over the samples in registers) against the layer-by-layer path
    (``wcmc_cat_broadcast_split`` + the fused layer pair): output BIT FOR BIT; backward against an fp64 evaluation on the
    product's own hidden activation and output that rounds where the kernel rounds (d_out, dh, c, h to bf16 as operands of
    the two-term data gradients / one-term weight gradients; exact sums for the bias gradients), and against the unfused
    backward of the same mode."""
    from conftest import rel_l2
    o = ops()
    assert o.reduced_backward()
    b, s, h, w, outc = case
    flat = gen(b * s, 64, h, w, seed=400)
    prop = gen(b, 64, h, w, seed=401)
    params = [gen(128, 128, 1, 1, seed=402, scale=(2.0 / 128) ** 0.5 * 1.7), gen(128, seed=403, scale=0.2),
              gen(outc, 128, 1, 1, seed=404, scale=(2.0 / 128) ** 0.5 * 1.7), gen(outc, seed=405, scale=0.2)]
    g = gen(b * s, outc, h, w, seed=406)
    res, acts = {}, []
    for fused in (True, False):
        monkeypatch.setattr(o, "FUSE_FINAL", fused)
        monkeypatch.setattr(o, "DEBUG_ACTS", None if fused else acts)
        fd = o.to_nhwc_raw(flat.to(DEV)).requires_grad_(True)
        pd = o.to_nhwc_raw(prop.to(DEV)).requires_grad_(True)
        ps = [t.to(DEV).requires_grad_(True) for t in params]
        out = o.cat_broadcast_chain(fd, pd, s, 1, 0, ["relu", "relu"], ps)
        out.backward(g.to(DEV))
        res[fused] = (out.detach().clone(), [fd.grad.clone(), pd.grad.clone()] + [t.grad.clone() for t in ps])
    monkeypatch.setattr(o, "DEBUG_ACTS", None)
    from wcmc_amd._lib import lib
    assert lib().wcmc_final2_supported(64, 64, 128, outc, h * w) == 1
    if os.environ.get("WCMC_DEBUG_LIB") == "1" and os.environ.get("WCMC_F2W", "0") != "0":
        # (the debug library's wave-private forward experiment, final2w_fwd_kernel: h bit-identical, the output layer's 128 exact
        # products per output summed in another k-slot order)
        assert_close(res[True][0], res[False][0], tol=1e-6, what="fused final chain, output")
        assert float(((res[True][0] > 0) != (res[False][0] > 0)).float().mean()) <= 1e-4
    else:
        assert torch.equal(res[True][0], res[False][0])
    bf = lambda t: t.float().bfloat16().double()
    W0, b0, W1, b1 = [t.double() for t in params]
    M = b * s * h * w
    rows = lambda t, c: t.double().permute(0, 2, 3, 1).reshape(-1, c)
    C = torch.cat([rows(flat, 64), rows(prop.unsqueeze(1).expand(b, s, 64, h, w).reshape(b * s, 64, h, w), 64)], 1)
    assert len(acts) == 2
    H, O = rows(acts[0].detach().cpu(), 128), rows(acts[1].detach().cpu(), outc)
    do = rows(g, outc) * (O > 0)
    doh = bf(do)
    dh = (doh @ W1.view(outc, 128)) * (bf(H) > 0)
    dhh = bf(dh)
    dc = dhh @ W0.view(128, 128)
    dflat = dc[:, :64].reshape(b * s, h, w, 64).permute(0, 3, 1, 2)
    dprop = dc[:, 64:].reshape(b, s, h, w, 64).sum(1).permute(0, 3, 1, 2)
    want = [dflat, dprop, dhh.t() @ bf(C), dh.sum(0), doh.t() @ bf(H), do.sum(0)]
    for got, wnt, unf, name in zip(res[True][1], want, res[False][1], ("dflat", "dprop", "dw0", "db0", "dw1", "db1")):
        e = rel_l2(got.reshape(wnt.shape), wnt)
        assert e <= 5e-4, "%s: relative L2 %.3e against the fp64 emulation" % (name, e)
        assert rel_l2(got, unf) <= 8e-3, name


def test_kernel_apply_known_answers():
    o = ops()
    n, h, w, k = 1, 26, 29, 21
    data = (gen(n, 3, h, w, seed=33) + 1.0).to(DEV)
    uniform = torch.zeros(n, k * k, h, w, device=DEV)
    out = o.kernel_apply(data, uniform)
    box = F.avg_pool2d(F.pad(data.cpu(), (10, 10, 10, 10)), 21, 1)          # zero-padded box mean
    assert_close(out, box, tol=1e-5, what="uniform logits = zero-padded box mean")
    spike = torch.zeros(n, k * k, h, w, device=DEV)
    spike[:, 10 * 21 + 10] = 80.0                                           # centre tap
    assert_close(o.kernel_apply(data, spike), data, tol=1e-6, what="centre spike = identity")
    shifted = torch.zeros(n, k * k, h, w, device=DEV)
    shifted[:, 10 * 21 + 12] = 80.0                                         # dy=0, dx=+2
    want = torch.zeros_like(data)
    want[..., :, :-2] = data[..., :, 2:]
    assert_close(o.kernel_apply(data, shifted), want, tol=1e-6, what="tap order (dy,dx) row-major")
    # cropped (strided) radiance view, as KPCN.forward passes it
    big = (gen(n, 3, h + 8, w + 8, seed=34) + 1.0).to(DEV)
    view = big[..., 4:-4, 4:-4]
    lg = gen(n, k * k, h, w, seed=35).to(DEV)
    assert_close(o.kernel_apply(view, lg), o.kernel_apply(view.contiguous(), lg), tol=1e-7, what="strided data")


# ---------------------------------------------------------------------------- U-Net / PathNet glue
def test_pool_upsample_cat():
    o = ops()
    x = gen(2, 20, 12, 16, seed=40)
    xr = x.double().requires_grad_(True)
    pr = F.max_pool2d(xr, 2, 2)
    ur = F.interpolate(pr, scale_factor=2, mode="bilinear", align_corners=False)
    cr = torch.cat([ur, xr], 1)
    g = gen(*cr.shape, seed=41)
    cr.backward(g.double())
    xd = x.to(DEV).requires_grad_(True)
    xn = o.as_nhwc(xd)
    p = o.maxpool2(xn)
    u = o.upsample2(p)
    c = o.cat_channels(u, xn)
    c.backward(g.to(DEV))
    assert_close(p, pr, tol=1e-7, what="maxpool")
    assert_close(u, ur, tol=1e-6, what="upsample")
    assert_close(c, cr, tol=1e-6, what="cat")
    assert_close(xd.grad, xr.grad, tol=1e-6, what="pool/upsample/cat backward")


def test_maxpool2_skip_sums_both_gradients_in_the_pooling_backward():
    """``ops.maxpool2_skip`` (a U-Net level's skip + pooled copy as one node): outputs and the input gradient BIT FOR BIT those of
    the separate ``maxpool2`` node plus autograd's add -- also with a strided skip gradient (a channel slice of a wider tensor),
    an odd channel count, and either gradient missing."""
    o = ops()
    for n, c, h, w in ((2, 20, 12, 16), (1, 7, 8, 8), (3, 64, 32, 32)):
        x = gen(n, c, h, w, seed=46)
        gs_wide, gp = gen(n, c + 12, h, w, seed=47), gen(n, c, h // 2, w // 2, seed=48)
        res = []
        for fused in (False, True):
            xd = x.to(DEV).requires_grad_(True)
            xn = o.as_nhwc(xd)
            skip, pooled = o.maxpool2_skip(xn) if fused else (xn, o.maxpool2(xn))
            gw = o.to_nhwc_raw(gs_wide.to(DEV))
            torch.autograd.backward([skip, pooled], [gw[:, 4:4 + c], gp.to(DEV)])
            res.append((pooled.detach().clone(), xd.grad.clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (n, c, h, w)
        xr = x.double().requires_grad_(True)
        pr = F.max_pool2d(xr, 2, 2)
        torch.autograd.backward([xr * 1, pr], [gs_wide[:, 4:4 + c].double(), gp.double()])
        assert_close(res[1][1], xr.grad, tol=1e-6, what="skip + pool backward")
        for which in (0, 1):                                  # one of the two outputs unused
            xd = x.to(DEV).requires_grad_(True)
            outs = o.maxpool2_skip(o.as_nhwc(xd))
            (outs[which] * (gs_wide[:, 4:4 + c] if which == 0 else gp).to(DEV)).sum().backward()
            xr = x.double().requires_grad_(True)
            ((xr if which == 0 else F.max_pool2d(xr, 2, 2)) * (gs_wide[:, 4:4 + c] if which == 0 else gp).double()).sum().backward()
            assert_close(xd.grad, xr.grad, tol=1e-6, what="one output only")


def test_spp_mean_and_cat_broadcast():
    o = ops()
    b, s, c, h, w = 2, 3, 8, 6, 10
    flat = gen(b * s, c, h, w, seed=42)
    prop = gen(b, c, h, w, seed=43)
    fr, pr = flat.double().requires_grad_(True), prop.double().requires_grad_(True)
    mr = fr.view(b, s, c, h, w).mean(1)
    catr = torch.cat([fr, pr.unsqueeze(1).repeat(1, s, 1, 1, 1).view(b * s, c, h, w)], 1)
    g1, g2 = gen(*mr.shape, seed=44), gen(*catr.shape, seed=45)
    (mr * g1.double()).sum().backward()
    (catr * g2.double()).sum().backward()
    fd, pd = flat.to(DEV).requires_grad_(True), prop.to(DEV).requires_grad_(True)
    m = o.spp_mean(fd, s)
    cat = o.cat_broadcast(fd, pd, s)
    ((m * g1.to(DEV)).sum() + (cat * g2.to(DEV)).sum()).backward()
    assert_close(m, mr, tol=1e-6, what="spp mean")
    assert_close(cat, catr, tol=1e-7, what="cat broadcast")
    assert_close(fd.grad, fr.grad, tol=1e-6, what="d flat")
    assert_close(pd.grad, pr.grad, tol=1e-6, what="d prop")


@pytest.mark.parametrize("geom", [(2, 3, 8, 5, 6, 10), (1, 1, 16, 12, 7, 9), (2, 4, 64, 64, 9, 33)])
def test_fused_chain_glue_matches_the_separate_ops(geom):
    """The PathNet glue fusions (embedding chain + spp mean as one node, concatenation written as the final chain's
    split input) against the separate autograd ops they replace: same outputs, same parameter and input gradients
    up to the bf16x2 split of the fused gradient (2^-17 relative per element)."""
    o = ops()
    if o.PRECISION != "bf16x3":
        pytest.skip("fusions exist on the split-bf16 path only")
    b, s, c1, c2, h, w = geom
    x = gen(b * s, 20, h, w, seed=60)
    prop_in = gen(b, c2, h, w, seed=61)
    w1, b1 = gen(c1, 20, 1, 1, seed=62, scale=0.3), gen(c1, seed=63, scale=0.1)
    w2, b2 = gen(5, c1 + c2, 1, 1, seed=64, scale=0.2), gen(5, seed=65, scale=0.1)
    g_out, g_mean = gen(b * s, 5, h, w, seed=66), gen(b, c1, h, w, seed=67)
    res = []
    for fused in (False, True):
        ts = [t.to(DEV).requires_grad_(True) for t in (x, prop_in, w1, b1, w2, b2)]
        xd, pd, w1d, b1d, w2d, b2d = ts
        if fused:
            flat, mean = o.conv_chain_spp_mean(xd, s, 1, 0, ["linear"], [w1d, b1d])
            out = o.cat_broadcast_chain(flat, pd, s, 1, 0, ["relu"], [w2d, b2d])
        else:           # the separate nodes (what the exact-fp32 mode runs): chain, spp mean, concatenation, chain
            flat = o.conv_chain(xd, 1, 0, ["linear"], [w1d, b1d])
            mean = o.spp_mean(flat, s)
            out = o.conv_chain(o.cat_broadcast(flat, pd, s), 1, 0, ["relu"], [w2d, b2d])
        ((out * g_out.to(DEV)).sum() + (mean * g_mean.to(DEV)).sum()).backward()
        res.append([out.detach(), mean.detach()] + [t.grad for t in ts])
    names = ["final out", "spp mean", "d x", "d prop", "d w1", "d b1", "d w2", "d b2"]
    for a, bb, nm in zip(res[0], res[1], names):
        assert_close(bb, a, tol=2e-5, what="fused vs separate: " + nm)


@pytest.mark.parametrize("geom", [(2, 3, 7, 3, 9, 11), (1, 8, 5, 6, 16, 70), (2, 2, 24, 2, 5, 4)])
def test_sample_features_cat(geom):
    """interfaces.py:394-403 / 797-806: cat([features, P, repeat_S(P.var(1).mean(1) / S).detach()], 2), P possibly a
    channel slice (the disentanglement options hand the lower half to the denoiser)."""
    o = ops()
    b, s, c, cp, h, w = geom
    feat = gen(b, s, c, h, w, seed=50)
    pfull = gen(b, s, 2 * cp, h, w, seed=51)
    fr, pr = feat.double().requires_grad_(True), pfull.double().requires_grad_(True)
    psl = pr[:, :, :cp]
    pvar = psl.var(1).mean(1, keepdims=True) / s
    want = torch.cat([fr, psl, torch.stack([pvar] * s, axis=1).detach()], 2)
    g = gen(*want.shape, seed=52)
    want.backward(g.double())
    fd, pd = feat.to(DEV).requires_grad_(True), pfull.to(DEV).requires_grad_(True)
    got = o.sample_features_cat(fd, pd[:, :, :cp])
    got.backward(g.to(DEV))
    assert_close(got, want, tol=1e-6, what="sample cat")
    assert torch.equal(fd.grad.cpu(), fr.grad.float()) and torch.equal(pd.grad.cpu(), pr.grad.float())


@pytest.mark.parametrize("shape", [(8, 3, 92, 92), (2, 3, 20, 20), (1, 1, 5, 7), (3, 4, 33, 17)])
def test_fused_image_losses_match_torch(shape):
    """L1Loss / RelativeMSE of the (N,C,H,W) outputs as HIP ops (wcmc_image_loss_fwd, wcmc_l1_mean_bwd)
    against torch on strided views (crops of larger tensors, like the interface hands them over), forward and backward;
    repeated launches agree bit for bit (fixed-order reduction)."""
    o = ops()
    n, c, h, w = shape
    xb = (gen(n, c, h + 4, w + 6, seed=70) * 2).to(DEV)
    rb = (gen(n, c, h + 8, w + 8, seed=71) * 2).abs().to(DEV)
    rb[0, 0, 4, 4] = 0.0
    x = xb[:, :, 2:2 + h, 3:3 + w].detach().requires_grad_(True)
    ref = rb[:, :, 4:4 + h, 4:4 + w]
    with torch.no_grad():
        x.data[0, 0, 0, 0] = ref[0, 0, 0, 0]                     # an exact tie: sign(0) = 0 in the backward
    xr = x.detach().double().cpu().requires_grad_(True)
    rr = ref.double().cpu()
    lr = torch.nn.L1Loss()(xr, rr)
    (lr * 1.7).backward()
    l = o.l1_mean(x, ref)
    (l * 1.7).backward()
    assert_close(l, lr, tol=1e-6, what="l1 mean")
    assert_close(x.grad, xr.grad, tol=1e-6, what="l1 backward")
    assert x.grad[0, 0, 0, 0] == 0
    l1, rel = o.image_metrics(x, ref, 1e-2)
    want_rel = 0.5 * torch.mean((xr.detach() - rr) ** 2 / (rr ** 2 + 1e-2))
    assert_close(l1, lr, tol=1e-6, what="metrics l1")
    assert_close(rel, want_rel, tol=1e-6, what="relative mse")
    l1b, relb = o.image_metrics(x, ref, 1e-2)
    assert torch.equal(l1, l1b) and torch.equal(rel, relb) and torch.equal(l1, l.detach())
    from wcmc_amd.support.losses import RelativeMSE
    with torch.no_grad():
        assert torch.equal(RelativeMSE()(x, ref), rel)           # the module takes the same kernel when no gradient is needed
    x2 = x.detach().clone().requires_grad_(True)
    RelativeMSE()(x2, ref).backward()                            # ... and torch's expression when one is
    assert x2.grad is not None and torch.isfinite(x2.grad).all()


def test_image_losses_against_reference_goldens(golden_dir):
    """support/losses.py:245-320 on device tensors (RelativeMSE, SMAPE, TonemappedMSE, TonemappedRelativeMSE)."""
    from wcmc_amd.support import losses as pl
    d = np.load(os.path.join(golden_dir, "losses_image.npz"))
    ref = torch.from_numpy(d["ref"]).to(DEV)
    for name in ("RelativeMSE", "SMAPE", "TonemappedMSE", "TonemappedRelativeMSE"):
        x = torch.from_numpy(d["im"]).to(DEV).requires_grad_(True)
        loss = getattr(pl, name)()(x, ref)
        loss.backward()
        np.testing.assert_allclose(loss.item(), d[name], rtol=1e-5, err_msg=name)
        np.testing.assert_allclose(x.grad.cpu().numpy(), d[name + "_grad"], rtol=1e-4, atol=1e-9, err_msg=name)


@pytest.mark.parametrize("geom", [(2, 16, 8, 6, 10), (1, 128, 64, 16, 12), (3, 8, 5, 4, 2)])
def test_cat_upsample_chain_equals_upsample_then_cat(geom):
    """A U-Net level's right chain on cat([upsample2(deep), skip], 1): the kernel that upsamples inside the
    concatenation (wcmc_cat_upsample_split) gives the separate ops' results bit for bit, forward and backward."""
    o = ops()
    n, c1, c2, hd, wd = geom
    deep, skip = gen(n, c1, hd, wd, seed=90), gen(n, c2, 2 * hd, 2 * wd, seed=91)
    wt = [gen(12, c1 + c2, 3, 3, seed=92, scale=0.2), gen(12, seed=93, scale=0.1), gen(5, 12, 3, 3, seed=94, scale=0.2),
          gen(5, seed=95, scale=0.1)]
    g = gen(n, 5, 2 * hd, 2 * wd, seed=96)
    res = []
    for fused in (True, False):
        d, s_ = deep.to(DEV).requires_grad_(True), skip.to(DEV).requires_grad_(True)
        ps = [t.to(DEV).requires_grad_(True) for t in wt]
        if fused:
            y = o.cat_upsample_chain(d, s_, 3, 1, ["relu", "leaky_relu"], ps)
        else:
            y = o.cat_broadcast_chain(o.upsample2(d), s_, 1, 3, 1, ["relu", "leaky_relu"], ps)
        y.backward(g.to(DEV))
        res.append([y.detach().clone(), d.grad.clone(), s_.grad.clone()] + [t.grad.clone() for t in ps])
    for a, b, nm in zip(res[0], res[1], ["y", "d_deep", "d_skip", "dw0", "db0", "dw1", "db1"]):
        assert torch.equal(a, b), "fused upsample-concat differs in " + nm
    # and against torch's bilinear upsampling (align_corners=False) in fp64
    up = F.interpolate(deep.double(), scale_factor=2, mode="bilinear", align_corners=False)
    x = torch.cat([up, skip.double()], 1)
    h1 = F.relu(F.conv2d(x, wt[0].double(), wt[1].double(), padding=1))
    want = F.leaky_relu(F.conv2d(h1, wt[2].double(), wt[3].double(), padding=1), 0.01)
    assert_close(res[0][0], want, tol=1e-4, what="cat-upsample chain vs fp64")


@pytest.mark.parametrize("cp", [3, 2, 6])
def test_pbuffer_cat(cp):
    o = ops()
    b, s, cb, h, w = 2, 4, 35, 10, 70
    base = gen(b, cb, h, w, seed=46)
    p = gen(b, s, cp, h, w, seed=47) + 1.0
    pr = p.double().requires_grad_(True)
    outr = assemble_input(base.double(), pr)
    g = gen(*outr.shape, seed=48)
    outr.backward(g.double())
    pd = o.to_nhwc_raw(p.view(b * s, cp, h, w).to(DEV)).unflatten(0, (b, s)).requires_grad_(True)
    out = o.pbuffer_cat(base.to(DEV), pd)
    out.backward(g.to(DEV))
    assert_close(out, outr, tol=2e-6, what="pbuffer cat")
    assert_close(pd.grad, pr.grad, tol=1e-6, what="pbuffer cat backward")
    # a channel-sliced view (disentanglement) of a wider P-buffer
    wide = gen(b, s, cp + 3, h, w, seed=49).to(DEV)
    assert_close(o.pbuffer_cat(base.to(DEV), wide[:, :, :cp]),
                 assemble_input(base, wide[:, :, :cp].cpu()), tol=2e-6, what="sliced P")


# ---------------------------------------------------------------------------- FeatureMSE
def test_feature_mse_against_reference_goldens(golden_dir):
    o = ops()
    d = np.load(os.path.join(golden_dir, "losses_fmse.npz"))
    for i in range(int(d["n"])):
        p = torch.from_numpy(d["p_%d" % i]).to(DEV).requires_grad_(True)
        ref = torch.from_numpy(d["ref_%d" % i]).to(DEV)
        ip = torch.from_numpy(d["idx_patch_%d" % i]).to(DEV)
        ib = torch.from_numpy(d["idx_batch_%d" % i]).to(DEV) if bool(d["non_local_%d" % i]) else None
        loss = o.feature_mse(p, ref, ip, ib)
        loss.backward()
        np.testing.assert_allclose(loss.item(), d["loss_%d" % i], rtol=1e-5)
        assert_close(p.grad, torch.from_numpy(d["grad_%d" % i]), tol=1e-5, what="FeatureMSE grad %d" % i)


def test_grs_against_reference_goldens(golden_dir):
    from wcmc_amd.support.losses import GlobalRelativeSimilarityLoss
    o = ops()
    d = np.load(os.path.join(golden_dir, "losses_grs.npz"))
    for i in range(int(d["n"])):
        p = torch.from_numpy(d["p_%d" % i]).to(DEV).requires_grad_(True)
        ref = torch.from_numpy(d["ref_%d" % i]).to(DEV)
        ip, ib = torch.from_numpy(d["idx_patch_%d" % i]).to(DEV), torch.from_numpy(d["idx_batch_%d" % i]).to(DEV)
        loss = o.grs_loss(p, ref, ip, ib, 2.0)
        loss.backward()
        np.testing.assert_allclose(loss.item(), d["loss_%d" % i], rtol=1e-5)
        assert_close(p.grad, torch.from_numpy(d["grad_%d" % i]), tol=1e-5, what="GRS grad %d" % i)
    # module form draws like the reference (patch then batch on the CPU generator)
    torch.manual_seed(2000)
    m = GlobalRelativeSimilarityLoss()
    loss = m(torch.from_numpy(d["p_0"]).to(DEV), torch.from_numpy(d["ref_0"]).to(DEV))
    np.testing.assert_allclose(loss.item(), d["loss_0"], rtol=1e-5)


def test_feature_mse_module_strided_and_seeded(golden_dir):
    from wcmc_amd.support.losses import FeatureMSE
    d = np.load(os.path.join(golden_dir, "losses_fmse.npz"))
    i = 0
    p = torch.from_numpy(d["p_%d" % i])
    b, s, c, h, w = p.shape
    o = ops()
    pn = o.to_nhwc_raw(p.view(b * s, c, h, w).to(DEV)).unflatten(0, (b, s))      # NHWC-backed, as PathNet returns
    torch.manual_seed(int(d["seed_%d" % i]))
    loss = FeatureMSE(non_local=True)(pn, torch.from_numpy(d["ref_%d" % i]).to(DEV))
    np.testing.assert_allclose(loss.item(), d["loss_%d" % i], rtol=1e-5)
    bad = pn.clone()
    bad[0, 0, 0, 0, 0] = float("nan")
    with pytest.raises(RuntimeError, match="Infinite loss at train time."):
        FeatureMSE(non_local=True)(bad, torch.from_numpy(d["ref_%d" % i]).to(DEV))


def test_feature_mse_full_size_properties():
    """B=8,S=8,92x92 (N = 541,696 rows): identity pairing gives exactly 0; the loss is invariant to
    relabelling pairs (pi -> pi^-1 gives the same multiset of pairs)."""
    o = ops()
    b, s, c, h, w = 8, 8, 3, 92, 92
    p = (gen(b * s, c, h, w, seed=50) + 1).to(DEV)
    pn = o.to_nhwc_raw(p).unflatten(0, (b, s))
    ref = (gen(b, 3, h, w, seed=51) + 1).to(DEV)
    ident_p = torch.arange(s * h * w, device=DEV)
    ident_b = torch.arange(b * s * h * w, device=DEV)
    assert o.feature_mse(pn, ref, ident_p, ident_b).item() == 0.0
    g = torch.Generator().manual_seed(52)
    ip, ib = torch.randperm(s * h * w, generator=g).to(DEV), torch.randperm(b * s * h * w, generator=g).to(DEV)
    inv_p, inv_b = torch.empty_like(ip), torch.empty_like(ib)
    inv_p[ip] = torch.arange(ip.numel(), device=DEV)
    inv_b[ib] = torch.arange(ib.numel(), device=DEV)
    l1, l2 = o.feature_mse(pn, ref, ip, ib).item(), o.feature_mse(pn, ref, inv_p, inv_b).item()
    np.testing.assert_allclose(l1, l2, rtol=1e-5)
    want = ol.feature_mse(p.cpu().view(b, s, c, h, w), ref.cpu(), ip.cpu(), ib.cpu()).item()
    np.testing.assert_allclose(l1, want, rtol=1e-4)


# ---------------------------------------------------------------------------- optimiser
def test_clip_adam_matches_torch():
    o = ops()
    n = 100003
    p0, g0 = gen(n, seed=60), gen(n, seed=61, scale=3.0)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        pr.grad = g0.clone() * step
        torch.nn.utils.clip_grad_value_([pr], 1.0)
        opt.step()
        g = (g0 * step * 2).to(DEV)                       # grad_scale=0.5 undoes the factor 2
        o.clip_adam_(p, g, m, v, step, 1e-3, grad_scale=0.5)
        assert torch.equal(g.cpu(), pr.grad)              # clipped gradient left behind, like the reference
    assert_close(p, pr, tol=1e-6, what="clip+adam params")
    st = opt.state[pr]
    assert_close(m, st["exp_avg"], tol=1e-6, what="exp_avg")
    assert_close(v, st["exp_avg_sq"], tol=1e-6, what="exp_avg_sq")
    # a zero guard makes the launch a no-op (non-finite loss upstream)
    before = (p.clone(), m.clone(), v.clone())
    o.clip_adam_(p, (g0 * 7).to(DEV), m, v, 4, 1e-3, guard=torch.zeros((), device=DEV))
    assert torch.equal(p, before[0]) and torch.equal(m, before[1]) and torch.equal(v, before[2])
    o.clip_adam_(p, (g0 * 7).to(DEV), m, v, 4, 1e-3, guard=torch.ones((), device=DEV))
    assert not torch.equal(p, before[0])
    # a NaN gradient reaches the parameter like through clip_grad_value_ (torch.clamp propagates NaN) + Adam; a
    # min/max clamp would turn it into -clip and train on silently.  Entries 5 (vector body) and n-1 (scalar tail).
    gn = g0.clone()
    gn[5] = gn[n - 1] = float("nan")
    pr.grad = gn.clone()
    torch.nn.utils.clip_grad_value_([pr], 1.0)
    opt.step()
    o.clip_adam_(p, gn.to(DEV), m, v, 5, 1e-3)
    for i in (5, n - 1):
        assert torch.isnan(pr[i]) and torch.isnan(p[i]) and torch.isnan(m[i])
    assert int(torch.isnan(p).sum()) == 2


# ------------------------------------------------------------------------ weight normalisation (round 5)
def test_weight_norm_multi_matches_torch_weight_norm():
    """``wcmc_weight_norm_fwd`` / ``_bwd`` (all layers of a model in one launch each) against ``torch._weight_norm`` in fp64:
    PathNet's layer shapes plus a row length that is not a multiple of four (the scalar path) and a one-row layer."""
    o = ops()
    shapes = [(64, 36, 1, 1), (64, 64, 3, 3), (128, 384, 3, 3), (3, 128, 1, 1), (5, 7, 3, 3), (1, 9, 1, 1), (256, 256, 3, 3)]
    gs, vs, rg, rv = [], [], [], []
    for i, shp in enumerate(shapes):
        v = gen(*shp, seed=100 + i)
        g = gen(shp[0], 1, 1, 1, seed=200 + i) + 1.5
        vs.append(v.to(DEV).requires_grad_(True)); gs.append(g.to(DEV).requires_grad_(True))
        rv.append(v.double().requires_grad_(True)); rg.append(g.double().requires_grad_(True))
    ws = o.weight_norm_multi(gs, vs)
    wr = [torch._weight_norm(v, g, 0) for v, g in zip(rv, rg)]
    dws = [gen(*shp, seed=300 + i) for i, shp in enumerate(shapes)]
    torch.autograd.backward(ws, [d.to(DEV) for d in dws])
    torch.autograd.backward(wr, [d.double() for d in dws])
    for i, shp in enumerate(shapes):
        assert ws[i].shape == torch.Size(shp) and ws[i].is_contiguous()
        assert_close(ws[i], wr[i], tol=2e-6, what="weight_norm fwd %s" % (shp,))
        assert_close(vs[i].grad, rv[i].grad, tol=5e-6, what="weight_norm dv %s" % (shp,))
        assert_close(gs[i].grad, rg[i].grad, tol=5e-6, what="weight_norm dg %s" % (shp,))
    # an unused layer's gradient stays None (as torch.autograd leaves it), the others are unaffected
    for t in gs + vs:
        t.grad = None
    ws = o.weight_norm_multi(gs, vs)
    (ws[1] * dws[1].to(DEV)).sum().backward()
    assert vs[0].grad is None and gs[2].grad is None
    assert_close(vs[1].grad, rv[1].grad, tol=5e-6, what="weight_norm dv, one live layer")


def test_weight_norm_rejects_bad_arguments():
    from wcmc_amd._lib import lib
    import ctypes
    h = lib()
    one_p, one_i = (ctypes.c_void_p * 1)(0), (ctypes.c_int * 1)(4)
    assert h.wcmc_weight_norm_fwd(0, one_p, one_p, one_p, one_p, one_i, one_i, None) == -1
    assert h.wcmc_weight_norm_fwd(33, one_p, one_p, one_p, one_p, one_i, one_i, None) == -1
    assert h.wcmc_weight_norm_fwd(1, one_p, one_p, one_p, one_p, one_i, one_i, None) == -1          # null layer pointers
    assert b"layer 0" in h.wcmc_last_error()
    assert h.wcmc_weight_norm_bwd(1, one_p, one_p, one_p, one_p, one_p, one_p, one_i, one_i, None) == -1


@pytest.mark.parametrize("weight_norm", [False, True])
def test_chain_applied_twice_in_one_backward_accumulates_both_weight_gradients(weight_norm, precision):
    """with gradient sinks registered (FusedClipAdam), a chain applied to TWO inputs inside one autograd engine
    run handed the same bucket view to both nodes -- the later node overwrote the earlier one's dw and the engine summed two
    aliases (2 * dw_B instead of dw_A + dw_B).  A sink is now handed out once per accumulation window."""
    from wcmc_amd.modules import ConvChain
    from wcmc_amd.optim import FusedClipAdam
    o = ops()
    torch.manual_seed(31)
    ref = om.ConvChain(12, 10, ksize=3, width=16, depth=2, pad=True, output_type="relu", weight_norm=weight_norm)
    mod = ConvChain(12, 10, ksize=3, width=16, depth=2, pad=True, output_type="relu", weight_norm=weight_norm)
    mod.load_state_dict(ref.state_dict())
    mod.to(DEV)
    ref = ref.double()
    opt = {"optim_m": torch.optim.Adam(mod.parameters(), lr=1e-3)}
    fo = FusedClipAdam({"m": mod}, opt)                       # registers the sinks
    xa, xb = gen(2, 12, 20, 24, seed=32).to(DEV), (gen(2, 12, 20, 24, seed=33) * 3.0).to(DEV)
    ga, gb = gen(2, 10, 20, 24, seed=34).to(DEV), gen(2, 10, 20, 24, seed=35).to(DEV)
    # each use on its own (one producer per parameter: the sinks' normal case), the optimiser's window closed in between
    alone = []
    for x, g in ((xa, ga), (xb, gb)):
        mod.zero_grad()
        o.release_grad_sinks(list(mod.parameters()))
        (mod(x) * g).sum().backward()
        alone.append({k: p.grad.detach().clone() for k, p in mod.named_parameters()})
    mod.zero_grad()
    o.release_grad_sinks(list(mod.parameters()))
    ((mod(xa) * ga).sum() + (mod(xb) * gb).sum()).backward()               # both uses in ONE engine run
    for k, p in mod.named_parameters():
        want = alone[0][k] + alone[1][k]
        assert_close(p.grad, want, tol=1e-6, what="twice-applied chain: grad %s is not the sum of the two uses' gradients" % k)
    if precision == "fp32":                                               # (and the sum is the right one: fp64 torch autograd)
        (ref(xa.cpu().double()) * ga.cpu().double()).sum().add((ref(xb.cpu().double()) * gb.cpu().double()).sum()).backward()
        named_r = dict(ref.named_parameters())
        for k, p in mod.named_parameters():
            assert_close(p.grad, named_r[k].grad, tol=2e-5, what="twice-applied chain grad " + k)
    # the optimiser still finds every gradient (sinks and fresh tensors alike) and a second window hands the sinks out again
    fo.step({"m": mod}, opt)
    mod.zero_grad()
    (mod(xa) * ga).sum().backward()
    views = {p.data_ptr(): v.data_ptr() for p, v in zip(fo.flats["m"].params, fo.flats["m"].grad_views())}
    in_views = lambda: all(p.grad.data_ptr() == views[p.data_ptr()] for p in mod.parameters())
    assert not o.split_path() or in_views(), "single use: every gradient lands in its bucket view"
    # ... and again after a zero_grad() WITHOUT an optimiser step in between (a graphed step's warm-up passes do exactly that):
    # the tensor that was handed out is gone, so the sink is free
    mod.zero_grad()
    (mod(xb) * gb).sum().backward()
    assert not o.split_path() or in_views(), "zero_grad() frees the sinks"


def test_deferred_multi_layer_slab_reduction_is_bit_identical(three_term_mode):
    """``ops.deferred_wgrad_reduce()``: the slab reductions of a chain's small layers collected and run as ONE launch
    (``wcmc_conv2d_wgrad_reduce_multi``) give the weight AND bias gradients of the per-layer reductions bit for bit -- a U-Net chain
    (3x3, padded), a 1x1 chain and a chain whose last layer has three outputs; a layer above ``DEFER_MAX_BYTES`` is not deferred."""
    from wcmc_amd.modules import ConvChain
    from wcmc_amd.optim import FusedClipAdam
    o = ops()
    for (cin, cout, ks, width, depth, hw) in ((64, 64, 3, 64, 3, (40, 36)), (384, 128, 3, 128, 3, (24, 24)), (36, 3, 1, 64, 3, (32, 64))):
        torch.manual_seed(7)
        mod = ConvChain(cin, cout, ksize=ks, width=width, depth=depth, pad=True, output_type="relu", weight_norm=False).to(DEV)
        # (gradient sinks registered, as in the product's step: a reduction is deferred only into memory that is certain to be the
        # gradient's when it runs -- a bucket view, or a weight-normalised layer's effective weight)
        fo = FusedClipAdam({"m": mod}, {"optim_m": torch.optim.Adam(mod.parameters(), lr=1e-3)})
        x = gen(4, cin, *hw, seed=70).to(DEV)
        g = gen(4, cout, *hw, seed=71).to(DEV)
        res = []
        for defer in (False, True):
            mod.zero_grad()
            o.release_grad_sinks(list(mod.parameters()))
            y = mod(x)
            if defer:
                with o.deferred_wgrad_reduce():
                    y.backward(g)
                    pend = sum(len(v[1]) for v in o._DEFERRED.values())
                assert 2 <= pend <= depth, "the chain's small layers should have been deferred: %d of %d were" % (pend, depth)
                assert o._DEFERRED is None
            else:
                y.backward(g)
            res.append([p.grad.clone() for p in mod.parameters()])
        for a, b, (k, _) in zip(res[0], res[1], mod.named_parameters()):
            assert torch.equal(a, b), "deferred reduction changes %s" % k
    old = o.DEFER_MAX_BYTES
    o.DEFER_MAX_BYTES = 1
    try:
        with o.deferred_wgrad_reduce():
            mod.zero_grad()
            o.release_grad_sinks(list(mod.parameters()))
            mod(x).backward(g)
            assert not o._DEFERRED
    finally:
        o.DEFER_MAX_BYTES = old
    del fo


@pytest.mark.parametrize("weight_norm", [False, True])
def test_deferred_reduction_never_writes_into_memory_that_is_not_the_gradients(weight_norm):
    """a deferred slab reduction writes dw / db long after the weight-gradient GEMM returned.  It may do so only into
    memory that is certain to be the gradient's by then: (i) parameters WITHOUT a registered bucket view, (ii) parameters that already
    hold a .grad (accumulation: autograd adds the returned tensor into .grad and frees it at once) and (iii) parameters that need no
    gradient are reduced inline; (iv) the weight-gradient side stream (``USE_SIDE_STREAM``) never queues an entry the weight-norm
    node's flush on the main stream would miss.  Every case: gradients equal to the un-deferred run bit for bit."""
    from wcmc_amd.modules import ConvChain
    from wcmc_amd.optim import FusedClipAdam
    o = ops()
    torch.manual_seed(9)
    mod = ConvChain(64, 64, ksize=3, width=64, depth=3, pad=True, output_type="relu", weight_norm=weight_norm).to(DEV)
    x = gen(4, 64, 40, 36, seed=72).to(DEV)
    g = gen(4, 64, 40, 36, seed=73).to(DEV)

    def grads(defer, passes=1, side=False):
        mod.zero_grad()
        o.release_grad_sinks(list(mod.parameters()))
        old = o.USE_SIDE_STREAM
        o.USE_SIDE_STREAM = side
        try:
            for _ in range(passes):
                y = mod(x)
                if defer:
                    with o.deferred_wgrad_reduce():
                        y.backward(g)
                else:
                    y.backward(g)
        finally:
            o.USE_SIDE_STREAM = old
        torch.cuda.synchronize()
        return [None if p.grad is None else p.grad.clone() for p in mod.parameters()]

    def same(a, b, what):
        for u, v, (k, _) in zip(a, b, mod.named_parameters()):
            assert (u is None) == (v is None) and (u is None or torch.equal(u, v)), "%s: %s differs" % (what, k)

    # (i) no sinks registered at all
    same(grads(False), grads(True), "no bucket views")
    # (ii) accumulation over two backward passes, without and with bucket views
    same(grads(False, passes=2), grads(True, passes=2), "accumulation, no bucket views")
    fo = FusedClipAdam({"m": mod}, {"optim_m": torch.optim.Adam(mod.parameters(), lr=1e-3)})
    same(grads(False, passes=2), grads(True, passes=2), "accumulation, bucket views")
    # (iii) a frozen layer
    frozen = [list(mod.parameters())[2]]
    for p in frozen:
        p.requires_grad_(False)
    try:
        same(grads(False), grads(True), "frozen parameter")
    finally:
        for p in frozen:
            p.requires_grad_(True)
    # (iv) weight gradients on the side stream
    same(grads(False, side=True), grads(True, side=True), "side stream")
    same(grads(False), grads(True, side=True), "side stream vs main stream")
    del fo


@pytest.mark.parametrize("kind,cls", [("smape", "SMAPE"), ("tonemapped_mse", "TonemappedMSE"), ("tonemapped_relative_mse", "TonemappedRelativeMSE")])
def test_sample_interface_losses_match_the_oracle(kind, cls):
    """SMAPE / TonemappedMSE / TonemappedRelativeMSE (support/losses.py:267-320) as HIP passes: value and dL/dx against the oracle's
    torch expressions in fp64 -- negative pixels (the tone map clamps them: zero gradient), a strided (cropped) operand, and through
    the ``support.losses`` classes the interfaces build."""
    from wcmc_amd.support import losses as hl
    o = ops()
    x = gen(2, 3, 37, 45, seed=80, scale=2.0) + 0.3
    ref = gen(2, 3, 37, 45, seed=81, scale=2.0).abs()
    big = gen(2, 3, 41, 49, seed=82, scale=2.0)
    for xin in (x, big[:, :, 2:39, 3:48] + 0.3):
        xr = xin.double().requires_grad_(True)
        want = getattr(ol, cls)()(xr, ref.double())
        want.backward()
        src = (big.to(DEV)[:, :, 2:39, 3:48] + 0.3 if xin is not x else x.to(DEV)).requires_grad_(True)
        got = getattr(hl, cls)()(src, ref.to(DEV))
        got.backward()
        assert_close(got.reshape(1), want.reshape(1), tol=2e-6, what=cls)
        assert_close(src.grad, xr.grad, tol=2e-6, what=cls + " gradient")
    assert torch.equal(o.image_loss2(x.to(DEV), ref.to(DEV), kind), o.image_loss2(x.to(DEV), ref.to(DEV), kind))


def test_clip_grad_norm_matches_torch():
    o = ops()
    shapes = [(64, 36, 1, 1), (64,), (128, 128, 3, 3), (3,), (5000,), (1,)]
    for scale, max_norm in ((1.0, 1000.0), (300.0, 250.0)):
        ps, pr = [], []
        for i, shp in enumerate(shapes):
            g = gen(*shp, seed=300 + i) * scale
            p = torch.nn.Parameter(torch.zeros(shp, device=DEV)); p.grad = g.to(DEV)
            q = torch.nn.Parameter(torch.zeros(shp, dtype=torch.float64)); q.grad = g.double()
            ps.append(p); pr.append(q)
        want = torch.nn.utils.clip_grad_norm_(pr, max_norm)
        got = o.clip_grad_norm_(ps, max_norm)
        assert_close(got.reshape(1), want.reshape(1), tol=2e-6, what="total norm")
        for p, q in zip(ps, pr):
            assert_close(p.grad, q.grad, tol=2e-6, what="clipped gradient")
        assert (float(want) > max_norm) == (scale > 1.0)
    # a non-finite gradient poisons every gradient, as torch's clamp(NaN) = NaN does
    ps = [torch.nn.Parameter(torch.zeros(shp, device=DEV)) for shp in shapes]
    for i, p in enumerate(ps):
        p.grad = gen(*shapes[i], seed=310 + i).to(DEV)
    ps[2].grad[5, 7, 1, 1] = float("nan")
    got = o.clip_grad_norm_(ps, 1000.0)
    assert torch.isnan(got).all() and all(torch.isnan(p.grad).all() for p in ps)


VARIANT_EXPR = "variant or switch_matrix or strip_equals or eight_wave or many_slabs or one_term_weight or three_term_forward"


def test_kernel_cross_checks_run_against_the_debug_build_in_a_subprocess():
    """The kernel A/B switches exist in the DEBUG build of the library only (csrc/common.h: ab_env), so the tests that hold a shipped
    kernel bit for bit against the kernel it replaced (`needs_debug_lib`; skipped in this process) are run here in a child process
    that loads ``libwcmc_hip_debug.so`` (``__graft_entry__.build()`` makes it beside the release library): the filter-row weight
    gradient against the one-tap kernel, eight against seven waves, the strip kernel-apply against the tile kernel, every entry of the
    switch matrix, ...  (A child process, never an exec: this one has initialised the GPU.)"""
    import subprocess
    import sys
    if DEBUG_LIB:
        pytest.skip("this process already runs against the debug library")
    from wcmc_amd._lib import LIB_PATH
    dbg = os.path.join(os.path.dirname(LIB_PATH), "libwcmc_hip_debug.so")
    if not os.path.isfile(dbg):
        pytest.skip("libwcmc_hip_debug.so has not been built (make -C wcmc_amd/csrc debug)")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, WCMC_DEBUG_LIB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider", "-k", VARIANT_EXPR],
                       env=env, capture_output=True, text=True, timeout=1200)
    tail = (r.stdout or "")[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    print(tail.strip().splitlines()[-1])
