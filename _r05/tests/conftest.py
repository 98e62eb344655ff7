import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The oracle runs on host cores; the GPU box shows 256 logical CPUs and PyTorch's default of one
# thread per CPU is far slower there than a modest pool.
import torch  # noqa: E402
torch.set_num_threads(min(16, os.cpu_count() or 1))


# Kernel A/B switches (WCMC_* variables that select a non-default kernel, tiling or schedule) exist in the DEBUG build of the library
# only (csrc/common.h: ab_env; `make -C wcmc_amd/csrc debug`, loaded with WCMC_DEBUG_LIB=1).  The tests that hold a shipped kernel
# bit for bit against the kernel it replaced need that build; against the release library they are skipped (and their variant
# legs collapse to the shipped plan); tests/test_gpu_ops.py::test_kernel_cross_checks_run_against_the_debug_build_in_a_subprocess runs
# them in a child process that loads the debug library:   WCMC_DEBUG_LIB=1 python -m pytest tests -m gpu -k "variant or switch_matrix or ..."
DEBUG_LIB = os.environ.get("WCMC_DEBUG_LIB") == "1"
needs_debug_lib = pytest.mark.skipif(not DEBUG_LIB, reason="kernel A/B switches exist in the debug build only (make -C wcmc_amd/csrc debug; WCMC_DEBUG_LIB=1)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


REDUCED = ("bf16x321h", "bf16x321", "bf16x321o")          # modes whose backward GEMMs run on two / one bf16 MFMAs per product


@pytest.fixture(params=["fp32", "bf16x3", "bf16x321", "bf16x321o", "bf16x321h"])
def precision(request):
    """Runs a GPU test once per conv arithmetic: exact fp32 MFMA, split-bf16 with 3 bf16 MFMAs per product everywhere, round 3's
    default -- still the default -- (forward 3, data gradient 2, weight gradient 1), and the two opt-in modes that run the forward
    of un-gated 5x5 output layers on ONE bf16 / ONE fp16 MFMA per product (wcmc_amd/ops.py)."""
    from wcmc_amd import ops
    old = ops.PRECISION
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision(old)


@pytest.fixture
def three_term_mode():
    """Tests that compare KERNELS with each other (bitwise or at 1e-5) run them on identical arithmetic: three MFMAs per
    product in every role ("bf16x3"); the reduced-term kernels of the default mode have their own exactness tests."""
    from wcmc_amd import ops
    old = ops.PRECISION
    ops.set_precision("bf16x3")
    yield
    ops.set_precision(old)


@pytest.fixture(scope="session")
def rccl_one_rank_group():
    """ONE one-rank RCCL process group for the whole pytest session (torch.distributed backend "nccl"), destroyed when the
    session ends.  A second init / destroy cycle of RCCL inside one process left the HIP runtime in a state in which a later,
    unrelated hipGraphLaunch segfaulted (full GPU suite, round 4): tests that need the communicator share this one."""
    import torch.distributed as dist
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", store=dist.HashStore(), rank=0, world_size=1)
    yield dist.group.WORLD
    if own:
        dist.destroy_process_group()


def ptol(precision, fp32_tol, x3_tol):
    return fp32_tol if precision == "fp32" else x3_tol


def gtol(precision, fp32_tol, x3_tol, x321_tol=8e-3):
    """Max-norm bar of a GRADIENT of one conv op on random operands.  In the "bf16x321" mode the backward GEMMs round dy (and,
    in the weight gradient, x) to bf16: 2^-9 per rounded operand, uncorrelated from pixel to pixel -- on the i.i.d. test
    operands that shows in full (the sums are random walks too), hence 8e-3 of the tensor's max; in the networks it averages
    out (profiles/r03_precision_ladder.txt: 1.09e-3 -> 1.22e-3 on the benchmarked step).  That each reduced-term kernel
    computes EXACTLY the gradient of the rounded operands is pinned separately (tests/test_gpu_ops.py::test_one_term_...,
    test_two_term_...)."""
    return x321_tol if precision in REDUCED else ptol(precision, fp32_tol, x3_tol)


def otol(precision, ks, act, cout, tol):
    """Forward bar of a chain whose OUTPUT layer is (ks, act, cout): in the "bf16x321o" mode an un-gated 5x5 output layer of
    seven-tile cout blocks multiplies x_hi x W_hi (one bf16 MFMA: both operands rounded to 8 bits, 2^-9 each) -- on the i.i.d.
    test operands 2-3e-3 of the tensor's max; exactness against fp64 on the ROUNDED operands is pinned at 2e-5 by
    tests/test_gpu_ops.py::test_one_term_output_layer_forward_*."""
    tiles = (cout + 15) // 16
    nt = min((7, 4, 2, 1), key=lambda t: (-(-tiles // t)) * (t + 2))          # x_pick_nt of csrc/conv_bf16x3.hip
    granted = ks == 5 and act == "linear" and nt == 7
    # ("bf16x321h": fp16 on both operands, 2^-12 each: 3e-4 on these operands; exactness: test_fp16_output_layer_forward_...)
    return 6e-3 if (precision == "bf16x321o" and granted) else 1e-3 if (precision == "bf16x321h" and granted) else tol


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 in fp64: the flip-robust gradient metric.  A ReLU unit whose pre-activation is within
    rounding of zero may land on either side in two correct implementations; it moves a handful of entries of an
    upstream weight gradient by a visible amount (so max|a-b|/max|b| jumps) but changes the tensor's L2 distance by
    ~1/sqrt(units) only."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def cosine(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return (torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300)).item()


def assert_grad_close(got, want, what="", l2=1e-3, cos=1e-6):
    """Gradient parity without a fallback: relative L2 <= l2 AND 1 - cosine <= cos, per tensor."""
    assert tuple(got.shape) == tuple(want.shape), (what, got.shape, want.shape)
    e, c = rel_l2(got, want), cosine(got, want)
    assert e <= l2 and 1.0 - c <= cos, "%s: rel L2 %.3e (<= %.1e), 1-cos %.3e (<= %.1e)" % (what, e, l2, 1.0 - c, cos)
    return e


class FlipCounter:
    """Counts ReLU / LeakyReLU sign disagreements between the HIP path and the oracle.

    The gradient of a ReLU network is discontinuous where a pre-activation crosses zero: two correct
    fp32 implementations that round differently can put a unit with |pre-activation| ~ 1e-6 on opposite
    sides, and that single unit moves every upstream weight gradient by ~1e-3 relative (measured:
    scripts/diag_chain.py -- 2e-6 with no flip, 1e-3..6e-3 with one).  Parity tests therefore hold
    gradients to the tight tolerance when no unit flipped and to `loose` otherwise.  `loose` is wide
    (1e-1) because the golden / unit-test networks are tiny (4..24 channels, 20x20 images), where one
    unit carries a visible share of a gradient; the split-bf16 arithmetic (1e-5 per layer instead of
    1e-6) flips ~10x more units than the fp32 MFMA path, which is the mode that pins gradients tightly."""

    def __enter__(self):
        from oracle import modules as om
        from wcmc_amd import ops
        self.om, self.ops = om, ops
        om.DEBUG_ACTS, ops.DEBUG_ACTS = [], []
        return self

    def __exit__(self, *exc):
        self.oracle_acts, self.hip_acts = self.om.DEBUG_ACTS, self.ops.DEBUG_ACTS
        self.om.DEBUG_ACTS, self.ops.DEBUG_ACTS = None, None
        return False

    def flips(self):
        """Activations are appended in call order, and the product runs the specular half before the diffuse one
        (it goes onto the forked stream first) while the oracle runs diffuse first: pair each oracle activation with
        the not-yet-used product activation of the same shape that disagrees least (the wrong partner disagrees on
        about half of its units)."""
        assert len(self.oracle_acts) == len(self.hip_acts), (len(self.oracle_acts), len(self.hip_acts))
        hip = [(tuple(b.shape), (b.detach() > 0).cpu()) for b in self.hip_acts]
        used, n = set(), 0
        for a in self.oracle_acts:
            pa = a > 0
            best, best_i = None, None
            for i, (shape, pb) in enumerate(hip):
                if i in used or shape != tuple(a.shape):
                    continue
                d = int((pa != pb).sum())
                if best is None or d < best:
                    best, best_i = d, i
                if d == 0:
                    break
            assert best_i is not None, "no product activation of shape %s" % (tuple(a.shape),)
            used.add(best_i)
            n += best
        return n

    def tol(self, tight, loose=1e-1):
        return tight if self.flips() == 0 else loose

    def check(self, got, want, tight, what="", l2=2e-2):
        """Gradient parity: the relative L2 bar `l2` always (no fallback); and, when no unit flipped, the max-norm
        bound `tight` on top (two implementations that made the same gate decisions agree entry by entry)."""
        if not hasattr(self, "_n"):
            self._n = self.flips()
        e = rel_l2(got, want)
        assert e <= l2, "%s: relative L2 %.3e > %.1e (%d flips)" % (what, e, l2, self._n)
        if self._n == 0:
            a, b = got.detach().double().cpu(), want.detach().double().cpu()
            m = ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
            assert m <= tight, "%s: max-norm %.3e > %.1e with no flipped unit" % (what, m, tight)


