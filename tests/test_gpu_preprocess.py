"""GPU parity of the data-step kernels against the reference's own outputs (G6) and,
at a realistic size, against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pre():
    from wcmc_amd.support.datasets import DenoisePreprocessor
    return DenoisePreprocessor()


def _close(got, want, rtol, atol, what):
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=rtol, atol=atol, err_msg=what)


@pytest.mark.parametrize("name", ["a", "b", "zero_depth"])
def test_preprocess_against_reference_golden(golden_dir, name):
    d = np.load(os.path.join(golden_dir, "preprocess.npz"))
    raw = torch.from_numpy(d[name + "/raw"]).to(DEV)
    pre = _pre()
    # log / sqrt / division differ from numpy's libm by an ulp or two; variances are sums of squares near zero
    _close(pre._preprocess_llpm(raw), d[name + "/llpm"], 2e-6, 1e-7, "llpm " + name)
    _close(pre._preprocess_kpcn(raw), d[name + "/kpcn"], 2e-5, 1e-6, "kpcn " + name)


def test_gradients_against_reference_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "preprocess.npz"))
    got = _pre()._gradients(torch.from_numpy(d["grad/buf"]).to(DEV))
    np.testing.assert_array_equal(got.cpu().numpy(), d["grad/out"])          # subtractions only: bit-exact


def test_preprocess_full_patch_against_oracle():
    """One 128x128 patch at 8 spp (the benchmark's per-patch raw size, 54.5 MB)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as mg
    from oracle import datasets as od
    raw = mg.raw_samples(128, 128, 8, 77)
    pre = _pre()
    x = torch.from_numpy(raw).to(DEV)
    _close(pre._preprocess_llpm(x), od.preprocess_llpm(raw), 2e-6, 1e-7, "llpm 128")
    _close(pre._preprocess_kpcn(x), od.preprocess_kpcn(raw), 5e-5, 2e-6, "kpcn 128")
    kp = pre._preprocess_kpcn(x)
    assert float(kp[..., 30].max()) <= 1.0 and float(kp[..., 30].min()) >= 0.0          # normalised, clipped depth
    assert torch.equal(kp[:, 0, 4:7], torch.zeros_like(kp[:, 0, 4:7]))                  # zero first column of d/dx


@pytest.mark.parametrize("spp", [1, 3, 6, 16])
def test_preprocess_kpcn_other_sample_counts(spp):
    """spp 3 and 6 take the one-lane-per-pixel statistics kernel, 1 and 16 the lane-per-sample one."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as mg
    from oracle import datasets as od
    raw = mg.raw_samples(21, 19, spp, 80 + spp)
    got = _pre()._preprocess_kpcn(torch.from_numpy(raw).to(DEV))
    _close(got, od.preprocess_kpcn(raw), 5e-5, 2e-6, "kpcn spp %d" % spp)


def test_preprocess_rejects_host_tensors():
    with pytest.raises(RuntimeError, match="no CPU path"):
        _pre()._preprocess_llpm(torch.zeros(2, 2, 2, 104))
