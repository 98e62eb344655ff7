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


def test_patch_batcher_against_the_reference_dataset_items(golden_dir):
    """PatchBatcher (the loader step on the device) == the real DenoiseDataset.__getitem__ items:
    same origins from the same numpy seed, copies bit-exact, the two target transforms and the albedo offset to 1 ulp."""
    import os
    from wcmc_amd.support.datasets import PatchBatcher
    d = np.load(os.path.join(golden_dir, "patches.npz"))
    P = int(d["patch"])
    dev = "cuda"
    kpcn, llpm, gt = (torch.from_numpy(d[k]).to(dev) for k in ("kpcn", "llpm", "gt"))
    for tag in ("llpm", "vanilla"):
        pb = PatchBatcher(patch_size=P, batch_size=8)
        assert pb.patches_per_image == int(d[tag + "/patches_per_image"])
        np.random.seed(int(d["seed"]))
        origins = pb.sample_origins(d["prob"])
        n = len([k for k in d.files if k.startswith(tag + "/") and k.endswith("/target_total")])
        batch = pb.batch(kpcn, llpm if tag == "llpm" else None, gt, origins[:n])
        assert set(batch) == {k.split("/")[-1] for k in d.files if k.startswith(tag + "/0/")}
        for i in range(n):
            for k, v in batch.items():
                want = d["%s/%d/%s" % (tag, i, k)]
                got = v[i].cpu().numpy()
                if k in ("target_diffuse", "target_specular", "kpcn_albedo", "kpcn_diffuse_in", "kpcn_specular_in"):
                    np.testing.assert_allclose(got, want, rtol=3e-7, atol=1e-7, err_msg="%s %d %s" % (tag, i, k))
                else:
                    np.testing.assert_array_equal(got, want, err_msg="%s %d %s" % (tag, i, k))
    with pytest.raises(ValueError):
        PatchBatcher(patch_size=P).batch(kpcn, llpm, gt, np.array([[d["kpcn"].shape[0] - P + 1, 0]]))
    # full size: one 128-pixel, 8-spp batch of 8 from a 512 x 512 image keeps the interface's contract
    H = 256
    g = torch.Generator().manual_seed(5)
    kp, ll, gg = torch.rand(H, H, 44, generator=g).to(dev), torch.rand(H, H, 8, 37, generator=g).to(dev), torch.rand(H, H, 9, generator=g).to(dev) + 1
    big = PatchBatcher().batch(kp, ll, gg, np.array([[0, 0], [128, 128], [17, 100]] + [[64, 3]] * 5))
    assert big["paths"].shape == (8, 8, 36, 128, 128) and big["kpcn_diffuse_in"].shape == (8, 35, 128, 128)
    assert torch.equal(big["paths"][2, 5, 7], ll[17:145, 100:228, 5, 8]) and torch.equal(big["target_total"][1, 2], gg[128:, 128:, 2])


def test_patch_loader_stages_images_and_equals_the_direct_path():
    """SURVEY.md 8f rank 3, loader half: ``PatchLoader`` (background reader thread -> pinned ring -> copy stream -> device
    preprocessing -> importance-sampled patch batches) yields exactly what the direct, unstaged calls produce for the same
    images and the same numpy seed; a reader error surfaces in the consumer."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as mg
    from wcmc_amd.support.datasets import DenoisePreprocessor, PatchBatcher
    from wcmc_amd.support.loader import PatchLoader
    H, W, S, P, B = 72, 88, 4, 32, 4
    rng = np.random.RandomState(5)
    images = []
    for i in range(3):
        prob = rng.rand(H, W)
        prob[H - P + 1:, :] = 0
        prob[:, W - P + 1:] = 0
        images.append({"raw": mg.raw_samples(H, W, S, 300 + i), "gt": rng.rand(H, W, 9).astype(np.float32),
                       "prob": prob / prob.sum()})
    calls = []

    def reader(i):
        calls.append(i)
        return images[i]

    loader = PatchLoader(reader, range(3), DEV, batch_size=B, patch_size=P, patches_per_image=8)
    assert len(loader) == 6
    np.random.seed(11)
    got = [{k: v.clone() for k, v in b.items()} for b in loader]
    assert len(got) == 6 and calls == [0, 1, 2]
    assert got[0]["paths"].shape == (B, S, 36, P, P) and got[0]["kpcn_diffuse_in"].shape == (B, 35, P, P)
    assert loader.stager.bytes_moved == sum(im["raw"].nbytes + im["gt"].nbytes for im in images)
    pre, bat = DenoisePreprocessor(), PatchBatcher(P, B)
    bat.patches_per_image = 8
    np.random.seed(11)
    k = 0
    for im in images:
        raw = torch.from_numpy(im["raw"]).to(DEV)
        kp, ll, gt = pre._preprocess_kpcn(raw), pre._preprocess_llpm(raw), torch.from_numpy(im["gt"]).to(DEV)
        origins = bat.sample_origins(im["prob"])
        for o in range(0, 8, B):
            want = bat.batch(kp, ll, gt, origins[o:o + B])
            assert want.keys() == got[k].keys()
            # one allocation per batch (its entries are views, each on a 256-byte boundary): a consumer on another stream keeps it
            # alive with one record_stream and frees one block
            assert len({v.untyped_storage().data_ptr() for v in want.values()}) == 1
            assert all(v.is_contiguous() and v.data_ptr() % 256 == 0 for v in want.values())
            for name in want:
                assert torch.equal(want[name], got[k][name]), (k, name)
            k += 1

    def bad_reader(i):
        if i == 1:
            raise OSError("image 1 is unreadable")
        return images[i]

    with pytest.raises(OSError, match="unreadable"):
        for _ in PatchLoader(bad_reader, range(3), DEV, batch_size=B, patch_size=P, patches_per_image=8):
            pass
