"""GPU counterparts of the per-image preprocessing methods of the reference's ``support.datasets.DenoiseDataset``
(``datasets.py:286-361,487-582``): same names, same layouts, torch CUDA tensors instead of numpy arrays.

The reference runs these in numpy on the loader's CPU worker (``_offline_preprocess`` :584-660 and the online
path of ``__getitem__``); at 24.6 MB of raw samples per 128x128 patch the loader, not the GPUs, bounds a real
training run.  ``PatchBatcher`` is the device-side counterpart of what ``__getitem__`` does per
patch (importance sampling of patch origins, cropping, channel selection, target transforms, channel-first layout,
``datasets.py:795-840,1026-1146``); file handling stays with the caller.
"""
import numpy as np
import torch

from .. import ops as _ops


class DenoisePreprocessor:
    MAX_DEPTH = 5                                   # datasets.py:68

    def __init__(self, max_depth=MAX_DEPTH):
        self.max_depth = max_depth

    def _gradients(self, buf):
        """(h, w, c) -> (h, w, 2c): horizontal and vertical backward differences (datasets.py:286-300)."""
        return _ops.gradients(buf)

    def _preprocess_llpm(self, sample):
        """raw (h, w, s, 104) -> (h, w, s, 37): path weight, radiance w/o weight, light intensity, throughputs,
        bounce types, roughnesses (datasets.py:302-361)."""
        return _ops.preprocess_llpm(sample, self.max_depth)

    def _preprocess_kpcn(self, sample):
        """raw (h, w, s, 104) -> (h, w, 44): diffuse / specular / normal / depth / albedo means, variances and
        gradients (datasets.py:487-582)."""
        return _ops.preprocess_kpcn(sample, self.max_depth)


class PatchBatcher:
    """KPCN-base-model batches straight from one image's device-resident buffers.

    ``sample_origins`` is the reference's importance sampling (``_sample_patches``, datasets.py:795-810): one
    ``np.random.choice`` over the flattened probability map (uniform if the map is not a distribution), so the same
    numpy seed yields the same patches; ``x = idx // w`` is the ROW and ``y = idx % w`` the column, as there.
    ``batch`` crops them and builds the dictionary ``KPCNInterface.preprocess`` asserts on (Appendix B of SURVEY.md) in
    one kernel launch (``ops.assemble_kpcn_patches``) -- 128 x 128 x 8 spp: 24.6 MB per patch that never visit the host.
    """
    PATCH_SIZE = 128                                # datasets.py:66

    def __init__(self, patch_size=PATCH_SIZE, batch_size=8):
        self.patch_size = patch_size
        self.patches_per_image = (256 // batch_size) * batch_size          # datasets.py:275

    def sample_origins(self, prob, n=None):
        h, w = prob.shape
        n = self.patches_per_image if n is None else n
        try:
            roi = np.random.choice(h * w, size=n, p=np.asarray(prob).reshape(h * w))
        except ValueError:
            roi = np.random.choice(h * w, size=n)
        return np.stack([roi // w, roi % w], axis=1).astype(np.int32)

    def check_origins(self, origins, h, w):
        """Windows must lie inside the image (the reference would silently return a smaller patch)."""
        o = origins.cpu().numpy() if isinstance(origins, torch.Tensor) else np.asarray(origins)
        if o.size and (int(o[:, 0].max()) + self.patch_size > h or int(o[:, 1].max()) + self.patch_size > w or int(o.min()) < 0):
            raise ValueError("PatchBatcher: a %d-pixel patch origin lies outside the %dx%d image" % (self.patch_size, h, w))

    def batch(self, kpcn, llpm, gt, origins, check=True):
        """kpcn (H,W,44), llpm (H,W,S,37) or None, gt (H,W,9): device tensors; origins: (B,2) rows/columns (numpy or
        tensor).  ``check=False``: the caller has run ``check_origins`` on them (``PatchLoader`` does, once per image, on the
        host copy -- checking a device tensor here would synchronise every batch)."""
        if check:
            self.check_origins(origins, *kpcn.shape[:2])
        o = torch.as_tensor(np.asarray(origins), dtype=torch.int32) if not isinstance(origins, torch.Tensor) else origins
        return _ops.assemble_kpcn_patches(kpcn, llpm, gt, o.to(kpcn.device, torch.int32).contiguous(), self.patch_size)
