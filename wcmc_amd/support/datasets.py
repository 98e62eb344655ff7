"""GPU counterparts of the per-image preprocessing methods of the reference's ``support.datasets.DenoiseDataset``
(``datasets.py:286-361,487-582``): same names, same layouts, torch CUDA tensors instead of numpy arrays.

The reference runs these in numpy on the loader's CPU worker (``_offline_preprocess`` :584-660 and the online
path of ``__getitem__``); at 24.6 MB of raw samples per 128x128 patch the loader, not the GPUs, bounds a real
training run.  Only the arithmetic is provided here; file handling, patch sampling and
the batch dictionary stay with the caller.
"""
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
