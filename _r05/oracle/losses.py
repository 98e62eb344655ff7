"""Oracle restatement of ``support/losses.py``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pinned by goldens G2-G4
(``tests/golden/losses_*.npz``) generated from the real reference.

The reference draws its pairing permutations from the global CPU generator
inside ``forward`` (``losses.py:35,50``), in the order patch -> batch
(``losses.py:105-109``).  Here the permutations are explicit arguments of the
functional forms; the Module forms draw them with the same calls in the same
order so that ``torch.manual_seed(s)`` reproduces the reference bit for bit.
"""
import math

import torch
import torch.nn as nn


def tonemap_gamma(img):
    """``losses.py:63-65``: Reinhard then gamma 1/2.2 (as the literal 0.454545)."""
    img = torch.clamp(img, min=0)
    return (img / (1 + img)) ** 0.454545


def _rows(t):
    """(B,S,C,H,W) -> (B, S*H*W, C), row order (s,h,w) (``losses.py:37,41``)."""
    b, s, c, h, w = t.shape
    return t.permute(0, 1, 3, 4, 2).reshape(b, s * h * w, c)


def pair_displacement(p_rows, r_rows, idx):
    """d_i = 1/2 |P_i - P_pi(i)|^2 - 1/2 |R_i - R_pi(i)|^2 along dim -2."""
    dp = 0.5 * ((p_rows - p_rows.index_select(-2, idx)) ** 2).sum(-1)
    dr = 0.5 * ((r_rows - r_rows.index_select(-2, idx)) ** 2).sum(-1)
    return dp - dr


def feature_mse(p_buffer, ref, idx_patch, idx_batch=None):
    """``FeatureMSE.forward`` for color='rgb' (``losses.py:82-113``).

    ``idx_patch``: permutation of S*H*W shared by every batch element
    (``intra_patch_dist`` ``losses.py:33-46``).  ``idx_batch``: permutation of
    B*S*H*W (``intra_batch_dist`` ``losses.py:48-61``) or None for
    ``non_local=False`` where the patch term is counted twice (``losses.py:111``).
    """
    b, s, c, h, w = p_buffer.shape
    r = tonemap_gamma(ref).unsqueeze(1).expand(b, s, 3, h, w)
    if not torch.isfinite(p_buffer).all() or not torch.isfinite(r).all():
        raise RuntimeError("Infinite loss at train time.")
    p_rows, r_rows = _rows(p_buffer), _rows(r)
    d = pair_displacement(p_rows, r_rows, idx_patch)
    loss_p = 0.5 * (d ** 2).mean()
    if idx_batch is None:
        return loss_p + loss_p
    d = pair_displacement(p_rows.reshape(-1, c), r_rows.reshape(-1, 3), idx_batch)
    return loss_p + 0.5 * (d ** 2).mean()


class FeatureMSE(nn.Module):
    def __init__(self, color="rgb", non_local=True):
        super().__init__()
        assert color == "rgb"
        self.non_local = non_local
        self.last_perms = None

    def forward(self, p_buffer, ref):
        b, s, c, h, w = p_buffer.shape
        idx_patch = torch.randperm(s * h * w)
        idx_batch = torch.randperm(b * s * h * w) if self.non_local else None
        self.last_perms = (idx_patch, idx_batch)
        return feature_mse(p_buffer, ref, idx_patch, idx_batch)


def global_relative_similarity(p_buffer, ref, idx_patch, idx_batch, alpha=2):
    """``GlobalRelativeSimilarityLoss.forward`` (``losses.py:186-211``)."""
    if not torch.isfinite(p_buffer).all() or not torch.isfinite(ref).all():
        raise RuntimeError("Infinite loss at train time.")
    b, s, c, h, w = p_buffer.shape
    r = tonemap_gamma(ref).unsqueeze(1).expand(b, s, 3, h, w)
    p_rows, r_rows = _rows(p_buffer), _rows(r)
    disp_p = pair_displacement(p_rows, r_rows, idx_patch).reshape(-1)
    disp_b = pair_displacement(p_rows.reshape(-1, c), r_rows.reshape(-1, 3), idx_batch)
    zero = torch.zeros(1, dtype=p_buffer.dtype)
    e = alpha * torch.cat([disp_p, disp_b, -disp_p, -disp_b, zero])
    out = torch.logsumexp(e, dim=0) - math.log(1 + 4 * b * s * h * w)
    return out / math.sqrt(alpha)


class GlobalRelativeSimilarityLoss(nn.Module):
    def __init__(self, alpha=2, color="rgb"):
        super().__init__()
        self.alpha = alpha

    def forward(self, p_buffer, ref):
        b, s, c, h, w = p_buffer.shape
        idx_patch = torch.randperm(s * h * w)
        idx_batch = torch.randperm(b * s * h * w)
        return global_relative_similarity(p_buffer, ref, idx_patch, idx_batch, self.alpha)


def _reinhard(im):
    im = torch.clamp(im, min=0)
    return im / (1 + im)


class RelativeMSE(nn.Module):
    """``losses.py:245-264``: 0.5 * mean((im-ref)^2 / (ref^2 + eps))."""

    def __init__(self, eps=1e-2):
        super().__init__()
        self.eps = eps

    def forward(self, im, ref):
        return 0.5 * torch.mean((im - ref) ** 2 / (ref ** 2 + self.eps))


class SMAPE(nn.Module):
    """``losses.py:267-284``; the denominator carries no gradient."""

    def __init__(self, eps=1e-2):
        super().__init__()
        self.eps = eps

    def forward(self, im, ref):
        den = self.eps + im.detach().abs() + ref.detach().abs()
        return ((im - ref).abs() / den).mean()


class TonemappedMSE(nn.Module):
    """``losses.py:287-302``."""

    def __init__(self, eps=1e-2):
        super().__init__()
        self.eps = eps

    def forward(self, im, ref):
        return 0.5 * torch.mean((_reinhard(im) - _reinhard(ref)) ** 2)


class TonemappedRelativeMSE(nn.Module):
    """``losses.py:305-320``."""

    def __init__(self, eps=1e-2):
        super().__init__()
        self.eps = eps

    def forward(self, im, ref):
        im, ref = _reinhard(im), _reinhard(ref)
        return 0.5 * torch.mean((im - ref) ** 2 / (ref ** 2 + self.eps))
