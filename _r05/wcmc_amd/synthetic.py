"""Synthetic KPCN-Manifold batches with the reference's schema.

Keys, shapes and value ranges follow what ``support/datasets.py`` hands to ``KPCNInterface``
(``datasets.py:1080-1126``; preprocessing ``:301-361,487-582``): 34 image-space channels
(+1 mean path weight with the manifold buffers), albedo-factored diffuse / log-specular noisy
buffers, ground-truth targets, and S per-sample 36-channel path descriptors whose bounces
beyond the sampled path length sit at the log floor (sparse descriptors).
"""
import math

import torch
import torch.nn.functional as F


def _blur(x, k=5):
    """k x k box blur with replicate padding (avg_pool2d: keeps MIOpen out of the data generator)."""
    return F.avg_pool2d(F.pad(x, (k // 2,) * 4, mode="replicate"), k, 1)


def _grads(x):
    """d/dx, d/dy finite differences with a zero first column / row (datasets.py:286-299)."""
    dx = torch.zeros_like(x)
    dy = torch.zeros_like(x)
    dx[..., :, 1:] = x[..., :, 1:] - x[..., :, :-1]
    dy[..., 1:, :] = x[..., 1:, :] - x[..., :-1, :]
    return dx, dy


def make_batch(b, s, h, seed=0, device="cpu", use_llpm=True):
    """One batch dict of fp32 tensors on ``device`` (keys of SURVEY.md Appendix B)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    rand = lambda *sh: torch.rand(*sh, generator=g, device=dev)
    randn = lambda *sh: torch.randn(*sh, generator=g, device=dev)
    eps = 0.00316
    albedo_gt = _blur(rand(b, 3, h, h))
    albedo = (albedo_gt + 0.05 * randn(b, 3, h, h)).clamp(0, 1)
    rad_d_gt = _blur(torch.exp(randn(b, 3, h, h) - 1.0)) * albedo_gt
    rad_s_gt = _blur(torch.exp(randn(b, 3, h, h) - 2.0))
    noise = lambda: torch.exp(0.5 * randn(b, 3, h, h))
    diffuse = rad_d_gt * noise() / (albedo + eps)                 # albedo-factored (datasets.py:546)
    specular = torch.log1p(rad_s_gt * noise())                    # log(1+x)       (datasets.py:550)
    var = lambda: (0.1 * randn(b, 1, h, h)).pow(2) / s
    normals = F.normalize(_blur(randn(b, 3, h, h)), dim=1)
    depth = _blur(rand(b, 1, h, h))

    def ten(x):                                                   # value(3) var(1) dx(3) dy(3)
        dx, dy = _grads(x)
        return torch.cat([x, var(), dx, dy], 1)

    nd, ndx, ndy = normals, *_grads(normals)
    gbuf = torch.cat([torch.cat([nd, var(), ndx, ndy], 1),                        # normals 10
                      torch.cat([depth, var(), *_grads(depth)], 1),               # depth 4
                      ten(albedo)], 1)                                            # albedo 10
    batch = {
        "kpcn_diffuse_in": torch.cat([ten(diffuse), gbuf], 1),
        "kpcn_specular_in": torch.cat([ten(specular), gbuf], 1),
        "kpcn_diffuse_buffer": diffuse,
        "kpcn_specular_buffer": specular,
        "kpcn_albedo": albedo + eps,
        "target_diffuse": rad_d_gt / (albedo_gt + eps),
        "target_specular": torch.log1p(rad_s_gt),
        "target_total": rad_d_gt + rad_s_gt,
    }
    if use_llpm:
        pw = torch.log(rand(b, 1, h, h).clamp_min(1e-30) + 1e-6) / 90.0           # datasets.py:319
        batch["kpcn_diffuse_in"] = torch.cat([batch["kpcn_diffuse_in"], pw], 1)
        batch["kpcn_specular_in"] = torch.cat([batch["kpcn_specular_in"], pw], 1)
        length = torch.randint(1, 7, (b, s, 1, h, h), generator=g, device=dev)    # path length 1..6
        bounce = torch.arange(6, device=dev).view(1, 1, 6, 1, 1)
        alive = (bounce < length).float()                                         # (b,s,6,h,h)
        alive3 = alive.repeat_interleave(3, 2)
        p = torch.empty(b, s, 36, h, h, device=dev)
        p[:, :, 0:3] = torch.log(rand(b, s, 3, h, h) + 1e-6) / 30.0
        p[:, :, 3:6] = torch.log(rand(b, s, 3, h, h) + 1e-8) / 10.0
        p[:, :, 6:24] = torch.log(rand(b, s, 18, h, h) * alive3 + 1e-6) / 30.0    # floor -0.4605 past the end
        p[:, :, 24:30] = torch.randint(0, 20, (b, s, 6, h, h), generator=g, device=dev).float() / 19.0 * alive
        p[:, :, 30:36] = torch.sqrt(rand(b, s, 6, h, h)) * alive
        batch["paths"] = p
    return {k: v.contiguous().float() for k, v in batch.items()}


LOG_FLOOR = math.log(1e-6) / 30.0
