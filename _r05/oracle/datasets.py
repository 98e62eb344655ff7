"""CPU restatement of the reference's per-image preprocessing (TEST INFRASTRUCTURE: only tests/ may import it).

``support/datasets.py`` of the reference:
  * ``DenoiseDataset._gradients``         :286-300
  * ``DenoiseDataset._preprocess_llpm``   :302-361
  * ``DenoiseDataset._preprocess_kpcn``   :487-582
  * raw channel map (``idx_nsy``, ``idx_g``, ``idx_sbmc``, ``idx_llpm``) :223-267, ``MAX_DEPTH = 5``

Pinned by ``tests/golden/preprocess.npz`` (outputs of the real methods, ``tests/golden/make_golden.py`` G6).
numpy fp32 throughout, like the reference.
"""
import numpy as np

MAX_DEPTH = 5


def channel_map(max_depth=MAX_DEPTH):
    """Raw (h, w, s, 38 + 11*(max_depth+1)) channel ranges used by the KPCN / LLPM preprocessors."""
    d = max_depth + 1
    return {
        "radiance": (2, 5), "diffuse": (5, 8),                                      # idx_nsy :229-232
        "bounce_types": (24 + d * 6, 24 + d * 7),                                   # idx_sbmc :249-254
        "albedo_at_diff": (24 + d * 7, 27 + d * 7), "normal_at_diff": (27 + d * 7, 30 + d * 7),
        "depth_at_diff": (30 + d * 7, 31 + d * 7),                                  # idx_g :242-247
        "path_weight": (31 + d * 7, 32 + d * 7), "radiance_wo_weight": (32 + d * 7, 35 + d * 7),
        "light_intensity": (35 + d * 7, 38 + d * 7), "throughputs": (38 + d * 7, 38 + d * 10),
        "roughnesses": (38 + d * 10, 38 + d * 11),                                  # idx_llpm :255-266
    }


def gradients(buf):
    """(h, w, c) -> (h, w, 2c): backward differences, zero first column / row (:286-300)."""
    dx = buf[:, 1:, ...] - buf[:, :-1, ...]
    dy = buf[1:, ...] - buf[:-1, ...]
    dx = np.pad(dx, [[0, 0], [1, 0], [0, 0]], mode="constant")
    dy = np.pad(dy, [[1, 0], [0, 0], [0, 0]], mode="constant")
    return np.concatenate([dx, dy], 2)


def preprocess_llpm(sample, max_depth=MAX_DEPTH):
    """(h, w, s, C) raw -> (h, w, s, 37) path descriptors (:302-361)."""
    m = channel_map(max_depth)
    g = lambda k: sample[..., m[k][0]:m[k][1]]
    feats = [np.log(g("path_weight") + 1e-6) / 90.0, np.log(g("radiance_wo_weight") + 1e-6) / 30.0,
             np.log(g("light_intensity") + 1e-8) / 10.0, np.log(g("throughputs") + 1e-6) / 30.0,
             g("bounce_types") / 19.0, np.sqrt(g("roughnesses"))]
    return np.concatenate(feats, axis=3)


def preprocess_kpcn(sample, max_depth=MAX_DEPTH):
    """(h, w, s, C) raw -> (h, w, 44) KPCN image-space features (:487-582)."""
    m = channel_map(max_depth)
    g = lambda k: sample[..., m[k][0]:m[k][1]]
    spp = sample.shape[2]
    eps = 0.00316
    normal = g("normal_at_diff").mean(2)
    normal_v = g("normal_at_diff").var(2).mean(2, keepdims=True) / spp
    depth = g("depth_at_diff").mean(2)
    depth_v = g("depth_at_diff").var(2)
    max_d = depth.max()
    if max_d > 0:
        depth = depth / max_d
        depth_v = depth_v / (max_d * max_d * spp)
    depth = np.clip(depth, 0, 1)
    albedo = g("albedo_at_diff").mean(2)
    albedo_v = g("albedo_at_diff").var(2).mean(2, keepdims=True) / spp
    albedo_sqr = ((albedo + eps) * (albedo + eps)).mean(2, keepdims=True)
    diff_sample = g("diffuse")
    diffuse = np.maximum(diff_sample, 0).mean(2)
    diffuse_v = np.maximum(diff_sample, 0).var(2).mean(2, keepdims=True) / spp
    spec_sample = np.maximum(g("radiance"), 0) - np.maximum(diff_sample, 0)
    specular = np.maximum(spec_sample, 0).mean(2)
    specular_v = np.maximum(spec_sample, 0).var(2).mean(2, keepdims=True) / spp
    specular_sqr = ((1 + specular) * (1 + specular)).mean(2, keepdims=True)
    diffuse = diffuse / (albedo + eps)
    diffuse_v = diffuse_v / albedo_sqr
    specular_v = specular_v / specular_sqr
    specular = np.log(1 + specular)
    feats = [diffuse, diffuse_v, gradients(diffuse), specular, specular_v, gradients(specular),
             normal, normal_v, gradients(normal), depth, depth_v, gradients(depth),
             albedo, albedo_v, gradients(albedo)]
    return np.concatenate(feats, axis=2)


def sample_patch_origins(prob, n):
    """``_sample_patches`` (datasets.py:795-810): n flat indices drawn with np.random.choice over the probability
    map (uniform when the map is not a distribution); returns (row, column) = (idx // w, idx % w)."""
    h, w = prob.shape
    try:
        roi = np.random.choice(h * w, size=n, p=prob.reshape(h * w))
    except ValueError:
        roi = np.random.choice(h * w, size=n)
    return np.stack([roi // w, roi % w], axis=1)


def assemble_kpcn_patch(kpcn, llpm, gt, origin, patch):
    """One item of ``DenoiseDataset.__getitem__`` for the KPCN base model (datasets.py:1076-1126), cropped at
    ``origin`` (:811-838) and transposed channel-first (:760-791).  kpcn (H,W,44), llpm (H,W,S,37) or None, gt (H,W,9)."""
    s = {}
    s["kpcn_diffuse_in"] = np.concatenate([kpcn[..., :10], kpcn[..., 20:]], axis=2)
    s["kpcn_specular_in"] = kpcn[..., 10:]
    s["kpcn_diffuse_buffer"] = kpcn[..., :3]
    s["kpcn_specular_buffer"] = kpcn[..., 10:13]
    s["kpcn_albedo"] = kpcn[..., 34:37] + 0.00316
    if llpm is not None:
        pw = llpm[..., :1].mean(2)
        s["kpcn_diffuse_in"] = np.concatenate((s["kpcn_diffuse_in"], pw), axis=2)
        s["kpcn_specular_in"] = np.concatenate((s["kpcn_specular_in"], pw), axis=2)
        s["paths"] = np.array(llpm[..., 1:])
    total, diffuse, albedo = gt[:, :, 0:3], gt[:, :, 3:6], gt[:, :, 6:]
    s["target_diffuse"] = diffuse / (albedo + 0.00316)
    s["target_specular"] = np.log(1 + total - diffuse)
    s["target_total"] = total
    x, y = int(origin[0]), int(origin[1])
    out = {}
    for k, v in s.items():
        v = v[x:x + patch, y:y + patch, ...]
        out[k] = np.transpose(v, (2, 0, 1)) if v.ndim == 3 else np.transpose(v, (2, 3, 0, 1))
    return out
