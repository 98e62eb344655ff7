"""Tiled full-image inference: the step after the path.

Mirrors, on device tensors, what the reference does around ``validate_batch``:
  * ``FullImageDataset`` tiling  ``support/datasets.py:1276-1299``  overlapping ``patch_size`` tiles at stride
    ``patch_size - 2*pad_size``; each tile owns its interior, tiles on the image border own their border too;
  * ``inference``                ``test_models.py:49-101``          replicate-pad the network output back to the
    tile size, paste each tile's owned window into the full image (radiance and P-buffers);
  * valid crop + has-hit composite ``test_models.py:217-232``       drop the outer (128 - 72) / 2 pixels and keep
    the noisy input where no surface was hit.
``test_models.py`` is not importable as shipped (``from train_kpcn import weights_init`` does not exist there), so
these are restatements checked by known-answer tests, not by a golden (parity unpinned).
"""
import torch
import torch.nn.functional as F


def tile_coords(h, w, patch_size=128, pad_size=32):
    """[(i_start, j_start, i_end, j_end, i, j)] as ``FullImageDataset`` builds them (datasets.py:1276-1294)."""
    stride = patch_size - 2 * pad_size
    assert (h - 2 * pad_size) % stride == 0 and (w - 2 * pad_size) % stride == 0
    coords = []
    for i in range(0, h - 2 * pad_size, stride):
        for j in range(0, w - 2 * pad_size, stride):
            i_start, j_start = i + pad_size, j + pad_size
            i_end, j_end = i + patch_size - pad_size, j + patch_size - pad_size
            if i == 0:
                i_start = 0
            if j == 0:
                j_start = 0
            if i == h - patch_size:
                i_end = i + patch_size
            if j == w - patch_size:
                j_end = j + patch_size
            coords.append((i_start, j_start, i_end, j_end, i, j))
    return coords


def inference(interface, dataloader, h, w, patch_size=128, use_llpm_buf=True, device=None):
    """``test_models.inference``: ``dataloader`` yields ``(batch, i_start, j_start, i_end, j_end, i, j)`` with a
    dict of (B, ...) tensors and per-item integer sequences.  Returns ``(out_rad (3,H,W), out_path)`` on the device
    (the reference additionally moves them to numpy HWC)."""
    interface.to_eval_mode()
    out_rad, out_path = None, None
    with torch.no_grad():
        for batch, i_start, j_start, i_end, j_end, i, j in dataloader:
            for k in batch:
                if isinstance(batch[k], torch.Tensor) and device is not None:
                    batch[k] = batch[k].to(device)
            out, p_buffers = interface.validate_batch(batch)
            if out_rad is None:
                out_rad = torch.zeros((3, h, w), device=out.device)
            pad_h, pad_w = patch_size - out.shape[2], patch_size - out.shape[3]
            if pad_h != 0 and pad_w != 0:
                out = F.pad(out, (pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2), 'replicate')
            if use_llpm_buf and out_path is None and p_buffers is not None:
                if isinstance(p_buffers, dict):
                    out_path = {key: torch.zeros((v.shape[1], v.shape[2], h, w), device=v.device)
                                for key, v in p_buffers.items()}
                else:
                    out_path = torch.zeros((p_buffers.shape[1], p_buffers.shape[2], h, w), device=p_buffers.device)
            for b in range(out.shape[0]):
                i0, i1, j0, j1, ib, jb = (int(i_start[b]), int(i_end[b]), int(j_start[b]), int(j_end[b]),
                                          int(i[b]), int(j[b]))
                out_rad[:, i0:i1, j0:j1] = out[b, :, i0 - ib:i1 - ib, j0 - jb:j1 - jb]
                if use_llpm_buf and out_path is not None:
                    if isinstance(p_buffers, dict):
                        for key in p_buffers:
                            out_path[key][:, :, i0:i1, j0:j1] = p_buffers[key][b, :, :, i0 - ib:i1 - ib, j0 - jb:j1 - jb]
                    else:
                        out_path[:, :, i0:i1, j0:j1] = p_buffers[b, :, :, i0 - ib:i1 - ib, j0 - jb:j1 - jb]
    return out_rad, out_path


def crop_and_composite(out_rad, noisy_input, has_hit, patch_size=128, valid_size=72):
    """``test_models.py:217-232`` on (H, W, 3) tensors: valid-core crop, then the noisy input wherever no surface
    was hit (background and emitters are not denoised)."""
    crop = (patch_size - valid_size) // 2
    out_rad = out_rad[crop:-crop, crop:-crop, ...]
    noisy_input = noisy_input[crop:-crop, crop:-crop, ...]
    has_hit = has_hit[crop:-crop, crop:-crop, ...]
    return torch.where(has_hit == 0, noisy_input, out_rad)
