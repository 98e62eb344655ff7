"""``support/utils.py:24-42`` of the reference: ``crop_like`` (shape arithmetic only)."""


def crop_like(src, tgt):
    """Center-crop the last two dims of ``src`` to those of ``tgt`` and return a view.

    ``crop = max(delta // 2, 0)`` at the start and ``delta - crop`` at the end, so an odd
    delta loses one more element at the end; a non-positive delta returns ``src``.
    """
    dh = src.shape[-2] - tgt.shape[-2]
    dw = src.shape[-1] - tgt.shape[-1]
    top, left = max(dh // 2, 0), max(dw // 2, 0)
    bottom, right = dh - top, dw - left
    if top > 0 or left > 0 or bottom > 0 or right > 0:
        return src[..., top:src.shape[-2] - bottom, left:src.shape[-1] - right]
    return src
