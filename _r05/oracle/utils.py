"""Oracle restatement of ``support/utils.py:24-42`` (``crop_like``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pinned by golden G1
(``tests/golden/crop_like.npz``, generated from the real reference).
"""


def crop_like(src, tgt):
    """Center-crop the last two dims of ``src`` to those of ``tgt``.

    ``crop = max(delta // 2, 0)`` at the start, ``delta - crop`` at the end (an odd
    delta crops one more at the end); a non-positive delta is a no-op and the
    result is a view (``support/utils.py:31-42``).
    """
    dh = src.shape[-2] - tgt.shape[-2]
    dw = src.shape[-1] - tgt.shape[-1]
    c0h, c0w = max(dh // 2, 0), max(dw // 2, 0)
    c1h, c1w = dh - c0h, dw - c0w
    if c0h > 0 or c0w > 0 or c1h > 0 or c1w > 0:
        return src[..., c0h:src.shape[-2] - c1h, c0w:src.shape[-1] - c1w]
    return src
