"""TEST INFRASTRUCTURE ONLY -- CPU restatement of ``cv.resize(img, (W, H))`` as the reference's reader calls it
(reader.py:79: default interpolation INTER_LINEAR, uint8, one channel).

PARITY UNPINNED: OpenCV (``opencv-python``, no version pinned in the reference's requirements.txt) is absent from the
reference tree and from this image, and the reference holds no resized fixture.  The restatement follows the library's
published algorithm (modules/imgproc/src/resize.cpp, 4.x):

* ``hal::resize``: ``scale = 1 / (dsize / ssize)`` in double; a reduction by exactly 2 x 2 with INTER_LINEAR is
  switched to INTER_AREA, whose 2 x 2 fast path is ``(s00 + s01 + s10 + s11 + 2) >> 2``;
* otherwise, per output column ``fx = (float)((dx + 0.5) * scale_x - 0.5)``, ``sx = floor(fx)``, ``fx -= sx``;
  ``sx < 0 -> (0, fx = 0)``; ``sx >= sw - 1 -> (sw - 1, fx = 0)``; weights ``saturate_cast<short>((1 - fx) * 2048)``,
  ``saturate_cast<short>(fx * 2048)`` (float product, round half to even).  Rows the same, except that the weights of a
  row are left alone and the two source rows are clamped into the image when they are read;
* ``HResizeLinear``: ``S = s[sx] * a0 + s[sx + 1] * a1`` (32-bit integers);
* ``VResizeLinear`` (uchar): ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``;
* equal sizes: a copy.
"""
import numpy as np


def _coef(c):
    v = np.rint((c * np.float32(2048.0)).astype(np.float64))
    return np.clip(v, -32768, 32767).astype(np.int64)


def resize_linear_u8(img, H, W):
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    sh, sw = img.shape
    if (sh, sw) == (H, W):
        return img.copy()
    scale_x, scale_y = 1.0 / (np.float64(W) / sw), 1.0 / (np.float64(H) / sh)
    eps = np.finfo(np.float64).eps
    if abs(scale_x - 2.0) < eps and abs(scale_y - 2.0) < eps:
        s = img.astype(np.int64)
        return ((s[0:2 * H:2, 0:2 * W:2] + s[0:2 * H:2, 1:2 * W:2] + s[1:2 * H:2, 0:2 * W:2] + s[1:2 * H:2, 1:2 * W:2] + 2) >> 2
                ).astype(np.uint8)
    fx = ((np.arange(W, dtype=np.float64) + 0.5) * scale_x - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = fx - sx.astype(np.float32)
    lo, hi = sx < 0, None
    fx[lo] = 0; sx[lo] = 0
    hi = sx >= sw - 1
    fx[hi] = 0; sx[hi] = sw - 1
    a0, a1 = _coef(np.float32(1.0) - fx), _coef(fx)
    fy = ((np.arange(H, dtype=np.float64) + 0.5) * scale_y - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int64)
    fy = fy - sy.astype(np.float32)
    b0, b1 = _coef(np.float32(1.0) - fy), _coef(fy)
    s = img.astype(np.int64)
    x1 = np.minimum(sx + 1, sw - 1)
    rows = s[:, sx] * a0[None, :] + s[:, x1] * a1[None, :]          # horizontal pass of every source row
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    S0, S1 = rows[y0], rows[y1]
    out = (((b0[:, None] * (S0 >> 4)) >> 16) + ((b1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)
