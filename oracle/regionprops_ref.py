"""Oracle for the 32 region-property columns -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Restates ``calculate_regionprops`` (/root/reference/detection.py:75-159), i.e.
``skimage.measure.regionprops(mask, intensity_image=img)[0]`` mapped through
``regionprop2zooprocess``.  The scikit-image formulas followed are those of the 0.18.3 install in
the build container (``skimage/measure/_regionprops.py``, ``_regionprops_utils.py``,
``_moments.py``, ``morphology/convex_hull.py``; line numbers in SURVEY.md section 8(a) R9-table).
Pinned by tests/golden/regionprops_skimage.npz (made by tests/golden/make_golden_skimage.py
with that scikit-image).
"""
from math import atan2, pi, sqrt

import numpy as np
from scipy import ndimage as ndi

PROP_KEYS = [
    "width", "height", "bx", "by", "circ.", "area_exc", "area_rprops", "%area", "major", "minor",
    "centroid_y", "centroid_x", "convex_area", "min_intensity", "max_intensity", "mean_intensity",
    "intden", "perim.", "elongation", "range", "perimareaexc", "perimmajor", "circex", "angle",
    "bounding_box_area", "eccentricity", "equivalent_diameter", "euler_number", "extent",
    "local_centroid_col", "local_centroid_row", "solidity",
]

EULER_COEFS2D_8 = np.array([0, 0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0, -1, 0])


def perimeter4(M):
    """skimage.measure.perimeter(image, 4) (_regionprops_utils.py:186-249)."""
    img = M.astype(np.uint8)
    strel4 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    er = ndi.binary_erosion(img, strel4, border_value=0)
    border = img - er
    w = np.zeros(50)
    w[[5, 7, 15, 17, 25, 27]] = 1
    w[[21, 33]] = sqrt(2)
    w[[13, 23]] = (1 + sqrt(2)) / 2
    pim = ndi.convolve(border, np.array([[10, 2, 10], [2, 1, 2], [10, 2, 10]]), mode="constant", cval=0)
    h = np.bincount(pim.ravel(), minlength=50)
    return float(h @ w)


def euler8(M):
    """skimage euler_number(image, connectivity=2) (_regionprops_utils.py:140-181)."""
    img = np.pad((M > 0).astype(int), 1, mode="constant")
    XF = ndi.convolve(img, np.array([[0, 0, 0], [0, 1, 4], [0, 2, 8]]), mode="constant", cval=0)
    h = np.bincount(XF.ravel(), minlength=16)
    return int(EULER_COEFS2D_8 @ h)


def _cross(o, a, b):
    return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])


def convex_hull_mask(M):
    """skimage.morphology.convex_hull_image(image) for 2-D (convex_hull.py:73-145): hull of the
    +-0.5 diamond points of the row/column-extremal pixels, then every pixel centre inside or on
    the hull polygon.  Done here in exact integer arithmetic on doubled coordinates."""
    H, W = M.shape
    if not M.any():
        return np.zeros_like(M, bool)
    pts = set()
    rows = np.nonzero(M.any(axis=1))[0]
    first = M.argmax(axis=1)
    last = W - 1 - M[:, ::-1].argmax(axis=1)
    # row-extremal pixels carry the hull (possible_hull also adds column extremes: same hull)
    for r in rows.tolist():
        for c in (int(first[r]), int(last[r])):
            pts.update(((2 * r - 1, 2 * c), (2 * r + 1, 2 * c), (2 * r, 2 * c - 1), (2 * r, 2 * c + 1)))
    P = sorted(pts)
    lower, upper = [], []
    for p in P:
        while len(lower) >= 2 and _cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(P):
        while len(upper) >= 2 and _cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    hull = lower[:-1] + upper[:-1]
    R, Cc = np.mgrid[0:H, 0:W]
    R2, C2 = 2 * R.astype(np.int64), 2 * Cc.astype(np.int64)
    inside = np.ones((H, W), bool)
    n = len(hull)
    for i in range(n):
        a, b = hull[i], hull[(i + 1) % n]
        inside &= ((b[0] - a[0]) * (C2 - a[1]) - (b[1] - a[1]) * (R2 - a[0])) >= 0
    return inside


def calculate_regionprops(mask, img):
    """Same signature and dictionary as the reference's calculate_regionprops
    (detection.py:75-159); ``mask`` is uint8 with 255 on the region, ``img`` the intensity crop."""
    M = mask == 255
    h, w = M.shape
    N = int(M.sum())
    rr, cc = np.nonzero(M)
    m10, m01 = int(rr.sum()), int(cc.sum())
    m20, m02, m11 = int((rr * rr).sum()), int((cc * cc).sum()), int((rr * cc).sum())
    cr, ccn = m10 / N, m01 / N
    mu20 = m20 / N - cr * cr          # central second moments / N
    mu02 = m02 / N - ccn * ccn
    mu11 = m11 / N - cr * ccn
    a, b, c = mu02, -mu11, mu20       # inertia tensor [[a, b], [b, c]] (_moments.py:408-428)
    ev = np.clip(np.linalg.eigvalsh(np.array([[a, b], [b, c]])), 0, None)
    l1, l2 = float(max(ev)), float(min(ev))
    major, minor = 4 * sqrt(l1), 4 * sqrt(l2)
    if a - c == 0:
        orientation = -pi / 4 if b < 0 else pi / 4
    else:
        orientation = 0.5 * atan2(-2 * b, c - a)
    filled = ndi.binary_fill_holes(M, np.ones((3, 3)))
    FA = int(filled.sum())
    vals = (img * M)[M]
    imin, imax, imean = vals.min(), vals.max(), float(np.mean(vals))
    perim = perimeter4(M)
    conv = int(convex_hull_mask(M).sum())
    with np.errstate(divide="ignore", invalid="ignore"):
        return {
            "width": w, "height": h, "bx": 0, "by": 0,
            "circ.": (4 * np.pi * FA) / perim ** 2,
            "area_exc": N, "area_rprops": FA, "%area": 1 - (N / FA),
            "major": major, "minor": minor,
            "centroid_y": float(np.mean(rr)), "centroid_x": float(np.mean(cc)),
            "convex_area": conv,
            "min_intensity": imin, "max_intensity": imax, "mean_intensity": imean,
            "intden": FA * imean,
            "perim.": perim,
            "elongation": np.divide(major, minor),
            "range": imax - imin,
            "perimareaexc": perim / N,
            "perimmajor": perim / major,
            "circex": np.divide(4 * np.pi * N, perim ** 2),
            "angle": orientation / np.pi * 180 + 90,
            "bounding_box_area": h * w,
            "eccentricity": 0 if l1 == 0 else sqrt(1 - l2 / l1),
            "equivalent_diameter": sqrt(4 * N / pi),
            "euler_number": euler8(M),
            "extent": N / (h * w),
            "local_centroid_col": m01 / N, "local_centroid_row": m10 / N,
            "solidity": N / conv,
        }
