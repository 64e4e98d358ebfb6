"""The OpenCV part of the oracle has no executable reference here (cv2 absent, reference ships no
fixtures): PARITY UNPINNED.  These tests tie the sequential restatement (border following, shoelace,
polygon fill) to hand-derived answers and to trace-independent definitions computed with
scipy.ndimage -- the same definitions the HIP path is built on."""
import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import cv_ref as cv

S8 = np.ones((3, 3), int)


def contours(img):
    return cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)[0]


def test_known_answers():
    img = np.zeros((12, 16), np.uint8)
    img[5, 5] = 255
    (c,) = contours(img)
    assert cv.contourArea(c) == 0.0 and cv.boundingRect(c) == (5, 5, 1, 1)          # single pixel
    img[:] = 0
    img[3, 2:11] = 255
    (c,) = contours(img)
    assert cv.contourArea(c) == 0.0 and cv.boundingRect(c) == (2, 3, 9, 1)          # 1 x n line
    assert c.reshape(-1, 2).tolist() == [[2, 3], [10, 3]]
    img[:] = 0
    img[3:5, 2:4] = 255
    assert cv.contourArea(contours(img)[0]) == 1.0                                  # 2 x 2
    for w, h in [(3, 2), (5, 7), (11, 4)]:
        img[:] = 0
        img[1:1 + h, 2:2 + w] = 255
        (c,) = contours(img)
        assert cv.contourArea(c) == (w - 1) * (h - 1)                               # solid rectangle
        assert c.reshape(-1, 2).tolist() == [[2, 1], [2, h], [w + 1, h], [w + 1, 1]]   # counter-clockwise on screen
    img[:] = 0
    img[4, 6] = img[6, 6] = img[5, 5] = img[5, 7] = 255                             # diamond ring, isolated centre
    img[5, 6] = 255
    cs = contours(img)
    assert len(cs) == 1 and cv.contourArea(cs[0]) == 2.0
    m = np.zeros_like(img)
    cv.drawContours(m, cs, -1, 255, cv.FILLED)
    assert m.sum() == 5 * 255


def test_order_nesting_and_border():
    img = np.zeros((20, 20), np.uint8)
    img[2:14, 2:14] = 255
    img[4:12, 4:12] = 0          # ring
    img[6:9, 6:9] = 255          # blob nested in the hole: not an external contour
    img[16:20, 0:3] = 255        # touches the frame edge, found last -> returned first
    img[0, 19] = 255             # top-right pixel, found first -> returned last
    cs = contours(img)
    assert [c.reshape(-1, 2)[0].tolist() for c in cs] == [[0, 16], [2, 2], [19, 0]]
    assert [cv.contourArea(c) for c in cs] == [6.0, 121.0, 0.0]
    lab = cv.label_map(img, cs)
    assert lab[7, 7] == 0 and lab[2, 2] == 2 and lab[17, 1] == 1 and lab[0, 19] == 3
    m = np.zeros_like(img)
    cv.drawContours(m, [cs[1]], -1, 255, cv.FILLED)
    assert (m[2:14, 2:14] == 255).all() and m.sum() == 144 * 255                    # hole and nested blob filled


def test_area_thresholds_exact():
    # contourArea exactly 10.0 (kept by "< 10 -> continue") and exactly 400.0 (not saved by "> 400")
    img = np.zeros((40, 40), np.uint8)
    img[1:4, 1:7] = 255          # 6 x 3 -> (5)(2) = 10
    img[10:31, 10:31] = 255      # 21 x 21 -> 400
    a = sorted(cv.contourArea(c) for c in contours(img))
    assert a == [10.0, 400.0]


def _topological(img):
    """External components, quad-count area, bbox, hole-filled mask -- no border following."""
    fg = img > 0
    lab, n = ndi.label(fg, structure=S8)
    bl, _ = ndi.label(np.pad(~fg, 1, constant_values=True))
    out = []
    cross = ndi.generate_binary_structure(2, 1)
    for k in range(1, n + 1):
        comp = lab == k
        if not (ndi.binary_dilation(np.pad(comp, 1), structure=cross) & (bl == bl[0, 0])).any():
            continue                                            # enclosed by a hole of another component
        filled = ndi.binary_fill_holes(comp)
        rr, cc = np.nonzero(comp)
        F = np.pad(filled, 1).astype(int)
        q = F[:-1, :-1] + F[:-1, 1:] + F[1:, :-1] + F[1:, 1:]
        out.append(((rr[0], cc[0]), (q == 4).sum() + 0.5 * (q == 3).sum(),
                    (cc.min(), rr.min(), cc.max() - cc.min() + 1, rr.max() - rr.min() + 1), filled, comp))
    out.sort(key=lambda t: t[0], reverse=True)                  # reverse raster order of the first pixel
    return out


@pytest.mark.parametrize("seed", range(4))
def test_trace_equals_topological_definition(seed):
    rng = np.random.default_rng(seed)
    for it in range(250):
        H, W = rng.integers(3, 40, 2)
        p = rng.choice([0.1, 0.3, 0.45, 0.55, 0.6, 0.7, 0.9])
        img = (rng.random((H, W)) < p).astype(np.uint8) * 255
        if it % 3 == 0:
            img = ndi.binary_dilation(img > 0).astype(np.uint8) * 255
        cs = contours(img)
        top = _topological(img)
        assert len(cs) == len(top)
        lab = cv.label_map(img, cs)
        for k, (c, t) in enumerate(zip(cs, top)):
            p0 = c.reshape(-1, 2)[0]
            assert (p0[1], p0[0]) == t[0]
            assert cv.contourArea(c) == t[1]
            assert cv.boundingRect(c) == tuple(int(v) for v in t[2])
            m = np.zeros_like(img)
            cv.drawContours(m, [c], -1, 255, cv.FILLED)
            assert np.array_equal(m == 255, t[3])
            assert np.array_equal(lab == k + 1, t[4])


def test_circle_is_symmetric_and_monotone():
    for r in (1, 2, 7, 50, 120):
        m = np.zeros((2 * r + 9, 2 * r + 9), np.uint8)
        c = r + 4
        cv.circle(m, (c, c), r, 255, -1)
        b = m == 255
        assert np.array_equal(b, b[::-1]) and np.array_equal(b, b[:, ::-1]) and np.array_equal(b, b.T)
        rows = b.sum(axis=1)
        assert rows[c] == 2 * r + 1 and rows[c - r] >= 1 and rows[c - r - 1] == 0
        assert (np.diff(rows[:c + 1]) >= 0).all()
        yy, xx = np.nonzero(b)
        d2 = (yy - c) ** 2 + (xx - c) ** 2
        assert d2.max() <= (r + 0.5) ** 2 + 1 and (~b[(np.mgrid[0:m.shape[0], 0:m.shape[1]][0] - c) ** 2 + (np.mgrid[0:m.shape[0], 0:m.shape[1]][1] - c) ** 2 <= (r - 1) ** 2]).sum() == 0
