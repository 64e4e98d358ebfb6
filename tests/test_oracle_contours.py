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


def test_more_known_answers():
    """further hand-derived cases of SURVEY.md 8(c): hollow diamond, diagonal contact, spur, ring in ring"""
    img = np.zeros((12, 16), np.uint8)
    img[4, 6] = img[6, 6] = img[5, 5] = img[5, 7] = 255                             # diamond ring, centre empty
    cs = contours(img)
    assert len(cs) == 1 and cv.contourArea(cs[0]) == 2.0 and cv.boundingRect(cs[0]) == (5, 4, 3, 3)
    m = np.zeros_like(img)
    cv.drawContours(m, cs, -1, 255, cv.FILLED)
    assert m[5, 6] == 255 and m.sum() == 5 * 255          # the centre is enclosed (background is 4-connected): filled
    img[:] = 0
    img[2:5, 2:5] = 255
    img[5:8, 5:8] = 255                                   # two squares touching at a corner: ONE 8-connected object
    cs = contours(img)
    assert len(cs) == 1 and cv.boundingRect(cs[0]) == (2, 2, 6, 6)
    assert cv.contourArea(cs[0]) == 8.0                   # two 2 x 2 polygons of pixel centres; the diagonal link adds nothing
    assert cs[0].reshape(-1, 2).tolist() == [[2, 2], [2, 4], [4, 4], [5, 5], [5, 7], [7, 7], [7, 5], [5, 5], [4, 4], [4, 2]]
    img[:] = 0
    img[3:7, 3:7] = 255
    img[4, 7:12] = 255                                    # a 1-pixel spur: traversed out and back, zero area of its own;
    (c,) = contours(img)                                  # the border cuts the two corners at its foot diagonally: + 2 x 1/2
    assert cv.contourArea(c) == 9.0 + 1.0 and cv.boundingRect(c) == (3, 3, 9, 4)
    assert c.reshape(-1, 2).tolist() == [[3, 3], [3, 6], [6, 6], [6, 5], [7, 4], [11, 4], [7, 4], [6, 3]]
    img = np.zeros((24, 24), np.uint8)
    img[1:23, 1:23] = 255; img[3:21, 3:21] = 0            # ring
    img[5:19, 5:19] = 255; img[7:17, 7:17] = 0            # ring inside its hole: not external
    img[10:13, 10:13] = 255                               # blob inside the inner ring's hole: not external either
    cs = contours(img)
    assert len(cs) == 1 and cv.contourArea(cs[0]) == 21.0 * 21.0
    lab = cv.label_map(img, cs)
    assert lab[1, 1] == 1 and lab[5, 5] == 0 and lab[11, 11] == 0
    m = np.zeros_like(img)
    cv.drawContours(m, cs, -1, 255, cv.FILLED)
    assert (m[1:23, 1:23] == 255).all() and m.sum() == 22 * 22 * 255


def test_cumulative_mask_and_iteration_order():
    """detection.py:206,222,245-246: ONE mask per frame collects the FILLED contours of the saved objects in iteration order
    (OpenCV's: last found first).  Object A lies inside the bounding box of the L-shaped object B and below B's first
    pixel, so A is iterated first (c = 0): A's mask window shows A alone although B's bar crosses A's bounding box... no: B
    is not drawn yet; B's window (c = 1) also shows A's fill."""
    from oracle import pipeline_ref as ref
    H, W = 120, 160
    cleaned = np.full((H, W), 255, np.uint8)
    cleaned[10:100, 10:20] = 50                           # B: vertical bar ...
    cleaned[10:20, 10:150] = 50                           # ... and horizontal bar (an L turned upside down); first pixel (10, 10)
    cleaned[40:70, 60:100] = 80                           # A: 40 x 30 block inside B's bounding box; first pixel (40, 60)
    cleaned[75:80, 15:70] = 120                           # C: a thin bar touching nothing... it touches B's vertical bar: part of B
    det = ref.detect_on_img((cleaned, cleaned, [100.0, 50.0], "f_0001.png"), None, mask_img=False)
    rows, crops = det["rows"], {c[0]: c for c in det["crops"]}
    assert [r[0] for r in rows] == [0, 1] and [r[11] for r in rows] == [1, 1]
    # c = 0 is A (found last), c = 1 is B (+ the bar C joined to it)
    assert tuple(rows[0][7:11]) == (60, 40, 40, 30) and rows[0][6] == 39.0 * 29.0
    assert tuple(rows[1][7:11]) == (10, 10, 140, 90)
    _, masked_a, raw_a, mask_a = crops[0]
    _, masked_b, raw_b, mask_b = crops[1]
    assert (mask_a == 255).all()                          # A alone (C's bar ends at x = 69, y 75..79: outside A's box)
    assert mask_b.shape == (90, 140)
    assert (mask_b[30:60, 50:90] == 255).all()            # A's fill inside B's window: drawn one iteration earlier
    want_b = np.zeros((90, 140), bool)
    want_b[0:90, 0:10] = True; want_b[0:10, 0:140] = True; want_b[65:70, 5:60] = True; want_b[30:60, 50:90] = True
    assert np.array_equal(mask_b == 255, want_b)
    assert np.array_equal(masked_b, np.where(want_b, cleaned[10:100, 10:150], 255))
    assert rows[1][12 + 5] == want_b.sum()                # area_exc of B's row counts A's pixels too (regionprops of the window)
    assert np.array_equal(raw_b, cleaned[10:100, 10:150])
