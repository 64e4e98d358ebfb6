"""Pins oracle/regionprops_ref.py to scikit-image: the 32 columns of calculate_regionprops against
tests/golden/regionprops_skimage.npz (scikit-image 0.18.3 run in the build container)."""
import os

import numpy as np

from oracle.regionprops_ref import PROP_KEYS, calculate_regionprops

GOLD = os.path.join(os.path.dirname(__file__), "golden", "regionprops_skimage.npz")


def load_cases():
    g = np.load(GOLD, allow_pickle=False)
    assert list(g["keys"]) == PROP_KEYS
    shapes, exp = g["shapes"], g["expected"]
    mo = io = 0
    cases = []
    for (h, w), e in zip(shapes, exp):
        nb = (h * w + 7) // 8
        m = np.unpackbits(g["masks"][mo:mo + nb])[:h * w].reshape(h, w).astype(bool)
        img = g["imgs"][io:io + h * w].reshape(h, w)
        mo += nb
        io += h * w
        cases.append((m, img, e))
    return cases


def test_oracle_matches_skimage_golden():
    cases = load_cases()
    assert len(cases) >= 160
    for n, (m, img, e) in enumerate(cases):
        got = calculate_regionprops((m * 255).astype(np.uint8), img)
        assert list(got.keys()) == PROP_KEYS
        for k, want in zip(PROP_KEYS, e):
            g = float(got[k])
            if np.isnan(want) or np.isinf(want):
                assert (np.isnan(g) and np.isnan(want)) or g == want, (n, k, g, want)
            else:
                np.testing.assert_allclose(g, want, rtol=1e-9, atol=1e-9, err_msg=f"case {n} {k}")
