"""The only outputs of the hot path that the reference itself holds: ten rows of the object table printed in
pipeline.ipynb cell 1 (tests/golden/reference_notebook_rows.json, transcribed by make_reference_notebook_rows.py).

They pin DERIVED-COLUMN IDENTITIES ONLY -- how the visible columns relate to each other and to the file name -- for the
formulas this build restates (oracle/regionprops_ref.py, imageprocessing_amd/tables.py, props_from_sums in the library).
They cannot pin the OpenCV half of the oracle (contour order, fill, the disk): that stays parity-unpinned.
"""
import json
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = json.load(open(os.path.join(HERE, "golden", "reference_notebook_rows.json")))["rows"]


def test_fixture_shape():
    assert len(ROWS) == 10 and ROWS[0]["object_boundbox_area"] == 3717.0 and ROWS[2]["object_euler_nr"] == 2.0


@pytest.mark.parametrize("r", ROWS, ids=[str(r["row"]) for r in ROWS])
def test_derived_column_identities(r):
    bba, ext, eqd, sol, esd = (r[k] for k in ("object_boundbox_area", "object_extent", "object_equivalent_diameter",
                                              "object_solidity", "esd"))
    # extent = N / (h * w) (R9-table col 29) with an integer pixel count N; six printed decimals of extent
    n_f = ext * bba
    N = round(n_f)
    assert abs(n_f - N) <= 0.6e-6 * bba + 1e-9
    assert bba == int(bba) and 0 < N <= bba
    # equivalent_diameter = sqrt(4 N / pi) (col 27): the value this build computes for that N prints as the notebook's
    assert abs(math.sqrt(4 * N / math.pi) - eqd) <= 0.6e-6
    # solidity = N / convex_area (col 32) with an integer convex area >= N
    conv = N / sol
    assert abs(conv - round(conv)) <= 0.6e-6 * conv / sol + 1e-6 and round(conv) >= N
    # esd = round(2 sqrt(area 27^2 / pi), 2) (EcoTaxa_preparation.py:34-36,115) from the contourArea column: that area is
    # a multiple of 0.5 (shoelace through pixel centres), above min_area_to_save, and below the pixel count
    # plus what other objects' fills add to the window (euler > 1 rows: F8) -- for single-piece rows area < N
    area = math.pi * (esd / 54.0) ** 2
    tol = 2 * area * 0.005 / esd + 1e-9
    assert abs(area * 2 - round(area * 2)) <= 2 * tol
    assert area > 400
    if r["object_euler_nr"] == 1.0:
        assert area < N
    # euler_number is an integer and may exceed 1: the window holds every filled saved object that reaches into it (F8)
    assert r["object_euler_nr"] == int(r["object_euler_nr"])
    # centroids lie inside the window: local_centroid_col < w, local_centroid_row < h, w * h = bounding box area
    assert 0 <= r["object_local_centroid_col"] and 0 <= r["object_local_centroid_row"]
    assert r["object_local_centroid_col"] * r["object_local_centroid_row"] < bba
    assert 0 <= r["object_eccentricity"] < 1


def test_filename_fields_and_esd_through_crop_dataframe():
    """the notebook's file-name columns (date, time, pressure [dbar], temperature, index) and esd come out of
    tables.crop_dataframe for a table holding these ten objects"""
    from imageprocessing_amd import tables
    frames = sorted({r["filename"].rsplit("_", 1)[0] for r in ROWS})
    fnames = [f + ".png" for f in frames]                   # <date>-<time>_<pressure>bar_<temperature>C.png
    fidx = {f: i for i, f in enumerate(frames)}
    rows = sorted(ROWS, key=lambda r: (fidx[r["filename"].rsplit("_", 1)[0]], r["index"]))
    n = len(rows)
    off = np.zeros(len(frames) + 1, np.int32)
    for r in rows:
        off[fidx[r["filename"].rsplit("_", 1)[0]] + 1:] += 1
    area = np.array([round(2 * math.pi * (r["esd"] / 54.0) ** 2) / 2 for r in rows])       # the half-integer the esd came from
    props = np.zeros((n, 32))
    tab = SimpleNamespace(n_frames=len(frames), frame_row_offset=off, frame_valid=np.ones(len(frames), np.uint8),
                          frame_mean=np.zeros(len(frames)), frame_std=np.zeros(len(frames)),
                          c=np.array([r["index"] for r in rows], np.int32), area=area,
                          x=np.zeros(n, np.int32), y=np.zeros(n, np.int32), w=np.full(n, 50, np.int32), h=np.full(n, 50, np.int32),
                          saved=np.ones(n, np.uint8), saved_row=np.arange(n, dtype=np.int32), saved_props=props)
    df = tables.crop_dataframe(tab, fnames, [(0.0, 0.0)] * len(frames), "/data")
    assert len(df) == n
    for (_, g), r in zip(df.iterrows(), sorted(ROWS, key=lambda r: (r["date"], r["time"], r["index"]))):
        assert g["filename"] == r["filename"] and g["index"] == r["index"]
        assert str(g["date"]) == str(r["date"]) and str(g["time"]) == str(r["time"])
        assert abs(g["pressure [dbar]"] - r["pressure [dbar]"]) < 1e-9 and g["temperature"] == r["temperature"]
        assert g["esd"] == r["esd"]
