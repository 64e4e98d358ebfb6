"""tables.crop_dataframe (the crop table straight from an object table) against the restated gen_crop_df reading the
per-frame CSV files back -- host logic, no GPU: the table is assembled from the oracle's detections."""
import csv
import os
from types import SimpleNamespace

import numpy as np
import pytest

from tests import helpers


def _table_from_rows(dets):
    """An ObjectTable-shaped namespace from the oracle's per-frame CSV rows."""
    off, rows = [0], []
    for d in dets:
        rows += [] if d["corrupt"] else d["rows"]
        off.append(len(rows))
    saved = np.array([r[11] for r in rows], np.uint8)
    w = np.array([r[9] for r in rows], np.float64)
    h = np.array([r[10] for r in rows], np.float64)
    x = np.array([r[7] for r in rows], np.float64) - np.where(saved == 0, w / 2, 0)
    y = np.array([r[8] for r in rows], np.float64) - np.where(saved == 0, h / 2, 0)
    first = [d["rows"][0] if (not d["corrupt"] and d["rows"]) else None for d in dets]
    return SimpleNamespace(
        n_frames=len(dets), frame_row_offset=np.array(off, np.int32), frame_valid=np.array([0 if d["corrupt"] else 1 for d in dets], np.uint8),
        frame_mean=np.array([r[4] if r else np.nan for r in first]), frame_std=np.array([r[5] if r else np.nan for r in first]),
        c=np.array([r[0] for r in rows], np.int32), area=np.array([r[6] for r in rows], np.float64),
        x=x.astype(np.int32), y=y.astype(np.int32), w=w.astype(np.int32), h=h.astype(np.int32), saved=saved,
        saved_row=np.nonzero(saved)[0].astype(np.int32),
        saved_props=np.array([[float(v) for v in r[12:]] for r in rows if r[11]], np.float64).reshape(-1, 32))


@pytest.mark.parametrize("min_save,size_filter", [(400, 0), (400, 30), (10**9, 0), (0, 0)])
def test_crop_dataframe_equals_gen_crop_df_of_the_csv_files(tmp_path, min_save, size_filter):
    import pandas as pd
    from imageprocessing_amd import synth, tables
    from oracle import ecotaxa_ref
    H = W = 256
    n = 14
    objs, off = synth.make_objects(n, H, W, 8, per_frame=8, dmin=30, dmax=600)
    stack = synth.render(n, H, W, 8, objs, off)
    stack[3] = 90                                            # corrupt frame: a 14-column CSV, skipped
    order = [5, 2, 9, 0, 13, 1, 7, 3, 11, 4, 6, 8, 10, 12]   # capture order differs from the sorted file names
    names = [f"20230401-12{order[i]:02d}3045_{1.5 + 0.25 * order[i]:06.2f}bar_{11.0 + order[i] / 8:.2f}C.png" for i in range(n)]
    mask = helpers.make_disk(H, W, 120)
    items = [(it[0], it[1], it[2], names[i]) for i, it in enumerate(helpers.oracle_bgcorrect(stack))]
    dets = helpers.oracle_detect(items, mask, want_labels=False, min_area_to_save=min_save)
    if min_save == 0:
        dets[6] = dict(dets[6], rows=[])                     # a frame without objects: an empty CSV, skipped
    data = tmp_path / "Data"
    data.mkdir()
    for d, name in zip(dets, names):
        with open(data / (name[:-4] + ".csv"), "w", newline="") as f:     # detection.py:61-73,293-297
            csv.writer(f, delimiter=",").writerows(["corrupt image!"] if d["corrupt"] else d["rows"])
    tab = _table_from_rows(dets)
    mean_raw = [it[2] for it in items]
    if min_save == 10**9:                                    # nothing saved: the reference fails on the empty split (:84-98)
        with pytest.raises(Exception):
            ecotaxa_ref.gen_crop_df(str(data), False, size_filter)
        with pytest.raises(Exception):
            tables.crop_dataframe(tab, names, mean_raw, str(data), False, size_filter)
        return
    want, n_empty = ecotaxa_ref.gen_crop_df(str(data), False, size_filter)
    got = tables.crop_dataframe(tab, names, mean_raw, str(data), False, size_filter)
    assert n_empty == (1 if min_save == 0 else 0)
    assert len(want) > 5
    assert list(got.columns) == list(want.columns)
    assert list(got.index) == list(want.index)
    # values the CSV stores as decimal text come back through pandas' fast float parser: equal to 1e-12, not to the bit
    pd.testing.assert_frame_equal(got, want, check_exact=False, rtol=1e-12, atol=0)
    assert got["img_id"].is_monotonic_increasing and got["img_id"].iloc[0] == 1
    assert os.path.dirname(got["full_path"].iloc[0]) == str(data)


def test_settings_the_gpu_path_does_not_implement_are_refused():
    """a drop-in refuses what it does not do (detection.py:209-218 `resize`), it does not ignore it; no GPU needed"""
    import pytest
    from imageprocessing_amd import detection
    s = detection.DetectionSettings("d", "c", "dc", "m", "i", 400, 10, 1, False, True, True, True, True, True, 1200)
    with pytest.raises(NotImplementedError, match="resize"):
        detection.check_settings(s)
    with pytest.raises(NotImplementedError, match="resize"):
        detection.detect_batch([], s, None)
    s.resize = False
    detection.check_settings(s)
