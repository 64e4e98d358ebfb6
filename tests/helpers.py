"""Shared helpers of the parity tests: run the oracle over a stack and flatten its output into
the same shape as imageprocessing_amd.engine.ObjectTable."""
import numpy as np

from oracle import pipeline_ref as ref
from oracle import cv_ref
from oracle.regionprops_ref import PROP_KEYS


def oracle_bgcorrect(stack, n_bg_imgs=6):
    images = [(stack[i], f"img_{i:05d}.png") for i in range(stack.shape[0])]
    return [ref.correct_img(i, images, n_bg_imgs) for i in range(len(images))]


def oracle_detect(items, mask, want_labels=True, **kw):
    """items: list of correct_img tuples.  Returns list of per-frame dicts from detect_on_img."""
    out = []
    for it in items:
        cleaned = it[1]
        if isinstance(cleaned, list):
            out.append(ref.detect_on_img(it, mask, want_labels=want_labels, **kw))
        else:
            out.append(ref.detect_on_img((it[0].copy(), it[1].copy(), it[2], it[3]), mask, want_labels=want_labels, **kw))
    return out


def make_disk(H, W, radius):
    m = np.zeros((H, W), np.uint8)
    cv_ref.circle(m, (int(H / 2), int(W / 2)), radius, 255, -1)
    return m


def compare_table(tab, dets, H, W, labels=None, check_props=True, rtol=1e-5):
    """tab: ObjectTable from the GPU; dets: oracle per-frame results.  Asserts bit-exact integer
    columns / crops and float columns within rtol."""
    n = len(dets)
    assert tab.n_frames == n
    props_all = tab.props                                   # [n_rows, 32], nan for unsaved rows
    for f, d in enumerate(dets):
        a, b = tab.frame_row_offset[f], tab.frame_row_offset[f + 1]
        if d["corrupt"]:
            assert tab.frame_valid[f] == 0 and a == b, f"frame {f}: corrupt frame must have no rows"
            continue
        assert tab.frame_valid[f] == 1
        rows = d["rows"]
        assert b - a == len(rows), f"frame {f}: {b - a} rows vs oracle {len(rows)}"
        if labels is not None:
            assert np.array_equal(labels[f], d["labels"]), f"frame {f}: label map differs"
            assert tab.frame_n_contours[f] == int(d["labels"].max()), f"frame {f}: contour count"
        crops = {c[0]: c for c in d["crops"]}
        for k, row in enumerate(rows):
            i = a + k
            assert tab.c[i] == row[0] == k
            np.testing.assert_allclose(tab.frame_mean[f], row[4], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(tab.frame_std[f], row[5], rtol=1e-9, atol=1e-9)
            assert tab.area[i] == row[6], f"frame {f} obj {k}: area {tab.area[i]} vs {row[6]}"
            assert tab.saved[i] == row[11]
            if row[11]:
                assert (tab.x[i], tab.y[i], tab.w[i], tab.h[i]) == tuple(row[7:11]), f"frame {f} obj {k}: bbox"
            else:
                assert (tab.x[i] + tab.w[i] / 2, tab.y[i] + tab.h[i] / 2, tab.w[i], tab.h[i]) == tuple(row[7:11])
                assert np.isnan(props_all[i]).all()
        # crops + props of the saved objects
        s_of_row = {int(r): s for s, r in enumerate(tab.saved_row)}
        for k, row in enumerate(rows):
            if not row[11]:
                continue
            i = a + k
            s = s_of_row[i]
            _, masked, raw, cmask = crops[k]
            if tab.crops:
                assert np.array_equal(tab.crop(s, "mask"), cmask), f"frame {f} obj {k}: mask crop"
                assert np.array_equal(tab.crop(s, "masked"), masked), f"frame {f} obj {k}: masked crop"
                assert np.array_equal(tab.crop(s, "raw"), raw), f"frame {f} obj {k}: raw crop"
            if check_props:
                want = np.array([float(v) for v in row[12:]])
                got = props_all[i]
                for j, key in enumerate(PROP_KEYS):
                    if np.isnan(got[j]) and key in check_props_pending():
                        continue
                    if np.isinf(want[j]) or np.isnan(want[j]):
                        assert (np.isinf(got[j]) and np.isinf(want[j])) or (np.isnan(got[j]) and np.isnan(want[j])), key
                        continue
                    np.testing.assert_allclose(got[j], want[j], rtol=rtol, atol=1e-7, err_msg=f"frame {f} obj {k} prop {key}")


def check_props_pending():
    """Property columns not produced on the device yet; the GPU emits NaN for them."""
    from imageprocessing_amd import engine
    return getattr(engine, "PENDING_PROPS", set())
