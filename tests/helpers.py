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


def torch_bgcorrect(frames, n_bg_imgs=6):
    """bg_correction.py:72-81 restated with torch uint8 operations on the device the frames live on (size-independent
    check of full-size batches: seconds for 200 frames of 2560 x 2560).  Returns (cleaned uint8 [n,H,W], raw sum int64 [n],
    raw sum of squares int64 [n])."""
    import torch
    n = frames.shape[0]
    nw = 2 * n_bg_imgs
    pair = torch.minimum(frames[:-1], frames[1:])              # pair[k] = min(I[k], I[k+1])  (:76)
    cleaned = torch.empty_like(frames)
    for i in range(n):
        start = i if i + nw < n else n - nw                    # :72
        bg = pair[start:start + nw - 1].amax(dim=0)            # :78
        img = frames[i]
        cleaned[i] = 255 - (torch.maximum(img, bg) - torch.minimum(img, bg))      # :80-81
    x = frames.view(n, -1)
    s1 = torch.stack([x[i].sum(dtype=torch.int64) for i in range(n)])
    s2 = torch.stack([(x[i].to(torch.int32) ** 2).sum(dtype=torch.int64) for i in range(n)])
    return cleaned, s1.cpu().numpy(), s2.cpu().numpy()


def mean_std_from_sums(s1, s2, cnt):
    """np.mean / np.std (population) from exact integer sums"""
    m = s1 / cnt
    return m, np.sqrt(np.maximum(s2 / cnt - m * m, 0.0))


def check_batch_properties(tab, labels, cleaned, mask, frames_to_check=None, n_rows_sample=60, n_crops_sample=40, seed=0):
    """Size-independent consistency of one processed batch, computed with torch on the device (detection.py:178-247):
    masked statistics recomputed from the cleaned frames, label map <-> threshold plane, label map <-> table (bbox, contour
    count), crop identities.  tab needs fetched crops for the crop part."""
    import torch
    n = tab.n_frames
    inside = (mask == 255) if mask is not None else torch.ones_like(cleaned[0], dtype=torch.bool)
    cnt = int(inside.sum())
    for f in (range(n) if frames_to_check is None else frames_to_check):
        if not tab.frame_valid[f]:
            assert tab.frame_row_offset[f + 1] == tab.frame_row_offset[f] and int(labels[f].max()) == 0
            continue
        corr = (255 - cleaned[f].to(torch.int64)) * inside            # detection.py:178-188, exact integers
        s1, s2 = int(corr.sum()), int((corr * corr).sum())
        np.testing.assert_allclose(tab.frame_mean[f], s1 / cnt, rtol=1e-12)
        np.testing.assert_allclose(tab.frame_std[f], np.sqrt(s2 / cnt - (s1 / cnt) ** 2), rtol=1e-9)
        lab = labels[f]
        assert int(lab.max()) == tab.frame_n_contours[f]
        thr = corr > 10
        assert bool((lab > 0).logical_and(~thr).any()) is False       # labelled pixels are thresholded pixels
    rng = np.random.default_rng(seed)
    if tab.n_rows:
        for i in rng.choice(tab.n_rows, min(n_rows_sample, tab.n_rows), replace=False):
            f, k = int(tab.frame[i]), int(tab.label[i])
            ys, xs = torch.nonzero(labels[f] == k, as_tuple=True)
            assert len(ys) > 0
            box = (int(xs.min()), int(ys.min()), int(xs.max() - xs.min()) + 1, int(ys.max() - ys.min()) + 1)
            assert box == (tab.x[i], tab.y[i], tab.w[i], tab.h[i])
            assert len(ys) <= 2 * tab.area[i] + 2 * (tab.w[i] + tab.h[i]) + 4       # area bounds the pixel count (Pick)
    assert tab.crop_bytes == int((tab.w[tab.saved_row].astype(np.int64) * tab.h[tab.saved_row]).sum())
    if not tab.crops or not len(tab.saved_row):
        return
    hc = {}
    for s in rng.choice(len(tab.saved_row), min(n_crops_sample, len(tab.saved_row)), replace=False):
        i = int(tab.saved_row[s])
        f = int(tab.frame[i])
        if f not in hc:
            hc[f] = (cleaned[f].cpu().numpy(), labels[f].cpu().numpy())
        x, y, w, h = tab.x[i], tab.y[i], tab.w[i], tab.h[i]
        raw, msk, msd = tab.crop(s, "raw"), tab.crop(s, "mask"), tab.crop(s, "masked")
        assert np.array_equal(raw, hc[f][0][y:y + h, x:x + w])
        assert set(np.unique(msk)) <= {0, 255}
        assert np.array_equal(msd, np.where(msk == 255, raw, 255))
        own = hc[f][1][y:y + h, x:x + w] == tab.label[i]
        assert (msk[own] == 255).all()                                # the object's own pixels are in its mask
        assert tab.saved_props[s][5] == (msk == 255).sum()            # area_exc = pixels of the mask window


def compare_frame_with_oracle(tab, labels, cleaned, ms, i, images, hmask, n_bg_imgs=6, rtol=1e-5):
    """frame i of a processed full-size batch against the oracle: cleaned, raw statistics, label map, every row"""
    it = ref.correct_img(i, images, n_bg_imgs)
    assert np.array_equal(cleaned[i].cpu().numpy(), it[1])
    np.testing.assert_allclose(ms[i], it[2], rtol=1e-9)
    d = ref.detect_on_img((it[0].copy(), it[1].copy(), it[2], it[3]), hmask, want_labels=True)
    assert np.array_equal(labels[i].cpu().numpy(), d["labels"])
    a, b = tab.frame_row_offset[i], tab.frame_row_offset[i + 1]
    assert b - a == len(d["rows"])
    props = tab.props
    s_of_row = {int(r): s for s, r in enumerate(tab.saved_row)}
    crops = {c[0]: c for c in d["crops"]}
    for k, row in enumerate(d["rows"]):
        assert tab.c[a + k] == k
        assert tab.area[a + k] == row[6] and tab.saved[a + k] == row[11]
        if row[11]:
            assert (tab.x[a + k], tab.y[a + k], tab.w[a + k], tab.h[a + k]) == tuple(row[7:11])
            np.testing.assert_allclose(props[a + k], np.array([float(v) for v in row[12:]]), rtol=rtol, atol=1e-7)
            if tab.crops:
                s = s_of_row[a + k]
                _, masked, raw, cmask = crops[k]
                assert np.array_equal(tab.crop(s, "mask"), cmask) and np.array_equal(tab.crop(s, "masked"), masked)
                assert np.array_equal(tab.crop(s, "raw"), raw)
        else:
            assert (tab.x[a + k] + tab.w[a + k] / 2, tab.y[a + k] + tab.h[a + k] / 2, tab.w[a + k], tab.h[a + k]) == tuple(row[7:11])
    return len(d["rows"]), sum(r[11] for r in d["rows"])
