"""The drop-in entry points of the reference executed end to end on the GPU: run_segmenter on a folder of PNG files
(segmenter.py:108-265), run_detection through a real queue (detection.py:300-344), compute_radius (segmenter.py:28-63),
run_reader / read_img incl. the device resize (reader.py:54-110), pseg_set_stream -- every file they write compared with
the oracle."""
import csv
import os
import random
import threading
from queue import Queue

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

H = W = 2560


def _parse(v):
    try:
        return float(v)
    except ValueError:
        return v


def compare_outputs(save_path, names, stack, mask):
    """Data/<stem>.csv, Crops/, Deconv_crops/, Masks/ under save_path against the oracle run over ``stack`` (the frames in
    batch order, one batch).  Returns (rows, saved crops) checked."""
    from PIL import Image
    o_items = helpers.oracle_bgcorrect(stack)
    nrows = ncrops = 0
    for it, name in zip(o_items, names):
        det = helpers.oracle_detect([(it[0], it[1], it[2], name)], mask, want_labels=False)[0]
        got = list(csv.reader(open(os.path.join(save_path, "Data", name[:-4] + ".csv"))))
        if det["corrupt"]:
            assert got == [list("corrupt image!")]          # csv.writer splits the string into characters (detection.py:297)
            continue
        assert len(got) == len(det["rows"]), name
        for grow, orow in zip(got, det["rows"]):
            assert len(grow) == 44 and grow[1] == orow[1]
            assert int(grow[0]) == orow[0] and int(grow[11]) == orow[11]
            for j in list(range(2, 11)) + list(range(12, 44)):
                g, o = _parse(grow[j]), float(orow[j])
                if np.isnan(o):
                    assert grow[j] == "nan"
                elif np.isinf(o):
                    assert np.isinf(g)
                else:
                    np.testing.assert_allclose(g, o, rtol=1e-5, atol=1e-7)
            if orow[11]:
                assert "." not in grow[7] and "." not in grow[12] and "." not in grow[25]     # ints print as ints
            else:
                assert "." in grow[7]                                                          # centre floats (detection.py:271)
            nrows += 1
        for c, masked, raw, cmask in det["crops"]:
            fn = f"{name[:-4]}_{c}.png"
            assert np.array_equal(np.asarray(Image.open(os.path.join(save_path, "Deconv_crops", fn))), masked)
            assert np.array_equal(np.asarray(Image.open(os.path.join(save_path, "Crops", fn))), raw)
            assert np.array_equal(np.asarray(Image.open(os.path.join(save_path, "Masks", fn))), cmask)
            ncrops += 1
        n_png = sum(1 for f in os.listdir(os.path.join(save_path, "Crops")) if f.startswith(name[:-4] + "_"))
        assert n_png == len(det["crops"])
    return nrows, ncrops


@pytest.fixture(scope="module")
def frames24():
    from imageprocessing_amd import synth
    from imageprocessing_amd.engine import Segmenter
    seg = Segmenter(H, W, max_frames=24, device=0)
    objs, off = synth.make_objects(24, H, W, 8, per_frame=12, dmin=20.0, dmax=300.0, ring_frac=0.3)
    stack = seg.synth_frames(24, 8, objs, off).cpu().numpy()
    seg.close()
    stack[12] = 120                                         # a corrupt frame (std = 0); not among the first 10: compute_radius
                                                            # takes their maximum and would find no dark surround
    return stack


def test_run_segmenter_on_a_folder_of_png_files(tmp_path, frames24):
    from PIL import Image
    from imageprocessing_amd import segmenter
    from oracle import cv_resize_ref, pipeline_ref
    src, dst = tmp_path / "PNG", tmp_path / "out"
    src.mkdir()
    stack = frames24.copy()
    names = [f"PISCO-{i:04d}_20240801_img.png" for i in range(24)]
    small = stack[16][::2, ::2].copy()                      # frame 16 is stored at 1280 x 1280: the reader's resize (reader.py:79)
    stack[16] = cv_resize_ref.resize_linear_u8(small, H, W)
    order = list(range(24))
    random.Random(3).shuffle(order)                         # directory order is not time order
    for i in order:
        Image.fromarray(small if i == 16 else stack[i]).save(src / names[i], compress_level=1)
    (src / "PISCO-0100_empty.png").write_bytes(b"")                      # imghdr.what -> None: filtered (segmenter.py:162)
    (src / "PISCO-0101_text.png").write_bytes(b"this is not an image\n" * 4)
    (src / "notes.txt").write_text("ignored")
    good = (src / names[5]).read_bytes()
    (src / "PISCO-0102_truncated.png").write_bytes(good[:len(good) // 3])  # valid header, broken body: the reader fails on it
    segmenter.run_segmenter(str(src), str(dst), False)
    # settings.csv (segmenter.py:65-84) and the radius (:28-63)
    rows = list(csv.reader(open(dst / "settings.csv")))
    assert rows[0] == ["Field Name", "Value"] and rows[1] == ["data source", str(src)]
    sett = dict(rows[2:])
    radius = pipeline_ref.compute_radius(stack[:10])
    assert int(sett["mask_radius"]) == radius == 1200
    assert sett["min_area_to_save"] == "400" and sett["min_area_to_segment"] == "10" and sett["mask_img"] == "True"
    assert [r[0] for r in rows[2:]] == ["data_path", "raw_crop_path", "deconv_crop_path", "mask_path", "img_path", "min_area_to_save",
                                       "min_area_to_segment", "n_sigma", "save_bb_image", "save_crops", "equalize_hist", "resize",
                                       "clear_save_path", "mask_img", "mask_radius"]
    # the unreadable file is reported like a corrupt frame and takes no part in the batch
    assert list(csv.reader(open(dst / "Data" / "PISCO-0102_truncated.csv"))) == [list("corrupt image!")]
    assert not (dst / "Data" / "PISCO-0100_empty.csv").exists() and not (dst / "Data" / "PISCO-0101_text.csv").exists()
    assert len(os.listdir(dst / "Data")) == 25
    mask = helpers.make_disk(H, W, radius)
    nrows, ncrops = compare_outputs(str(dst), names, stack, mask)
    assert nrows > 150 and ncrops > 60
    # resume (segmenter.py:156-161): nothing is left to do, no file is touched
    before = {f: os.path.getmtime(dst / "Data" / f) for f in os.listdir(dst / "Data")}
    with pytest.raises(Exception):
        segmenter.run_segmenter(str(src), str(dst), False)  # the reference fails the same way on an empty file list (compute_radius)
    assert before == {f: os.path.getmtime(dst / "Data" / f) for f in os.listdir(dst / "Data")}


def test_run_detection_through_a_queue_and_run_reader(tmp_path, frames24):
    """reader -> bg stage -> queue -> run_detection at the reference's frame size, host arrays on the queue (what an
    unchanged consumer such as deconvolution.py sees), one unreadable file in the batch."""
    from PIL import Image
    from imageprocessing_amd import bg_correction, detection, reader
    n = 14
    stack = frames24[:n]
    src = tmp_path / "in"
    src.mkdir()
    names = [f"SO298-{i:04d}_x.png" for i in range(n)]
    for i in range(n):
        Image.fromarray(stack[i]).save(src / names[i], compress_level=1)
    (src / "SO298-0014_bad.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 50)
    files = [(str(src / nm), i) for i, nm in enumerate(names + ["SO298-0014_bad.png"])]
    out = reader.ReaderOutput(n + 1)
    reader.run_reader(files, out, 4, False)
    assert out.done.is_set() and list(out.failed) == [n] and all(out.images[i] is not None for i in range(n))
    assert np.array_equal(out.images[3][0], stack[3]) and out.images[3][1] == names[3]
    q = Queue(maxsize=0)
    bg_correction.run_bg_correction(out, q, 6, None).join()
    assert q.qsize() == n + 1
    dirs = {k: str(tmp_path / k) for k in ("Data", "Crops", "Deconv_crops", "Masks", "Images")}
    for d in dirs.values():
        os.makedirs(d)
    settings = detection.DetectionSettings(dirs["Data"], dirs["Crops"], dirs["Deconv_crops"], dirs["Masks"], dirs["Images"],
                                           400, 10, 1, False, True, True, False, True, True, 1150)
    detection.run_detection(q, settings, 8, n + 1, None)
    assert q.empty()
    assert list(csv.reader(open(os.path.join(dirs["Data"], "SO298-0014_bad.csv")))) == [list("corrupt image!")]
    nrows, ncrops = compare_outputs(str(tmp_path), names, stack, helpers.make_disk(H, W, 1150))
    assert nrows > 80 and ncrops > 30


def test_compute_radius_and_the_three_disks(tmp_path, frames24):
    """compute_radius against the oracle's restatement of segmenter.py:45-63 (also on frames whose dark surround begins
    further in), and the three statements of cv.circle -- segmenter._disk, the oracle, pseg_disk_mask -- against each other."""
    from PIL import Image
    from imageprocessing_amd import segmenter
    from imageprocessing_amd.engine import Segmenter
    from oracle import cv_ref, pipeline_ref
    files = []
    for i in range(10):
        Image.fromarray(frames24[i]).save(tmp_path / f"a-{i}_.png", compress_level=1)
        files.append(str(tmp_path / f"a-{i}_.png"))
    assert segmenter.compute_radius(files) == pipeline_ref.compute_radius(frames24[:10]) == 1200
    yy, xx = np.mgrid[0:H, 0:W]
    r2 = (yy - 1280) ** 2 + (xx - 1280) ** 2
    for r_dark in (1062, 1100, 1149):
        fr = frames24[:10].copy()
        fr[:, r2 > r_dark * r_dark] = 20
        assert segmenter.radius_from_frames(fr) == pipeline_ref.compute_radius(fr)
    seg = Segmenter(H, W, max_frames=1, device=0)
    for radius in (1000, 1050, 1137, 1200, 1250, 1300):
        o = np.zeros((H, W), np.uint8)
        cv_ref.circle(o, (1280, 1280), radius, 255, -1)
        assert np.array_equal(segmenter._disk((H, W), radius), o)
        assert np.array_equal(seg.disk_mask(radius).cpu().numpy(), o)
    seg.close()


def test_native_reader_and_device_resize(tmp_path):
    """run_reader(native=True) leaves the frames at their decoded size; the background stage resizes them on the device with
    the reader's cv.resize arithmetic (reader.py:79) -- frames of two different sizes in one batch."""
    from PIL import Image
    from imageprocessing_amd import bg_correction, reader
    from oracle import cv_resize_ref
    rng = np.random.default_rng(12)
    n, T = 13, (256, 320)
    small = [rng.integers(0, 256, (128, 160) if i % 3 else (200, 300), dtype=np.uint8) for i in range(n)]
    files = []
    for i, im in enumerate(small):
        Image.fromarray(im).save(tmp_path / f"f-{i}_.png")
        files.append((str(tmp_path / f"f-{i}_.png"), i))
    out = reader.ReaderOutput(n)
    reader.run_reader(files, out, 3, False, native=True)
    assert all(out.images[i][0].shape == small[i].shape for i in range(n))
    q = Queue()
    bg_correction.run_bg_correction(out, q, 6, None, resize_to=T).join()
    stack = np.stack([cv_resize_ref.resize_linear_u8(im, *T) for im in small])
    want = helpers.oracle_bgcorrect(stack)
    for i in range(n):
        got = q.get()
        assert got[3] == f"f-{i}_.png"
        assert np.array_equal(got[1], want[i][1])
        np.testing.assert_allclose(got[2], want[i][2], rtol=1e-9)


def test_reader_failure_never_blocks_the_stages():
    """A reader that dies, a batch with fewer readable frames than the window, frames of mixed sizes without a resize target:
    the background stage still puts one tuple per frame (run_detection would wait forever otherwise)."""
    from imageprocessing_amd import bg_correction, reader
    rng = np.random.default_rng(2)
    # (a) the reader finishes having filled 3 of 8 slots: fewer frames than n_bg_imgs -> the library refuses, all corrupt
    out = reader.ReaderOutput(8)
    for i in (0, 2, 5):
        out.add_output(rng.integers(0, 256, (64, 64), dtype=np.uint8), f"x{i}.png", i)
    out.done.set()
    q = Queue()
    bg_correction.run_bg_correction(out, q, 6, None).join(timeout=60)
    items = [q.get(timeout=5) for _ in range(8)]
    assert all(isinstance(it[1], list) for it in items) and [it[3] for it in items][0] == "x0.png"
    # (b) mixed frame sizes without resize_to: an error inside the stage, printed, and still 13 tuples
    out = reader.ReaderOutput(13)
    for i in range(13):
        out.add_output(rng.integers(0, 256, (64, 64) if i else (64, 128), dtype=np.uint8), f"y{i}.png", i)
    q = Queue()
    bg_correction.run_bg_correction(out, q, 6, None).join(timeout=60)
    assert q.qsize() == 13 and all(isinstance(q.get()[1], list) for _ in range(13))
    # (c) one failed slot among 13: the other 12 form the batch
    out = reader.ReaderOutput(13)
    frames = rng.integers(0, 256, (13, 64, 64), dtype=np.uint8)
    for i in range(13):
        if i == 4:
            out.mark_failed("y4.png", 4)
        else:
            out.add_output(frames[i], f"y{i}.png", i)
    q = Queue()
    bg_correction.run_bg_correction(out, q, 6, None).join(timeout=60)
    items = [q.get(timeout=5) for _ in range(13)]
    want = helpers.oracle_bgcorrect(np.delete(frames, 4, axis=0))
    assert isinstance(items[4][1], list) and items[4][3] == "y4.png"
    for it, w in zip(items[:4] + items[5:], want):
        assert np.array_equal(it[1], w[1])


def test_set_stream_runs_the_context_on_the_callers_stream():
    """pseg_set_stream: the context's kernels are queued on a stream of the caller (here a torch stream), so that work queued
    on that stream before the call is ordered before it without a host synchronisation."""
    import torch
    from imageprocessing_amd.engine import Segmenter
    H2, W2, n = 128, 192, 20
    seg = Segmenter(H2, W2, max_frames=n, device=0)
    ref = Segmenter(H2, W2, max_frames=n, device=0)
    try:
        st = torch.cuda.Stream()
        assert seg.L.pseg_set_stream(seg.h, st.cuda_stream) == 0
        assert seg.L.pseg_set_stream(None, st.cuda_stream) == -1
        rng = np.random.default_rng(4)
        base = torch.from_numpy(rng.integers(0, 200, (n, H2, W2), dtype=np.uint8)).cuda()
        torch.cuda.synchronize()
        frames = torch.empty_like(base)
        with torch.cuda.stream(st):
            big = torch.ones((4096, 4096), device="cuda")
            for _ in range(20):                             # keeps the stream busy while the call is queued behind it
                big = big @ big * 1e-4
            frames.copy_(base)
            frames += 7                                     # the library must see these writes: same stream, no host sync
        want = ref.process(base + 7, threshold=60)
        got = seg.process(frames, threshold=60)
        assert torch.equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert np.array_equal(got[3].area, want[3].area) and got[3].n_rows > 0
    finally:
        seg.close()
        ref.close()
