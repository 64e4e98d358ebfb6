"""BASELINE.json configs[3] -- UVP5-like density, ~2000 small particles per frame -- through the C ABI against the oracle
(detection.py:222-273 under density: many kept rows, overlapping bounding boxes, the cumulative mask).  Two workloads:
``uvp5_dense`` (diameters 2-12 px: rows, but no object above min_area_to_save) and ``uvp5_dense_saved`` (diameters 8-40 px:
~550 saved objects per frame).  At 512 x 512 the particle density per pixel and the pixel diameters are those of the full
frame and every frame is compared with the oracle; at 2560 x 2560 one frame is compared with the oracle and the batch
through the size-independent properties."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def _scaled(cfg_name, W):
    """the config's object statistics at frame width W: same particles per pixel, same diameters in pixels"""
    from imageprocessing_amd import synth
    cfg = dict(synth.CONFIGS[cfg_name])
    sc = 2560 / W
    cfg["per_frame"] = int(round(cfg["per_frame"] / sc / sc))
    cfg["dmin"] *= sc                                      # make_objects scales diameters by W / 2560
    cfg["dmax"] *= sc
    return cfg


@pytest.mark.parametrize("cfg_name,min_saved", [("uvp5_dense", 0), ("uvp5_dense_saved", 10)])
def test_dense_parity_small(gpu_seg_factory, cfg_name, min_saved):
    import torch
    from imageprocessing_amd import synth
    H = W = 512
    n = 13
    seg = gpu_seg_factory(H, W)
    objs, off = synth.make_objects(n, H, W, 4, **_scaled(cfg_name, W))
    stack = synth.render(n, H, W, 4, objs, off)
    mask = helpers.make_disk(H, W, 240)
    items = helpers.oracle_bgcorrect(stack)
    dets = helpers.oracle_detect(items, mask)
    labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
    cleaned, ms, valid, tab = seg.process(torch.from_numpy(stack).cuda(), mask=torch.from_numpy(mask).cuda(), labels=labels,
                                          fetch_crops=True)
    cl = cleaned.cpu().numpy()
    for i, it in enumerate(items):
        assert np.array_equal(cl[i], it[1])
    helpers.compare_table(tab, dets, H, W, labels=labels.cpu().numpy())
    per_frame = np.diff(tab.frame_row_offset)
    assert per_frame.min() >= 30, per_frame                # 80 particles per frame, most of them kept rows
    saved = np.bincount(tab.frame[tab.saved == 1], minlength=n)
    assert saved.min() >= min_saved, saved
    if min_saved == 0:
        assert tab.saved.sum() == 0                        # configs[3] as BASELINE.json words it: nothing above 400 px^2


def test_dense_saved_lower_save_threshold(gpu_seg_factory):
    """configs[3] particles with min_area_to_save lowered to 30: nearly every kept row is saved -- hundreds of tiny windows
    (narrower than 16 / 8 / 4 px: every copy path of the crop kernel) with neighbours inside their bounding boxes"""
    import torch
    from imageprocessing_amd import synth
    H = W = 512
    n = 12
    seg = gpu_seg_factory(H, W)
    objs, off = synth.make_objects(n, H, W, 14, **_scaled("uvp5_dense", W))
    stack = synth.render(n, H, W, 14, objs, off)
    mask = helpers.make_disk(H, W, 240)
    items = helpers.oracle_bgcorrect(stack)
    dets = helpers.oracle_detect(items, mask, min_area_to_save=30)
    labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
    tab = seg.process(torch.from_numpy(stack).cuda(), mask=torch.from_numpy(mask).cuda(), labels=labels, fetch_crops=True,
                      min_area_to_save=30.0)[3]
    helpers.compare_table(tab, dets, H, W, labels=labels.cpu().numpy())
    assert tab.saved.sum() >= 15 * n


@pytest.mark.parametrize("cfg_name", ["uvp5_dense", "uvp5_dense_saved"])
def test_dense_fullsize(cfg_name):
    """2560 x 2560, 2000 particles per frame: one frame completely against the oracle, the batch through the properties"""
    import torch
    from imageprocessing_amd import synth
    from imageprocessing_amd.engine import Segmenter
    H = W = 2560
    n = 16
    seg = Segmenter(H, W, max_frames=n, device=0)
    try:
        objs, off = synth.make_objects(n, H, W, 4, **synth.CONFIGS[cfg_name])
        frames = seg.synth_frames(n, 4, objs, off)
        mask = seg.disk_mask(1200)
        labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
        cleaned, ms, valid, tab = seg.process(frames, mask=mask, labels=labels, fetch_crops=True)
        rows = np.diff(tab.frame_row_offset)
        assert rows.min() > 800, rows                       # ~2000 particles, those below 10 px^2 dropped
        saved = np.bincount(tab.frame[tab.saved == 1], minlength=n)
        if cfg_name == "uvp5_dense_saved":
            assert saved.min() >= 450, saved
        else:
            assert saved.sum() == 0
        # every cleaned frame against the torch restatement of bg_correction.py:72-81
        want, s1, s2 = helpers.torch_bgcorrect(frames)
        assert torch.equal(cleaned, want)
        m, sd = helpers.mean_std_from_sums(s1.astype(np.float64), s2.astype(np.float64), float(H * W))
        np.testing.assert_allclose(ms[:, 0], m, rtol=1e-12)
        np.testing.assert_allclose(ms[:, 1], sd, rtol=1e-9)
        helpers.check_batch_properties(tab, labels, cleaned, mask, n_rows_sample=150, n_crops_sample=80)
        stack = frames.cpu().numpy()
        images = [(stack[i], f"img_{i:05d}.png") for i in range(n)]
        nrows, nsaved = helpers.compare_frame_with_oracle(tab, labels, cleaned, ms, 2, images, mask.cpu().numpy())
        assert nrows == rows[2] and nsaved == saved[2]
    finally:
        seg.close()


def test_density_changes_between_batches():
    """One context sees batches of very different density one after the other: what it has adapted to (run-table capacity,
    LDS size class of the band kernel, slow path, row-wise label fill: pseg_tuning_state) follows, and every batch equals the
    oracle -- the batch that finds the tables too small or its bands beyond the current class is repeated inside the call."""
    import torch
    from imageprocessing_amd.engine import Segmenter
    H, W, n = 256, 512, 7
    seg = Segmenter(H, W, max_frames=n, device=0)
    rng = np.random.default_rng(17)

    def noise(p):
        c = np.full((n, H, W), 255, np.uint8)
        for f in range(n):
            c[f][rng.random((H, W)) < p] = rng.integers(0, 240, 1, dtype=np.uint8)[0]
        return c

    def blobs():
        c = np.full((n, H, W), 255, np.uint8)
        for f in range(n):
            for _ in range(6):
                y, x, h, w = rng.integers(0, H - 40), rng.integers(0, W - 60), rng.integers(5, 40), rng.integers(5, 60)
                c[f, y:y + h, x:x + w] = rng.integers(0, 200)
        return c

    def run(cleaned):
        items = [(cleaned[f], cleaned[f], [100.0, 50.0], f"d_{f:03d}.png") for f in range(n)]
        dets = helpers.oracle_detect(items, None, min_area_to_segment=2, min_area_to_save=30, mask_img=False)
        labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
        tab = seg.detect(torch.from_numpy(cleaned).cuda(), min_area_to_segment=2, min_area_to_save=30, labels=labels, fetch_crops=True)
        helpers.compare_table(tab, dets, H, W, labels=labels.cpu().numpy())
        return seg.tuning_state()

    try:
        st = run(blobs())
        assert st["band_class"] == 1024 and not st["fill_by_rows"] and st["run_capacity"] > 0
        cap0 = st["run_capacity"]
        st = run(noise(0.02))                              # ~10 runs per row: more runs than the tables hold -> repeated; small class
        assert st["run_capacity"] > cap0 and st["band_class"] == 1024 and st["fill_by_rows"]
        st = run(noise(0.07))                              # ~33 runs per row, ~2100 per band
        assert st["band_class"] in (2048, 4096)
        st = run(noise(0.13))                              # ~58 runs per row, ~3700 per band
        assert st["band_class"] == 4096 and st["slow_path_calls"] == 0
        st = run(noise(0.5))                               # ~128 runs per row, ~8200 per band: global union-find, slow path for a while
        assert st["band_class"] == 4096 and st["slow_path_calls"] > 0
        st = run(noise(0.13))                              # still on the slow path, bands fit the large class again
        for _ in range(20):                                # calm batches: back to the small class and the span-wise fill
            st = run(blobs())
        assert st["band_class"] == 1024 and st["slow_path_calls"] == 0 and not st["fill_by_rows"]
    finally:
        seg.close()


@pytest.mark.parametrize("order", ["0", "1", "2", "3"])
def test_band_scan_order_and_its_fallback(monkeypatch, order):
    """The band kernel's look-back scan runs in block-index order by default (PSEG_BAND_ORDER=1) and in ticket order
    (=0) after a scan timed out; =2 makes the kernel raise the timeout flag itself on the first batch (totals first: the
    slow path), =3 on the context's third batch (chain and label fill already queued behind it): the call repeats the batch
    with tickets, the context stays on tickets, and every batch equals the oracle in all modes."""
    import torch
    from imageprocessing_amd.engine import Segmenter
    monkeypatch.setenv("PSEG_BAND_ORDER", order)
    H, W, n = 256, 512, 5
    rng = np.random.default_rng(23)
    seg = Segmenter(H, W, max_frames=n, device=0)

    def batch():
        c = np.full((n, H, W), 255, np.uint8)
        for f in range(n):
            for _ in range(9):
                y, x, h, w = rng.integers(0, H - 50), rng.integers(0, W - 70), rng.integers(4, 50), rng.integers(4, 70)
                c[f, y:y + h, x:x + w] = rng.integers(0, 200)
        return c

    def run(cleaned):
        items = [(cleaned[f], cleaned[f], [100.0, 50.0], f"o_{f:03d}.png") for f in range(n)]
        dets = helpers.oracle_detect(items, None, min_area_to_segment=2, min_area_to_save=30, mask_img=False)
        labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
        tab = seg.detect(torch.from_numpy(cleaned).cuda(), min_area_to_segment=2, min_area_to_save=30, labels=labels, fetch_crops=True)
        helpers.compare_table(tab, dets, H, W, labels=labels.cpu().numpy())
        return seg.tuning_state()

    try:
        st = run(batch())                                  # first batch: totals first (slow path)
        if order == "2":
            assert st["band_order"] == 0 and st["band_order_fallbacks"] == 1
        else:
            assert st["band_order"] == min(int(order), 1) and st["band_order_fallbacks"] == 0
        for _ in range(3):                                 # later batches: the chain is queued at once (fast path)
            st = run(batch())
        assert st["band_order"] == (0 if order in "23" else int(order))
        assert st["band_order_fallbacks"] == (1 if order in "23" else 0)
    finally:
        seg.close()


def test_size_scan_in_two_launches(monkeypatch):
    """The scan of the saved objects' sizes (crop offsets, row offsets, bands) takes one block, or SCAN_BLOCKS blocks in two launches
    where the previous batch saved many objects; PSEG_SCAN_MULTI_MIN=-1 sends every batch down the second way (also the first,
    and batches that save nothing or fewer objects than there are blocks): tables, crops and label maps equal the oracle."""
    import torch
    from imageprocessing_amd.engine import Segmenter
    monkeypatch.setenv("PSEG_SCAN_MULTI_MIN", "-1")
    H, W, n = 256, 512, 6
    rng = np.random.default_rng(31)
    seg = Segmenter(H, W, max_frames=n, device=0)

    def run(cleaned, save_min):
        items = [(cleaned[f], cleaned[f], [100.0, 50.0], f"z_{f:03d}.png") for f in range(n)]
        dets = helpers.oracle_detect(items, None, min_area_to_segment=2, min_area_to_save=save_min, mask_img=False)
        labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
        tab = seg.detect(torch.from_numpy(cleaned).cuda(), min_area_to_segment=2, min_area_to_save=save_min, labels=labels, fetch_crops=True)
        helpers.compare_table(tab, dets, H, W, labels=labels.cpu().numpy())
        return int(tab.saved.sum())

    try:
        empty = np.full((n, H, W), 255, np.uint8)
        assert run(empty, 30) == 0                           # nothing at all
        few = empty.copy()
        few[2, 40:80, 50:120] = 90; few[4, 100:130, 300:360] = 50
        assert run(few, 30) == 2                             # fewer objects than blocks
        many = empty.copy()
        for f in range(n):
            many[f][rng.random((H, W)) < 0.3] = 120
        assert run(many, 4) > 1500                           # every block has its chunk (also one that is not a multiple of 8)
        assert run(many, 12) > 100
    finally:
        seg.close()


def test_contexts_without_turns_scan_with_tickets(monkeypatch):
    """Index order needs the band kernel to be alone of its kind on the device, which only the per-device turn guarantees: a
    context created with PSEG_EXCLUSIVE=0 starts with tickets (and an explicit PSEG_BAND_ORDER still wins)."""
    from imageprocessing_amd.engine import Segmenter
    monkeypatch.setenv("PSEG_EXCLUSIVE", "0")
    seg = Segmenter(128, 128, max_frames=2, device=0)
    try:
        assert seg.tuning_state()["band_order"] == 0
    finally:
        seg.close()
    monkeypatch.setenv("PSEG_BAND_ORDER", "1")
    seg = Segmenter(128, 128, max_frames=2, device=0)
    try:
        assert seg.tuning_state()["band_order"] == 1
    finally:
        seg.close()
    monkeypatch.delenv("PSEG_EXCLUSIVE")
    monkeypatch.delenv("PSEG_BAND_ORDER")
    seg = Segmenter(128, 128, max_frames=2, device=0)
    try:
        assert seg.tuning_state()["band_order"] == 1
    finally:
        seg.close()
