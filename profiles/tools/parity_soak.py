"""One-off parity soak (not part of the test suite): seeded structured frames (tests/test_gpu_parity._structured_frame) and noise
frames at many geometries, every batch through pseg_detect and against the contour-tracing oracle with tests/helpers.compare_table
(bit-exact table, crops, label map; float columns to 1e-5).  usage: parity_soak.py <first seed> <number of seeds>
Mode "parts": parity_soak.py parts <first seed> <number of seeds> -- stacks of 30 - 520 frames cut into frame ranges for 2 - 8 ranks
(sharding.plan_ranges, pseg_process_part with forward halos) against the same stacks in whole reference batches (200 frames, the
last one 200 - 399): cleaned frames, label maps, statistics and merged tables identical.
Mode "bg": parity_soak.py bg <first seed> <number of seeds> -- the fused call (background correction + detection) on random stacks
of 12 - 90 frames (smooth fields + noise + a few flat frames that fail the std > 2 gate), random time chunks and pretended block
slots, against oracle correct_img / detect_on_img: cleaned frames bit for bit, raw and masked mean / std, validity, full tables.
Mode "pipe": parity_soak.py pipe <first seed> <number of seeds> -- the same kind of stacks through three contexts in flight
(round 3: deferred label fills on the device's streaming queue), every batch against the oracle."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from tests import helpers
from tests.test_gpu_parity import _structured_frame
from imageprocessing_amd.engine import Segmenter

def soak_bg(first, count):
    import os
    t0 = time.time()
    frames = rows = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(91000 + seed)
        H, W = [(64, 128), (96, 192), (128, 256), (80, 64), (200, 320)][seed % 5]
        n = int(rng.integers(12, 91))
        os.environ["PSEG_BG_CHUNKS"] = str(int(rng.integers(1, 5)))
        if seed % 3 == 0:
            os.environ["PSEG_BG_TAIL"] = str(int(rng.integers(2, 9)))
        else:
            os.environ.pop("PSEG_BG_TAIL", None)
        base = rng.integers(120, 230)
        yy, xx = np.mgrid[0:H, 0:W]
        stack = np.empty((n, H, W), np.uint8)
        for f in range(n):
            field = base + 20 * np.sin(xx / 37.0 + f / 5.0) + 10 * np.cos(yy / 23.0)
            img = field + rng.normal(0, 6, (H, W))
            for _ in range(int(rng.integers(0, 5))):          # dark particles
                cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(2, 14)
                img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] -= rng.integers(40, 120)
            stack[f] = np.clip(img, 0, 255).astype(np.uint8)
        for f in rng.integers(0, n, int(rng.integers(0, 3))):
            stack[f] = rng.integers(0, 256)                   # flat frame: std = 0 -> invalid
        mask = helpers.make_disk(H, W, min(H, W) // 2 - 3)
        items = helpers.oracle_bgcorrect(stack)
        dets = helpers.oracle_detect(items, mask, min_area_to_segment=3, min_area_to_save=30)
        seg = Segmenter(H, W, max_frames=n, device=0)
        try:
            labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
            cleaned, ms, valid, tab = seg.process(torch.from_numpy(stack).cuda(), mask=torch.from_numpy(mask).cuda(), labels=labels,
                                                  min_area_to_segment=3, min_area_to_save=30, fetch_crops=True)
            cl = cleaned.cpu().numpy()
            for i, it in enumerate(items):
                np.testing.assert_allclose(ms[i], it[2], rtol=1e-9, atol=1e-9)
                assert valid[i] == (0 if isinstance(it[1], list) else 1)
                if valid[i]:
                    assert np.array_equal(cl[i], it[1]), f"seed {seed} frame {i}"
            helpers.compare_table(tab, dets, H, W, labels=labels.cpu().numpy())
        finally:
            seg.close()
        frames += n; rows += tab.n_rows
        if (seed - first) % 20 == 19:
            print(f"bg seed {seed}: {frames} frames, {rows} rows equal to the oracle, {time.time() - t0:.0f} s", flush=True)
    print(f"parity soak (background correction + detection): seeds {first}..{first + count - 1}: {frames} frames, {rows} rows -- all equal to the oracle ({time.time() - t0:.0f} s)")


def soak_parts(first, count):
    """frame-range sharding (sharding.plan_ranges + pseg_process_part with forward halos) against whole reference batches"""
    from imageprocessing_amd import sharding
    t0 = time.time()
    frames = nparts = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(123000 + seed)
        H, W = [(64, 128), (96, 192), (80, 64)][seed % 3]
        n = int(rng.integers(30, 521))
        yy, xx = np.mgrid[0:H, 0:W]
        stack = np.empty((n, H, W), np.uint8)
        base = rng.integers(120, 230)
        for f in range(n):
            img = base + 20 * np.sin(xx / 37.0 + f / 5.0) + 10 * np.cos(yy / 23.0) + rng.normal(0, 6, (H, W))
            for _ in range(int(rng.integers(0, 5))):
                cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(2, 14)
                img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] -= rng.integers(40, 120)
            stack[f] = np.clip(img, 0, 255).astype(np.uint8)
        for f in rng.integers(0, n, int(rng.integers(0, 4))):
            stack[f] = rng.integers(0, 256)
        mask = torch.from_numpy(helpers.make_disk(H, W, min(H, W) // 2 - 3)).cuda()
        dev = torch.from_numpy(stack).cuda()
        seg = Segmenter(H, W, max_frames=420, device=0)
        try:
            cleaned = torch.empty((n, H, W), dtype=torch.uint8, device="cuda")
            labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
            ms = np.zeros((n, 2)); valid = np.zeros(n, np.uint8)
            blobs = []
            for a, m in sharding.batches_of(n):
                c, q, v, tab = seg.process(dev[a:a + m], mask=mask, labels=labels[a:a + m], cleaned=cleaned[a:a + m], min_area_to_segment=3, min_area_to_save=30)
                ms[a:a + m] = q; valid[a:a + m] = v
                blobs.append(sharding.pack_table(tab, frame_base=a))
            ref = sharding.merge_tables([sharding.unpack_table(b) for b in blobs])
            world = int(rng.integers(2, 9))
            blobs = []
            for parts in sharding.plan_ranges(n, world):
                for p in parts:
                    s_, ne, na = p["start"], p["n_emit"], p["n_avail"]
                    lab = torch.empty((ne, H, W), dtype=torch.int32, device="cuda")
                    c, q, v, tab = seg.process(dev[s_:s_ + na], mask=mask, labels=lab, min_area_to_segment=3, min_area_to_save=30, n_emit=None if ne == na else ne)
                    ok = torch.from_numpy(v.astype(bool)).cuda()
                    assert torch.equal(c[ok], cleaned[s_:s_ + ne][ok]), f"seed {seed} part at {s_}: cleaned"
                    assert torch.equal(lab, labels[s_:s_ + ne]), f"seed {seed} part at {s_}: labels"
                    np.testing.assert_array_equal(q, ms[s_:s_ + ne]); np.testing.assert_array_equal(v, valid[s_:s_ + ne])
                    blobs.append(sharding.pack_table(tab, frame_base=s_))
                    nparts += 1
            got = sharding.merge_tables([sharding.unpack_table(b) for b in blobs])
            for k in ref:
                if k != "shard_of_saved":
                    np.testing.assert_array_equal(got[k], ref[k], err_msg=f"seed {seed} world {world}: {k}")
        finally:
            seg.close()
        frames += n
        if (seed - first) % 20 == 19:
            print(f"parts seed {seed}: {frames} frames in {nparts} parts equal to the whole batches, {time.time() - t0:.0f} s", flush=True)
    print(f"parity soak (frame ranges with halos against whole batches): seeds {first}..{first + count - 1}: {frames} frames, {nparts} parts over 2 - 8 ranks -- all equal ({time.time() - t0:.0f} s)")


def soak_pipe(first, count):
    """the fused call through THREE CONTEXTS IN FLIGHT (pseg_submit / pseg_wait: deferred label fills, the streaming queue,
    refreshed tables -- DESIGN.md section 3, "Schedule of one device") on random stacks, every batch against the oracle"""
    t0 = time.time()
    frames = rows = deferred = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(555000 + seed)
        H, W = [(64, 128), (96, 192), (128, 256), (200, 320)][seed % 4]
        nb = 7
        n = int(rng.integers(14, 60))
        yy, xx = np.mgrid[0:H, 0:W]
        mask = helpers.make_disk(H, W, min(H, W) // 2 - 3)
        stacks = []
        for b in range(nb):
            base = rng.integers(120, 230)
            stack = np.empty((n, H, W), np.uint8)
            for f in range(n):
                img = base + 20 * np.sin(xx / 37.0 + f / 5.0) + 10 * np.cos(yy / 23.0) + rng.normal(0, 6, (H, W))
                for _ in range(int(rng.integers(0, 6))):
                    cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(2, 16)
                    img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] -= rng.integers(40, 120)
                stack[f] = np.clip(img, 0, 255).astype(np.uint8)
            for f in rng.integers(0, n, int(rng.integers(0, 3))):
                stack[f] = rng.integers(0, 256)
            stacks.append(stack)
        segs = [Segmenter(H, W, max_frames=n, device=0) for _ in range(3)]
        try:
            dmask = torch.from_numpy(mask).cuda()
            dev = [torch.from_numpy(st).cuda() for st in stacks]
            labs = [torch.empty((n, H, W), dtype=torch.int32, device="cuda") for _ in segs]
            got = [None] * nb
            inflight = []
            def take():
                b0, k0 = inflight.pop(0)
                c, ms, v, tab = segs[k0].wait(fetch_crops=True)
                got[b0] = (c.cpu().numpy(), ms.copy(), v.copy(), tab, labs[k0].cpu().numpy())
            for b in range(nb):
                k = b % 3
                if len(inflight) == 3:
                    take()
                segs[k].submit(dev[b], mask=dmask, labels=labs[k], min_area_to_segment=3, min_area_to_save=30)
                inflight.append((b, k))
            while inflight:
                take()
            deferred += sum(s_.tuning_state()["fills_deferred"] for s_ in segs)
            for b in range(nb):
                items = helpers.oracle_bgcorrect(stacks[b])
                dets = helpers.oracle_detect(items, mask, min_area_to_segment=3, min_area_to_save=30)
                cl, ms, valid, tab, lab = got[b]
                for i, it in enumerate(items):
                    np.testing.assert_allclose(ms[i], it[2], rtol=1e-9, atol=1e-9)
                    assert valid[i] == (0 if isinstance(it[1], list) else 1)
                    if valid[i]:
                        assert np.array_equal(cl[i], it[1]), f"seed {seed} batch {b} frame {i}"
                helpers.compare_table(tab, dets, H, W, labels=lab)
                frames += n; rows += tab.n_rows
        finally:
            for s_ in segs:
                s_.close()
        if (seed - first) % 10 == 9:
            print(f"pipe seed {seed}: {frames} frames, {rows} rows equal to the oracle, {deferred} fills deferred, {time.time() - t0:.0f} s", flush=True)
    print(f"parity soak (three contexts in flight, deferred fills): seeds {first}..{first + count - 1}: {frames} frames, {rows} rows, "
          f"{deferred} deferred fills -- all equal to the oracle ({time.time() - t0:.0f} s)")


if sys.argv[1] == "pipe":
    soak_pipe(int(sys.argv[2]), int(sys.argv[3]))
    sys.exit(0)
if sys.argv[1] == "parts":
    soak_parts(int(sys.argv[2]), int(sys.argv[3]))
    sys.exit(0)
if sys.argv[1] == "bg":
    soak_bg(int(sys.argv[2]), int(sys.argv[3]))
    sys.exit(0)
first, count = int(sys.argv[1]), int(sys.argv[2])
sizes = [(64, 64), (100, 128), (191, 256), (256, 512), (320, 192), (65, 1024), (512, 512), (130, 2560)]
t0 = time.time()
frames = objects = saved = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(77000 + seed)
    H, W = sizes[seed % len(sizes)]
    n = 8
    cleaned = np.full((n, H, W), 255, np.uint8)
    for f in range(n):
        if f % 4 == 3:
            fg = rng.random((H, W)) < rng.choice([0.05, 0.3, 0.55, 0.8])
        else:
            fg = _structured_frame(rng, H, W)
        vals = rng.integers(0, 244, size=(H, W), dtype=np.uint8)
        cleaned[f][fg] = vals[fg]
    seg_min, save_min = [(2, 12), (10, 400), (1, 3), (5, 50)][seed % 4]
    items = [(cleaned[f], cleaned[f], [100.0, 50.0], f"p_{f:03d}.png") for f in range(n)]
    dets = helpers.oracle_detect(items, None, min_area_to_segment=seg_min, min_area_to_save=save_min, mask_img=False)
    seg = Segmenter(H, W, max_frames=n, device=0)
    try:
        labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
        tab = seg.detect(torch.from_numpy(cleaned).cuda(), min_area_to_segment=seg_min, min_area_to_save=save_min, labels=labels, fetch_crops=True)
        helpers.compare_table(tab, dets, H, W, labels=labels.cpu().numpy())
    finally:
        seg.close()
    frames += n; objects += tab.n_rows; saved += int(tab.saved.sum())
    if (seed - first) % 20 == 19:
        print(f"seed {seed}: {frames} frames, {objects} rows, {saved} saved objects equal to the oracle, {time.time() - t0:.0f} s", flush=True)
print(f"parity soak: seeds {first}..{first + count - 1}: {frames} frames, {objects} rows, {saved} saved objects -- all equal to the oracle ({time.time() - t0:.0f} s)")
