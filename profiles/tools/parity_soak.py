"""One-off parity soak (not part of the test suite): seeded structured frames (tests/test_gpu_parity._structured_frame) and noise
frames at many geometries, every batch through pseg_detect and against the contour-tracing oracle with tests/helpers.compare_table
(bit-exact table, crops, label map; float columns to 1e-5).  usage: parity_soak.py <first seed> <number of seeds>"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from tests import helpers
from tests.test_gpu_parity import _structured_frame
from imageprocessing_amd.engine import Segmenter

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
