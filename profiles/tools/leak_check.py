"""Device and host memory of a long-running context: 3000 calls on batches of changing density (tables grow, size classes switch,
the slow path is taken now and then); free device memory (hipMemGetInfo through torch) and the process's resident set are printed
every 500 calls -- both must level off after the first rounds."""
import sys, resource, torch, numpy as np
sys.path.insert(0, '.')
from imageprocessing_amd.engine import Segmenter
H, W, n = 512, 512, 24
seg = Segmenter(H, W, max_frames=n, device=0)
rng = np.random.default_rng(5)
mask = seg.disk_mask(250)
batches = []
for p in (0.0005, 0.004, 0.02, 0.08, 0.3, 0.002):
    st = np.full((n, H, W), 200, np.uint8)
    st += rng.integers(0, 20, st.shape, dtype=np.uint8)
    for f in range(n):
        st[f][rng.random((H, W)) < p] = 40
    batches.append(torch.from_numpy(st).cuda())
cleaned = torch.empty((n, H, W), dtype=torch.uint8, device='cuda')
labels = torch.empty((n, H, W), dtype=torch.int32, device='cuda')
rows = 0
for i in range(3001):
    c, ms, v, tab = seg.process(batches[(i * 7 + i // 50) % len(batches)], mask=mask, cleaned=cleaned, labels=labels, min_area_to_segment=2, min_area_to_save=20,
                                fetch_crops="pinned" if i % 3 == 0 else False)
    rows += tab.n_rows
    if i % 500 == 0:
        free, total = torch.cuda.mem_get_info()
        print(f"call {i:5d}: device memory in use {(total - free) / 2**20:9.1f} MiB, host resident set {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:8.1f} MiB, rows so far {rows}, state {seg.tuning_state()}", flush=True)
seg.close()
