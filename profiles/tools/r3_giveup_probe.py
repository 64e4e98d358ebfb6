import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from imageprocessing_amd import synth
from imageprocessing_amd.engine import Segmenter
H = W = 2560; B = 200; NS = 3; NB = 10
segs = [Segmenter(H, W, max_frames=B, device=0) for _ in range(NS)]
mask = segs[0].disk_mask(1200)
stack = []
for b in range(NB):
    objs, off = synth.make_objects(B, H, W, 2, frame0=b * B, **synth.CONFIGS["pisco"])
    stack.append(segs[0].synth_frames(B, 2, objs, off, frame0=b * B))
cleaned = [torch.empty((B, H, W), dtype=torch.uint8, device="cuda") for _ in range(NS)]
labels = [torch.empty((B, H, W), dtype=torch.int32, device="cuda") for _ in range(NS)]
torch.cuda.synchronize()
total = 120
inflight = []
log = []
t0 = time.perf_counter()
prev = 0
for i in range(total):
    k = i % NS
    if len(inflight) == NS:
        i0, k0 = inflight.pop(0)
        segs[k0].wait()
        tw = time.perf_counter()
    segs[k].submit(stack[i % NB], mask=mask, cleaned=cleaned[k], labels=labels[k])
    ts = time.perf_counter()
    inflight.append((i, k))
    g = sum(s.tuning_state()["deferred_fills_given_up"] for s in segs)
    d = sum(s.tuning_state()["fills_deferred"] for s in segs)
    if g != prev:
        log.append((i, g, d, round((ts - t0) * 1e3, 2)))
        prev = g
for i0, k0 in inflight:
    segs[k0].wait()
print("total ms", round((time.perf_counter() - t0) * 1e3, 1), "give-ups noticed at (call, give-ups, deferred, ms):", log)
for s in segs: s.close()
