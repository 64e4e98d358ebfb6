#!/usr/bin/env python3
"""Rate of pseg_vit_tensors (classifier inputs from the packed device crops) on one 200-frame batch of the
bench workload: crops per second for chunks of 512 crops, HIP-synchronised wall time."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from imageprocessing_amd import synth
from imageprocessing_amd.engine import Segmenter

H = W = 2560
BATCH = 200
seg = Segmenter(H, W, max_frames=BATCH, device=0)
mask = seg.disk_mask(1200)
objs, off = synth.make_objects(BATCH, H, W, 2, frame0=0, **synth.CONFIGS["pisco"])
frames = seg.synth_frames(BATCH, 2, objs, off, frame0=0)
cleaned = torch.empty_like(frames)
_, _, _, tab = seg.process(frames, mask=mask, cleaned=cleaned)
ns = len(tab.saved_row)
chunk = 512
out = torch.empty((chunk, 3, 224, 224), dtype=torch.float32, device="cuda:0")
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    for first in range(0, ns - chunk + 1, chunk):
        seg.vit_tensors(first=first, count=chunk, out=out)
        done += chunk
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {done} crops of {ns} in {dt * 1e3:.2f} ms = {done / dt:,.0f} crops/s "
          f"({done * 3 * 224 * 224 * 4 / dt / 1e9:.0f} GB/s of float32 output)")
seg.close()
