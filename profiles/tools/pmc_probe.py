#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a few fused steps on one context, nothing else.
Run as  rocprofv3 --pmc FETCH_SIZE -d out -- python3 profiles/tools/pmc_probe.py   (one counter group per run)."""
import os
import sys

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
labels = torch.empty((BATCH, H, W), dtype=torch.int32, device="cuda:0")
for _ in range(3):
    seg.process(frames, mask=mask, cleaned=cleaned, labels=labels)
torch.cuda.synchronize()
seg.close()
