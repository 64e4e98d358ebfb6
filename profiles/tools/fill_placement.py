"""Does the label fill's time depend on where its map lies?  One process, one context: the map is allocated again and again behind
dummy allocations of different sizes (torch's caching allocator emptied in between), five calls each, label-fill time per call."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from imageprocessing_amd.engine import Segmenter
from imageprocessing_amd import synth
H = W = 2560; n = 200
seg = Segmenter(H, W, max_frames=n, device=0)
stack = torch.empty((n, H, W), dtype=torch.uint8, device='cuda')
objs, off = synth.make_objects(n, H, W, 2, frame0=0, **synth.CONFIGS['pisco'])
seg.synth_frames(n, 2, objs, off, frame0=0, out=stack)
mask = seg.disk_mask(1200)
cleaned = torch.empty_like(stack)
for trial, pad_mb in enumerate([0, 1, 3, 64, 511, 1024, 2049, 0, 7, 4096]):
    torch.cuda.empty_cache()
    dummy = torch.empty(pad_mb * (1 << 20) + 256 * (trial % 3), dtype=torch.uint8, device='cuda') if pad_mb else None
    labels = torch.empty((n, H, W), dtype=torch.int32, device='cuda')
    t = []
    for i in range(6):
        seg.process(stack, mask=mask, cleaned=cleaned, labels=labels)
        t.append(seg.last_ms(3))
    print(f"pad {pad_mb:5d} MB  labels at {labels.data_ptr():#x} (mod 2 MiB {labels.data_ptr() % (1 << 21):#x})  fill ms {[round(x, 3) for x in t[1:]]}", file=sys.stderr)
    del labels, dummy
seg.close()
