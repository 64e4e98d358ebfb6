import time, sys, torch, numpy as np
sys.path.insert(0, '.')
from imageprocessing_amd.engine import Segmenter
from imageprocessing_amd import synth
H=W=2560; n=200
seg=Segmenter(H,W,max_frames=n,device=0)
stack=torch.empty((n,H,W),dtype=torch.uint8,device='cuda')
objs,off=synth.make_objects(n,H,W,2,frame0=0,**synth.CONFIGS['pisco'])
seg.synth_frames(n,2,objs,off,frame0=0,out=stack)
mask=seg.disk_mask(1200)
cleaned=torch.empty_like(stack); labels=torch.empty((n,H,W),dtype=torch.int32,device='cuda')
for i in range(8):
    t0=time.perf_counter()
    c,ms,v,tab=seg.process(stack,mask=mask,cleaned=cleaned,labels=labels)
    t1=time.perf_counter()
    print("call %.0f us, device whole call %.0f us, groups %s" % ((t1-t0)*1e6, seg.last_ms(5)*1e3, [round(seg.last_ms(k)*1e3) for k in range(5)]), file=sys.stderr)
seg.close()
