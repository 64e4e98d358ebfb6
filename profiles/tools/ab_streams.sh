#!/bin/bash
# frames/s of the default bench for the two stream layouts and 2 - 4 contexts
mkdir -p gpurun_out
for sh in 0 1; do for k in 2 3 4; do
  PSEG_SHARED_STREAM=$sh timeout -k 10 200 python bench.py --cpu off --host-batches 0 --streams $k > gpurun_out/ab_s${sh}_k$k.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/ab_s${sh}_k$k.json')); g=d['roofline_groups']; print('shared $sh contexts $k', d['value'], d['ms_per_step'], 'bg', g['bg_correct']['avg_ms'], 'fill', g['label_map']['avg_ms'], 'crops', g['crops_props']['avg_ms'])"
done; done
