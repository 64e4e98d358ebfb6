#!/bin/bash
# timing events in the stream (PSEG_TIMING=1, default) against none: bench value, alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for t in 1 0; do
  PSEG_TIMING=$t timeout -k 10 120 python3 bench.py --cpu off --host-batches 0 --steps 20 --warmup 2 > gpurun_out/abt_${t}_$i.json 2> gpurun_out/abt_${t}_$i.err || exit 1
  python3 -c "import json; d=json.loads(open('gpurun_out/abt_${t}_$i.json').read().strip().splitlines()[-1]); print('timing=$t', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['roofline_groups'].items()})"
done; done
