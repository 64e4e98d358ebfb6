#!/bin/bash
# A/B of the two label-fill kernels (PSEG_FILL_ROWS=0/1) on the headline and the dense workload
set -e
cd $GRAFT_REPO_ROOT
for wl in pisco uvp5_dense_saved; do
  for fr in 0 1; do
    export PSEG_FILL_ROWS=$fr
    timeout -k 10 150 python3 bench.py --cpu off --host-batches 0 --workload $wl --steps 3 --warmup 1 > gpurun_out/ab_fill_${wl}_$fr.json 2> gpurun_out/ab_fill_${wl}_$fr.err
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_fill_${wl}_$fr.json').read().strip().splitlines()[-1]); print('$wl', 'rows=$fr', d['value'], d['ms_per_step'])"
    timeout -k 10 250 bash profiles/tools/kstats.sh abf_${wl}_$fr --workload $wl | grep -i "label_fill"
  done
done
