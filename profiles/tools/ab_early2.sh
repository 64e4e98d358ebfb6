#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "0 512" "15 32" "15 64" "25 64" "25 128" "35 128" "35 256" "0 512"; do
  set -- $cfg
  PSEG_EARLY_FILL=$1 PSEG_ZERO_BLOCKS=$2 timeout -k 10 150 python3 bench.py --cpu off --host-batches 0 --steps 10 --warmup 2 > gpurun_out/abe2.json 2> gpurun_out/abe2.err || { echo "bench $cfg failed"; tail -5 gpurun_out/abe2.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/abe2.json').read().strip().splitlines()[-1]); print('early=$1 blocks=$2', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['roofline_groups'].items()})"
done
