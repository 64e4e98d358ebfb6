#!/bin/bash
# the per-device turn (PSEG_EXCLUSIVE=1, default) against free-running contexts, by workload and number of contexts
cd $GRAFT_REPO_ROOT
for wl in "$@"; do for ex in 1 0; do for st in 3 4 6; do
  PSEG_EXCLUSIVE=$ex timeout -k 10 200 python3 bench.py --cpu off --host-batches 0 --steps 10 --workload $wl --streams $st > gpurun_out/abx.json 2>/dev/null || { echo "failed $wl $ex $st"; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/abx.json').read().strip().splitlines()[-1]); print('$wl exclusive=$ex contexts=$st', d['value'], d['ms_per_step'], d['path']['band_scan_fallbacks'], d['path']['results_checked_after_timed_region'])"
done; done; done
