#!/bin/bash
# the band kernel under test: parity + dense tests, then bench and per-kernel statistics on two workloads
# ab_band.sh <tag> [env assignments...]
set -e
cd $GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dense.py tests/test_gpu_async.py -x -q --timeout 120 --timeout-method thread > gpurun_out/abb_$tag.test.log 2>&1 || { echo "$tag: TESTS FAILED"; tail -15 gpurun_out/abb_$tag.test.log; exit 1; }
tail -1 gpurun_out/abb_$tag.test.log
for wl in pisco uvp5_dense_saved; do
  timeout -k 10 150 python3 bench.py --cpu off --host-batches 0 --workload $wl --steps 5 --warmup 1 > gpurun_out/abb_${tag}_$wl.json 2> gpurun_out/abb_${tag}_$wl.err
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/abb_${tag}_$wl.json').read().strip().splitlines()[-1]); print('$tag', '$wl', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['roofline_groups'].items()})"
  timeout -k 10 250 bash profiles/tools/kstats.sh abb_${tag}_$wl --workload $wl | grep -i "runs_band\|unions_rows\|label_fill"
done
