#!/bin/bash
# the default bench on the headline and the dense workload: bench_two.sh <tag> [bench args]
cd $GRAFT_REPO_ROOT
tag=$1; shift
for wl in pisco uvp5_dense_saved; do
  timeout -k 10 200 python3 bench.py --cpu off --host-batches 0 --workload $wl --steps 10 --warmup 2 "$@" > gpurun_out/b2_${tag}_$wl.json 2> gpurun_out/b2_${tag}_$wl.err || { echo "$wl failed"; tail -5 gpurun_out/b2_${tag}_$wl.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/b2_${tag}_$wl.json').read().strip().splitlines()[-1]); print('$tag', '$wl', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['roofline_groups'].items()}, d['path']['results_checked_after_timed_region'])"
done
