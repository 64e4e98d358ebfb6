#!/bin/bash
# early zero fill of the label map's leading frames (PSEG_EARLY_FILL=<frames>): tests with it on, then the bench for several values
cd $GRAFT_REPO_ROOT
PSEG_EARLY_FILL=3 timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dense.py tests/test_gpu_async.py tests/test_gpu_fullsize.py -x -q --timeout 200 --timeout-method thread > gpurun_out/abe_tests.log 2>&1 || { echo "TESTS FAILED"; tail -20 gpurun_out/abe_tests.log; exit 1; }
tail -1 gpurun_out/abe_tests.log
for x in "$@"; do
  PSEG_EARLY_FILL=$x timeout -k 10 150 python3 bench.py --cpu off --host-batches 0 --steps 10 --warmup 2 > gpurun_out/abe_$x.json 2> gpurun_out/abe_$x.err || { echo "bench $x failed"; tail -5 gpurun_out/abe_$x.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/abe_$x.json').read().strip().splitlines()[-1]); print('early=$x', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['roofline_groups'].items()}, d['path']['results_checked_after_timed_region'])"
done
