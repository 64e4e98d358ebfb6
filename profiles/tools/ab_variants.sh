#!/bin/bash
# builds of the library with other compile-time parameters (variants/libpiscoseg_<name>.so, loaded through PSEG_LIB):
# a correctness pass over the crop tests, then the crop kernels' device time and the bench on two workloads
set -e
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset PSEG_LIB; else export PSEG_LIB=$GRAFT_REPO_ROOT/variants/libpiscoseg_$v.so; fi
  timeout -k 10 200 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dense.py -x -q --timeout 120 --timeout-method thread > gpurun_out/abv_$v.test.log 2>&1 || { echo "$v: TESTS FAILED"; tail -5 gpurun_out/abv_$v.test.log; exit 1; }
  for wl in pisco uvp5_dense_saved; do
    timeout -k 10 150 python3 bench.py --cpu off --host-batches 0 --workload $wl --steps 3 --warmup 1 > gpurun_out/abv_${v}_$wl.json 2> gpurun_out/abv_${v}_$wl.err
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/abv_${v}_$wl.json').read().strip().splitlines()[-1]); print('$v', '$wl', d['value'], d['ms_per_step'])"
    timeout -k 10 250 bash profiles/tools/kstats.sh abv_${v}_$wl --workload $wl | grep -i "crops_sums\|k_convex\|band_map"
  done
done
