#!/bin/bash
# crop kernel variants (variants/libpiscoseg_<name>.so through PSEG_LIB; base = the in-tree build): crop tests, then kernel statistics
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset PSEG_LIB; else export PSEG_LIB=$GRAFT_REPO_ROOT/variants/libpiscoseg_$v.so; fi
  timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dense.py -x -q --timeout 120 --timeout-method thread > gpurun_out/abc_$v.test.log 2>&1 || { echo "$v: TESTS FAILED"; tail -15 gpurun_out/abc_$v.test.log; exit 1; }
  tail -1 gpurun_out/abc_$v.test.log
  for wl in uvp5_dense_saved pisco; do
    echo "== $v $wl"
    timeout -k 10 250 bash profiles/tools/kstats.sh abc_${v}_$wl --workload $wl | grep -i "crops_sums\|k_convex\|band_map" || exit 1
  done
done
