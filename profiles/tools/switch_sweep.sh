#!/bin/bash
# the parity tests under every documented switch of the library: one pytest process per setting
# (without the two tests that assert what a context adapts to on the default path: pseg_tuning_state)
cd $GRAFT_REPO_ROOT
rc=0
for sw in PSEG_DEFER_FILL=0 PSEG_STREAM_QUEUE=0 PSEG_FILL_REFRESH=1 PSEG_FILL_REFRESH=0 PSEG_TABLE_MODE=0 PSEG_TABLE_MODE=2 PSEG_CONVEX_SMALL=0 PSEG_RUNS_OLD=1 PSEG_NO_BANDS=1 PSEG_FILL_ROWS=1 PSEG_FILL_ROWS=0 PSEG_SHARED_STREAM=1 PSEG_EXCLUSIVE=0 PSEG_BG_GENERIC=1 PSEG_BG_CHUNKS=3 PSEG_BAND_ORDER=0 PSEG_TIMING=0 PSEG_CROP_CLASSES=0 PSEG_SCAN_MULTI_MIN=-1; do
  env $sw timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dense.py tests/test_gpu_async.py -k "not density_changes and not band_scan_order" -x -q --timeout 120 --timeout-method thread > gpurun_out/sweep_$sw.log 2>&1
  r=$?
  echo "$sw: rc=$r $(tail -1 gpurun_out/sweep_$sw.log)"
  if [ $r -ne 0 ]; then rc=1; tail -20 gpurun_out/sweep_$sw.log; break; fi
done
exit $rc
