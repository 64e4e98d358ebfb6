# where the table kernels run (PSEG_TABLE_MODE 0 before / 1 beside / 2 behind the label fill): bench lines per mode
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dense.py tests/test_gpu_async.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/tm_tests.log 2>&1 || { tail -40 gpurun_out/tm_tests.log; exit 1; }
tail -2 gpurun_out/tm_tests.log
for mode in 0 1 2 1 0; do
for v in "default:" "densesaved:--workload uvp5_dense_saved"; do
  name=${v%%:*}; args=${v#*:}
  PSEG_TABLE_MODE=$mode timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu off --host-batches 0 $args > gpurun_out/tm${mode}_$name.json 2> gpurun_out/tm${mode}_$name.err
  python - gpurun_out/tm${mode}_$name.json "mode$mode $name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
g=d["roofline_groups"]
print(f"{sys.argv[2]:18s} {d['value']:9.0f} f/s  {d['ms_per_step']:7.3f} ms/step  path {d['path']['frac_of_hbm_peak_per_gpu']:.3f} ", {k[:6]:(v["avg_ms"],v["avg_ms_alone"]) for k,v in g.items()})
PY
done
done
