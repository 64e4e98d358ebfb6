# full GPU test suite, then default / one-context / dense bench lines: r3_test_bench.sh <prefix under gpurun_out/>
set -e
cd $GRAFT_REPO_ROOT
P=${1:-tb}
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${P}_tests.log 2>&1 || { tail -40 gpurun_out/${P}_tests.log; exit 1; }
tail -2 gpurun_out/${P}_tests.log
for v in "default:" "one:--streams 1" "nolabels:--no-labels" "dense:--workload uvp5_dense" "densesaved:--workload uvp5_dense_saved"; do
  name=${v%%:*}; args=${v#*:}
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu off --host-batches 0 $args > gpurun_out/${P}_$name.json 2> gpurun_out/${P}_$name.err
  python - gpurun_out/${P}_$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
g=d["roofline_groups"]
print(f"{sys.argv[2]:10s} {d['value']:9.0f} f/s  {d['ms_per_step']:7.3f} ms/step  path {d['path']['frac_of_hbm_peak_per_gpu']:.3f} ", {k:(v["avg_ms"],v["avg_ms_alone"]) for k,v in g.items()})
PY
done
