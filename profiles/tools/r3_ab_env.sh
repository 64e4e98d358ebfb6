# A/B of one environment switch on one box: r3_ab_env.sh VAR "v1 v2 ..." [workloads...]; runs the parity tests first
# (workload = name:bench_args with _ for spaces, e.g. nolabels:--no-labels densesaved:--workload_uvp5_dense_saved)
set -e
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; shift 2
WL=${@:-"default: densesaved:--workload_uvp5_dense_saved"}
if [ -z "$AB_SKIP_TESTS" ]; then
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dense.py tests/test_gpu_async.py tests/test_gpu_dropin.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/ab_tests.log 2>&1 || { tail -40 gpurun_out/ab_tests.log; exit 1; }
tail -1 gpurun_out/ab_tests.log
fi
for rep in 1 2; do
for val in $VALS; do
for v in $WL; do
  name=${v%%:*}; args=$(echo ${v#*:} | tr _ ' ' | sed 's/uvp5 dense saved/uvp5_dense_saved/; s/uvp5 dense/uvp5_dense/; s/no labels/no-labels/')
  env $VAR=$val timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu off --host-batches 0 $args > gpurun_out/ab_${val}_$name.json 2> gpurun_out/ab_${val}_$name.err
  python - gpurun_out/ab_${val}_$name.json "$VAR=$val $name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
g=d["roofline_groups"]
print(f"{sys.argv[2]:32s} {d['value']:9.0f} f/s  {d['ms_per_step']:7.3f} ms/step  path {d['path']['frac_of_hbm_peak_per_gpu']:.3f} ", {k[:6]:(v["avg_ms"],v["avg_ms_alone"]) for k,v in g.items()})
PY
done
done
done
