# A/B of environment settings on one box: r3_ab.sh "CFG1 CFG2 ..." [workloads...]; CFG = VAR=v[,VAR=v...] (or "-" for none)
# workload = name:bench_args with _ for spaces (e.g. nolabels:--no-labels densesaved:--workload_uvp5_dense_saved)
# AB_TESTS=1 runs the parity tests first; AB_REPS (default 2) repetitions
set -e
cd $GRAFT_REPO_ROOT
CFGS=$1; shift
WL=${@:-"default:"}
if [ -n "$AB_TESTS" ]; then
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dense.py tests/test_gpu_async.py tests/test_gpu_dropin.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/ab_tests.log 2>&1 || { tail -40 gpurun_out/ab_tests.log; exit 1; }
tail -1 gpurun_out/ab_tests.log
fi
for rep in $(seq 1 ${AB_REPS:-2}); do
for cfg in $CFGS; do
for v in $WL; do
  name=${v%%:*}; args=$(echo ${v#*:} | tr _ ' ' | sed 's/uvp5 dense saved/uvp5_dense_saved/; s/uvp5 dense/uvp5_dense/; s/no labels/no-labels/; s/deconv standin/deconv-standin/')
  envs=$(echo $cfg | tr , ' '); [ "$cfg" = "-" ] && envs="AB_NONE=1"
  tag=$(echo $cfg | tr -c 'A-Za-z0-9=\n' _)
  env $envs timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu off --host-batches 0 $args > gpurun_out/ab_${tag}_$name.json 2> gpurun_out/ab_${tag}_$name.err
  python - gpurun_out/ab_${tag}_$name.json "$cfg $name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
g=d["roofline_groups"]
print(f"{sys.argv[2]:44s} {d['value']:9.0f} f/s  {d['ms_per_step']:7.3f} ms/step  path {d['path']['frac_of_hbm_peak_per_gpu']:.3f} def {d['path'].get('fills_deferred')}/{d['path'].get('deferred_fills_given_up')}", {k[:6]:(v["avg_ms"],v["avg_ms_alone"]) for k,v in g.items()})
PY
done
done
done
