# bench lines of the variants under the current defaults and with the round's scheduling switched off: r3_variants.sh <prefix>
set -e
cd $GRAFT_REPO_ROOT
P=${1:-var}
for cfg in "-" "PSEG_DEFER_FILL=0"; do
for v in "default:" "two:--streams 2" "one:--streams 1" "four:--streams 4" "threads:--driver threads" "nolabels:--no-labels" "dense:--workload uvp5_dense" "densesaved:--workload uvp5_dense_saved" "deconv:--deconv-standin"; do
  name=${v%%:*}; args=${v#*:}
  envs=$cfg; [ "$cfg" = "-" ] && envs="AB_NONE=1"
  env $envs timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu off --host-batches 0 $args > gpurun_out/${P}_$name.json 2> gpurun_out/${P}_$name.err || { tail -5 gpurun_out/${P}_$name.err; exit 1; }
  python - gpurun_out/${P}_$name.json "$cfg $name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
g=d["roofline_groups"]
print(f"{sys.argv[2]:30s} {d['value']:9.0f} f/s  {d['ms_per_step']:7.3f} ms/step  path {d['path']['frac_of_hbm_peak_per_gpu']:.3f} ", {k[:6]:(v["avg_ms"],v["avg_ms_alone"]) for k,v in g.items()})
PY
done
done
