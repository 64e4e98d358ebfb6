#!/bin/bash
# round 3 closing measurements on one box: r3_final.sh <prefix under gpurun_out/>
# kernel statistics of the default bench command (three contexts, deferred fills) and of one context, FETCH_SIZE / WRITE_SIZE
# passes (separate runs), the default bench line with the CPU baseline, the variants
set -e
R=$GRAFT_REPO_ROOT; P=${1:-fin}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
stats() {   # stats <name> <bench args...>
  local name=$1; shift
  rm -rf $R/gpurun_out/${P}_${name}_stats
  ( cd /tmp && timeout -s KILL 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_${name}_stats -- python3 $R/bench.py --cpu off --host-batches 0 "$@" > $R/gpurun_out/${P}_${name}_stats.log 2>&1 )
  cp "$(find $R/gpurun_out/${P}_${name}_stats -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/${P}_kernel_stats_${name}.csv
  rm -rf $R/gpurun_out/${P}_${name}_stats
}
stats default --steps 5 --warmup 1
stats one_context --streams 1 --steps 3 --warmup 1
stats uvp5_dense_saved --streams 1 --steps 3 --warmup 1 --workload uvp5_dense_saved
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/${P}_pmc_$c
  ( cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${P}_pmc_$c -- python3 $R/profiles/tools/pmc_probe.py > $R/gpurun_out/${P}_pmc_$c.log 2>&1 )
  cp "$(find $R/gpurun_out/${P}_pmc_$c -name '*counter_collection.csv' | head -1)" $R/gpurun_out/${P}_pmc_$c.csv
  rm -rf $R/gpurun_out/${P}_pmc_$c
done
cd $R
python3 profiles/tools/pmc_to_traffic.py gpurun_out/${P}_pmc_FETCH_SIZE.csv gpurun_out/${P}_pmc_WRITE_SIZE.csv gpurun_out/${P}_traffic.json > gpurun_out/${P}_traffic.txt
cp gpurun_out/${P}_traffic.json profiles/traffic.json          # (so that the bench lines below quote this code's counters)
timeout -k 10 400 python bench.py > gpurun_out/${P}_bench_default.json 2> gpurun_out/${P}_bench_default.err
for v in "one_context:--streams 1" "two_contexts:--streams 2" "threads:--driver threads" "no_labels:--no-labels" "uvp5_dense:--workload uvp5_dense" "uvp5_dense_saved:--workload uvp5_dense_saved" "deconv_standin:--deconv-standin" "force_dist:--force-dist" "no_deferred_fill:"; do
  name=${v%%:*}; args=${v#*:}
  envs="AB_NONE=1"; [ "$name" = "no_deferred_fill" ] && envs="PSEG_DEFER_FILL=0"
  env $envs timeout -k 10 300 python bench.py --cpu off --host-batches 0 $args > gpurun_out/${P}_bench_$name.json 2> gpurun_out/${P}_bench_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/${P}_bench_$name.err; }
done
python3 - $P <<'PY'
import json, sys, glob
P = sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/{P}_bench_*.json")):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    g = d["roofline_groups"]
    print(f"{f.split('_bench_')[1][:-5]:18s} {d['value']:9.0f} f/s {d['ms_per_step']:7.3f} ms/step path {d['path']['frac_of_hbm_peak_per_gpu']:.3f} roofline {d['roofline']['kernel']} {d['roofline']['frac']:.3f} deferred {d['path'].get('fills_deferred')} gaveup {d['path'].get('deferred_fills_given_up')}",
          {k[:6]: (v["avg_ms"], v["avg_ms_alone"]) for k, v in g.items()}, d.get("cpu_baseline", {}).get("value"))
PY
