#!/bin/bash
# per-kernel statistics of a short single-context bench run: kstats.sh <prefix under gpurun_out/> [bench args]
R=$PWD; P=${1:-kstats}; shift
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/${P}_stats
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_stats -- python3 $R/bench.py --cpu off --host-batches 0 --streams 1 --steps 3 --warmup 1 "$@" > $R/gpurun_out/${P}_stats.log 2>&1
cd $R
f=$(find gpurun_out/${P}_stats -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${P}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("k_synth"): continue
    print(f"{r['Name'][:58]:58s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
