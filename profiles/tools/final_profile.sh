#!/bin/bash
# the round's closing measurements on one box: per-kernel statistics (one context), FETCH_SIZE and WRITE_SIZE passes (separate
# runs), the default bench line.  Argument: prefix of the files under gpurun_out/
set -e
R=$PWD; P=${1:-final}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_stats -- python3 $R/bench.py --cpu off --host-batches 0 --streams 1 --steps 3 --warmup 1 > $R/gpurun_out/${P}_stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}_fetch -- python3 $R/profiles/tools/pmc_probe.py > $R/gpurun_out/${P}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}_write -- python3 $R/profiles/tools/pmc_probe.py > $R/gpurun_out/${P}_write.log 2>&1
cd $R
timeout -k 10 300 python bench.py > gpurun_out/${P}_bench.json
cat gpurun_out/${P}_bench.json
find gpurun_out/${P}_stats gpurun_out/${P}_fetch gpurun_out/${P}_write -name '*.csv' | head -20
