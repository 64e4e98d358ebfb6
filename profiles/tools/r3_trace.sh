#!/bin/bash
# kernel trace of a short bench run: r3_trace.sh <prefix> [bench args]; timeline + idle gaps into gpurun_out/<prefix>_timeline.txt
R=$PWD; P=${1:-trace}; shift
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${P}_trace -- python3 $R/bench.py --cpu off --host-batches 0 --steps 4 --warmup 1 "$@" > $R/gpurun_out/${P}_trace.log 2>&1
cd $R
f=$(find gpurun_out/${P}_trace -name '*kernel_trace.csv' | head -1)
{ python3 profiles/tools/trace_gaps.py "$f"; python3 profiles/tools/trace_timeline.py "$f" ${TL_ROWS:-110} $TL_ANCHOR; } > gpurun_out/${P}_timeline.txt 2>&1
rm -rf gpurun_out/${P}_trace
