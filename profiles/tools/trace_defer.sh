# kernel trace of bench.py --streams 3; was used with an experimental PSEG_DEFER=1 switch that is no longer in the source (profiles/README.md): without it this traces the plain turn
set -e
R=$PWD
export TMPDIR=/tmp PSEG_DEFER=1
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_defer -- python3 $R/bench.py --cpu-frames 0 --streams 3 --steps 16 --warmup 4 > $R/gpurun_out/tr_defer.log 2>&1
cd $R
f=$(find gpurun_out/tr_defer -name '*kernel_trace.csv' | head -1)
python profiles/tools/trace_overlap.py $f 60 > gpurun_out/tr_defer.txt
head -14 gpurun_out/tr_defer.txt
