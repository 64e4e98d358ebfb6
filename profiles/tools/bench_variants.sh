#!/bin/bash
# the bench variants of BASELINE.md section 5 with the round's final code (CPU baseline off: it is in the default line)
cd $GRAFT_REPO_ROOT
P=${1:-var}
run() { name=$1; shift; timeout -k 10 200 python3 bench.py --cpu off --host-batches 0 "$@" > gpurun_out/${P}_$name.json 2> gpurun_out/${P}_$name.err || { echo "$name failed"; tail -3 gpurun_out/${P}_$name.err; return 1; }
        python3 -c "import json; d=json.loads(open('gpurun_out/${P}_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['path']['frac_of_hbm_peak_per_gpu'])"; }
run one_context --streams 1 &&
run threads --driver threads &&
run no_labels --no-labels &&
run uvp5_dense --workload uvp5_dense &&
run uvp5_dense_saved --workload uvp5_dense_saved &&
run force_dist --force-dist &&
timeout -k 10 250 bash profiles/tools/kstats.sh ${P}_dense --workload uvp5_dense_saved > gpurun_out/${P}_dense_kstats.txt && head -16 gpurun_out/${P}_dense_kstats.txt
