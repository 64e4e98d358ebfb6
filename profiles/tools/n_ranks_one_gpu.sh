#!/bin/bash
# N ranks sharing the one GPU of the box (gloo, --single-device), a short stack: rehearsal of the N > 1 code path for rank counts the
# box cannot give GPUs to (parts per rank differ, ranks with few or no parts still join every gather).  n_ranks_one_gpu.sh <N> <frames>
cd $GRAFT_REPO_ROOT
N=$1; F=$2
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus $N --backend gloo --single-device --cpu off --host-batches 0 --steps 2 --warmup 1 --frames $F --streams 2 > gpurun_out/nranks_$N.json 2> gpurun_out/nranks_$N.err || { echo "FAILED N=$N"; tail -20 gpurun_out/nranks_$N.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/nranks_$N.json').read().strip().splitlines()[-1]); print('N=$N frames=$F', d['value'], d['ms_per_step'], d['scaling'], d['n_gpus'], {k: d['path'][k] for k in ('ranks_seen', 'halo_overhead', 'table_gather_messages_per_step', 'contexts_consistent_after_timed_region', 'band_scan_fallbacks')})"
