#!/bin/bash
# two ranks sharing the one GPU of the box (gloo, --single-device): a rehearsal of the N > 1 code path, and two processes' band
# kernels side by side on one device -- the scan in block-index order must neither time out nor change a result
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --single-device --cpu off --host-batches 0 --steps 3 --warmup 1 "$@" > gpurun_out/two_ranks.json 2> gpurun_out/two_ranks.err || { echo FAILED; tail -20 gpurun_out/two_ranks.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/two_ranks.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['scaling'], d['path'])"
