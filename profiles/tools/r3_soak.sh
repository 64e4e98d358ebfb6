# round 3: pipelined parity soak, the switch sweep and rehearsals of N ranks on one GPU
set -e
cd $GRAFT_REPO_ROOT
[ -n "$SKIP_SOAK" ] || timeout -k 10 900 python3 -u profiles/tools/parity_soak.py pipe 0 ${SOAK_SEEDS:-60} 2>&1 | tail -8
timeout -k 10 900 bash profiles/tools/switch_sweep.sh
timeout -k 10 500 bash profiles/tools/n_ranks_one_gpu.sh 2 2000
timeout -k 10 500 bash profiles/tools/n_ranks_one_gpu.sh 4 3000
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --single-device --cpu off --host-batches 0 --steps 2 --warmup 1 --frames 2000 > gpurun_out/selfspawn2.json 2> gpurun_out/selfspawn2.err || { tail -20 gpurun_out/selfspawn2.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/selfspawn2.json').read().strip().splitlines()[-1]); print('self-spawned', d['n_gpus'], d['value'], d['path'].get('ranks_seen'), d['path'].get('launch'), d['path'].get('fills_deferred'))"
