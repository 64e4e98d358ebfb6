set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k other_windows --timeout 120 --timeout-method thread 2>&1 | tail -3
for nb in 6 5 4 8; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu off --host-batches 0 --n-bg-imgs $nb > gpurun_out/b_win$nb.json 2> gpurun_out/b_win$nb.err
  python - gpurun_out/b_win$nb.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
g=d["roofline_groups"]
print(d["config"]["workload"][-70:], d["value"], {k:(v["avg_ms"],v["avg_ms_alone"]) for k,v in g.items()})
PY
done
