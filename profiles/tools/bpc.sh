#!/bin/bash
# throughput against the number of reference batches per library call
set -e
mkdir -p gpurun_out
for k in "$@"; do
  timeout -k 10 200 python bench.py --steps 48 --warmup 6 --cpu-frames 0 --batches-per-call $k --batches 12 > gpurun_out/bpc_$k.json
  python - <<PY
import json
d = json.load(open("gpurun_out/bpc_$k.json"))
print("batches per call $k", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["path"]["group_ms_per_step_single_context"])
PY
done
