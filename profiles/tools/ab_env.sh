#!/bin/bash
# the same library under different environment settings: prints the kernel-group times of each
set -e
mkdir -p gpurun_out
for v in "$@"; do
  export $v
  timeout -k 10 150 python bench.py --steps 12 --warmup 3 --cpu-frames 0 > gpurun_out/ab_env.json
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_env.json"))
print("$v", d["value"], d["ms_per_step"], d["path"]["group_ms_per_step_single_context"])
PY
done
