#!/bin/bash
# bench.py --streams 3 under environment settings (arguments VAR=value): throughput and single-context kernel-group times
set -e
mkdir -p gpurun_out
for v in "$@"; do
  export $v
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --cpu-frames 0 --streams 3 > gpurun_out/env_groups.json
  python - <<PY
import json
d = json.load(open("gpurun_out/env_groups.json"))
print("$v", d["value"], d["ms_per_step"], d["path"]["group_ms_per_step_single_context"])
PY
done
