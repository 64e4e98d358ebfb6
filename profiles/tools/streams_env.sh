#!/bin/bash
# bench.py --streams K under an environment setting: arguments are VAR=value:K pairs
set -e
mkdir -p gpurun_out
for a in "$@"; do
  v=${a%%:*}; k=${a##*:}
  export $v
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --cpu-frames 0 --streams $k > gpurun_out/streams_env.json
  python - <<PY
import json
d = json.load(open("gpurun_out/streams_env.json"))
print("$v streams $k", d["value"], d["ms_per_step"])
PY
done
