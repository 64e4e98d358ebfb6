#!/bin/bash
# throughput against the number of contexts (host threads) sharing the GPU
set -e
mkdir -p gpurun_out
for k in "$@"; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --cpu-frames 0 --streams $k > gpurun_out/streams_$k.json
  python - <<PY
import json
d = json.load(open("gpurun_out/streams_$k.json"))
print("streams $k", d["value"], d["ms_per_step"])
PY
done
