#!/bin/bash
# A/B of several builds of the library on one box (arguments: names of imageprocessing_amd/libpiscoseg_<name>.so, "base" =
# the product build): prints the kernel-group times of each
set -e
mkdir -p gpurun_out
for v in "$@"; do
  if [ $v = base ]; then unset PSEG_LIB; else export PSEG_LIB=$PWD/imageprocessing_amd/libpiscoseg_$v.so; fi
  timeout -k 10 150 python bench.py --steps 12 --warmup 3 --cpu-frames 0 > gpurun_out/ab_$v.json
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_$v.json"))
print("$v", d["value"], d["ms_per_step"], d["path"]["group_ms_per_step_single_context"])
PY
done
