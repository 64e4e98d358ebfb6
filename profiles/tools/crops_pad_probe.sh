# timing probe: needs a build that launches k_crops_sums with X_PAD_KB kilobytes of dynamic LDS (not in the product source) as libpiscoseg_x.so
export PSEG_LIB=$PWD/imageprocessing_amd/libpiscoseg_x.so
for kb in 0 8 14 20 28 0 14; do
  ( export X_PAD_KB=$kb; timeout -k 10 200 python bench.py --steps 40 --warmup 4 --cpu-frames 0 > gpurun_out/pad_$kb.json ) || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/pad_$kb.json')); print('pad_kb $kb', d['value'], d['ms_per_step'], d['path']['group_ms_per_step_single_context'])"
done
