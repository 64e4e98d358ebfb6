# timing-only probe: needs a build whose crop / hull launches are skipped under X_NOCROPS / X_NOCONVEX (not in the product source) as libpiscoseg_x.so
export PSEG_LIB=$PWD/imageprocessing_amd/libpiscoseg_x.so
for e in NONE X_NOCONVEX X_NOCROPS NONE X_NOCONVEX X_NOCROPS; do
  ( export $e=1; timeout -k 10 200 python bench.py --steps 40 --warmup 4 --cpu-frames 0 > gpurun_out/skip_$e.json ) || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/skip_$e.json')); print('$e', d['value'], d['ms_per_step'], d['path']['group_ms_per_step_single_context'])"
done
