for v in ${VARIANTS:-base c16 base c16 base c16}; do
  if [ $v = base ]; then unset PSEG_LIB; else export PSEG_LIB=$PWD/imageprocessing_amd/libpiscoseg_$v.so; fi
  timeout -k 10 200 python bench.py --steps 80 --warmup 6 --cpu-frames 0 > gpurun_out/ab40_$v.json || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/ab40_$v.json')); print('$v', d['value'], d['ms_per_step'], d['path']['group_ms_per_step_single_context'])"
done
