set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t2.log 2>&1 || { tail -30 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu off --host-batches 0 > gpurun_out/b_default.json 2> gpurun_out/b_default.err
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu off --host-batches 0 --deconv-standin > gpurun_out/b_deconv.json 2> gpurun_out/b_deconv.err
cat gpurun_out/b_default.json gpurun_out/b_deconv.json
