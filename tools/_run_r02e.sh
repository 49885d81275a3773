mkdir -p gpurun_out/r02e
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py tests/test_api_gpu.py tests/test_path_gpu.py -q -m gpu -x -k "actor_head or dreamer_agent or policy_steps or quantile or api_gpu or tiny_run or world_model or behaviour or gru_fwd" > gpurun_out/r02e/tests.log 2>&1; echo rc=$? >> gpurun_out/r02e/tests.log; tail -12 gpurun_out/r02e/tests.log
timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02e/imag.log 2>&1; head -8 gpurun_out/r02e/imag.log
for v in "1 1" "0 1" "0 2" "0 4"; do set -- $v
  echo "== pipe $1 batch $2" >> gpurun_out/r02e/gemm_sweep.log
  DV3_DIRECT_PIPE=$1 DV3_DIRECT_BATCH=$2 timeout -k 10 120 python tools/gemm_bench.py --tiles 9 --reps 30 2>&1 | grep -E "1024x 1536x  1024   0  1|1024x  512x   512   0  1|1024x 1024x   512   0  1|1024x  512x  1536   0  0|1024x 1536x  1024   0  0" >> gpurun_out/r02e/gemm_sweep.log
done
cat gpurun_out/r02e/gemm_sweep.log
DV3_DIRECT_PIPE=0 DV3_DIRECT_BATCH=2 timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02e/imag_p0b2.log 2>&1; head -2 gpurun_out/r02e/imag_p0b2.log | tail -1
rocprofv3 -L > gpurun_out/r02e/counters.txt 2>&1; grep -c . gpurun_out/r02e/counters.txt
