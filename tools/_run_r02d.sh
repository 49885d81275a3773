mkdir -p gpurun_out/r02d
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_path_gpu.py -q -m gpu -x -k "gru_fwd or onehot_linear or actor_head or teacher or sampling_epilogue or tiny_run or world_model or behaviour" > gpurun_out/r02d/tests.log 2>&1; echo rc=$? >> gpurun_out/r02d/tests.log; tail -4 gpurun_out/r02d/tests.log
timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02d/imag.log 2>&1; cat gpurun_out/r02d/imag.log
DV3_STACK_DETER=0 timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02d/imag_nostack.log 2>&1; head -2 gpurun_out/r02d/imag_nostack.log | tail -1
for w in 4 8; do for rn in 2 4 8; do
  echo "== waves $w rn $rn" >> gpurun_out/r02d/gemm_sweep.log
  DV3_DIRECT_WAVES=$w DV3_DIRECT_RN=$rn timeout -k 10 120 python tools/gemm_bench.py --tiles 9 --reps 30 2>&1 | grep -E "1024x 1536x  1024   0  1|1024x  512x   512   0  1|1024x 1024x   512   0  1" >> gpurun_out/r02d/gemm_sweep.log
done; done
cat gpurun_out/r02d/gemm_sweep.log
timeout -k 10 400 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "cfg2 or cfg3" > gpurun_out/r02d/full.log 2>&1; tail -3 gpurun_out/r02d/full.log
