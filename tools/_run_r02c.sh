mkdir -p gpurun_out/r02c
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_path_gpu.py -q -m gpu -x -k "onehot_linear or actor_head or teacher or sampling_epilogue or tiny_run or world_model or behaviour" > gpurun_out/r02c/tests.log 2>&1; echo rc=$? >> gpurun_out/r02c/tests.log; tail -6 gpurun_out/r02c/tests.log
for ch in 1 2 4; do
  DV3_IMAG_CHAINS=$ch timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02c/imag_ch$ch.log 2>&1
  head -3 gpurun_out/r02c/imag_ch$ch.log | tail -1
done
DV3_FUSE_SAMPLE=0 timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02c/imag_nosmp.log 2>&1; head -3 gpurun_out/r02c/imag_nosmp.log | tail -1
cat gpurun_out/r02c/imag_ch1.log
DV3_IMAG_CHAINS=2 timeout -k 10 300 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "cfg2" > gpurun_out/r02c/full_ch2.log 2>&1; tail -3 gpurun_out/r02c/full_ch2.log
