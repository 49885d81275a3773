#!/bin/bash
mkdir -p gpurun_out/r02w
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_path_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x > gpurun_out/r02w/t.log 2>&1; tail -4 gpurun_out/r02w/t.log
timeout -k 10 300 python tools/imag_bench.py 2>&1 | grep -v amdgpu.ids | head -3
timeout -k 10 300 python bench.py > gpurun_out/r02w/bench.log 2>&1 || exit 1
tail -1 gpurun_out/r02w/bench.log | cut -c1-200
