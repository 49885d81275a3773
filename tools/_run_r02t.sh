#!/bin/bash
mkdir -p gpurun_out/r02t
timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" > gpurun_out/r02t/tk.log 2>&1; tail -5 gpurun_out/r02t/tk.log
timeout -k 10 300 python tools/imag_bench.py > gpurun_out/r02t/imag.log 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/r02t/imag.log | head -50
timeout -k 10 300 python bench.py > gpurun_out/r02t/bench.log 2>&1 || exit 1
tail -1 gpurun_out/r02t/bench.log | cut -c1-300
