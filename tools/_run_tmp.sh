#!/bin/bash
mkdir -p gpurun_out/r03v
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "few_row or gemm" > gpurun_out/r03v/t.log 2>&1; tail -2 gpurun_out/r03v/t.log
timeout -k 10 600 python -m pytest tests/test_path_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "cfg4 or cfg5 or property" > gpurun_out/r03v/t2.log 2>&1; tail -2 gpurun_out/r03v/t2.log
timeout -k 10 500 python bench.py --no-cpu-baseline --config cfg4 --steps 8 --warmup 3 > gpurun_out/r03v/bench_cfg4.log 2>&1 || { tail -5 gpurun_out/r03v/bench_cfg4.log; exit 1; }
echo "cfg4 $(tail -1 gpurun_out/r03v/bench_cfg4.log | cut -c95-180)"
