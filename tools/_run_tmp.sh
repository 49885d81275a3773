#!/bin/bash
mkdir -p gpurun_out/r03r
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv" > gpurun_out/r03r/t.log 2>&1; tail -2 gpurun_out/r03r/t.log
timeout -k 10 300 python tools/conv_bench.py --only convT_s2 --no-dense 2>&1 | grep "convT_s2 "
timeout -k 10 300 python tools/conv_bench.py --only convT_s2 --no-dense --depth 96 --frames 4096 --reps 3 2>&1 | grep "convT_s2 "
