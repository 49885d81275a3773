#!/bin/bash
mkdir -p gpurun_out/r03h
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv" > gpurun_out/r03h/t.log 2>&1; tail -2 gpurun_out/r03h/t.log
timeout -k 10 300 python tools/conv_bench.py --only conv_wgrad --no-dense --depth 96 --frames 4096 --reps 3 2>&1 | grep "wgrad"
timeout -k 10 300 python tools/conv_bench.py --only conv_wgrad --no-dense 2>&1 | grep "wgrad"
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03h/bench.log 2>&1 || exit 1
tail -1 gpurun_out/r03h/bench.log | cut -c95-180
timeout -k 10 500 python bench.py --no-cpu-baseline --config cfg4 --steps 5 --warmup 2 > gpurun_out/r03h/bench_cfg4.log 2>&1 || exit 1
tail -1 gpurun_out/r03h/bench_cfg4.log | cut -c95-180
timeout -k 10 500 python bench.py --no-cpu-baseline --config cfg5 --steps 5 --warmup 2 > gpurun_out/r03h/bench_cfg5.log 2>&1 || exit 1
tail -1 gpurun_out/r03h/bench_cfg5.log | cut -c95-180
