#!/bin/bash
mkdir -p gpurun_out/r02x
DV3_SIDE_STREAM=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02x/bench_side.log 2>&1 || { tail -5 gpurun_out/r02x/bench_side.log; exit 1; }
tail -1 gpurun_out/r02x/bench_side.log | cut -c1-200
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02x/bench_noside.log 2>&1 || exit 1
tail -1 gpurun_out/r02x/bench_noside.log | cut -c1-200
