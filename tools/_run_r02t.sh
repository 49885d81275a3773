#!/bin/bash
mkdir -p gpurun_out/r03a
timeout -k 10 900 python -m pytest tests/ -q -m gpu -x > gpurun_out/r03a/t.log 2>&1; tail -4 gpurun_out/r03a/t.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03a/bench.log 2>&1 || exit 1
tail -1 gpurun_out/r03a/bench.log | cut -c95-180
timeout -k 10 300 python bench.py --no-cpu-baseline --config cfg3 --steps 10 --warmup 3 > gpurun_out/r03a/bench_cfg3.log 2>&1 || exit 1
tail -1 gpurun_out/r03a/bench_cfg3.log | cut -c95-180
