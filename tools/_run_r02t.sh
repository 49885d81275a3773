#!/bin/bash
mkdir -p gpurun_out/r03b
timeout -k 10 900 python -m pytest tests/ -q -m gpu -x > gpurun_out/r03b/t.log 2>&1; tail -4 gpurun_out/r03b/t.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03b/bench.log 2>&1 || exit 1
tail -1 gpurun_out/r03b/bench.log | cut -c95-180
