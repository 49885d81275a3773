#!/bin/bash
mkdir -p gpurun_out/r03u
timeout -k 10 900 python -m pytest tests/ -q -m gpu -x > gpurun_out/r03u/t.log 2>&1; tail -3 gpurun_out/r03u/t.log
timeout -k 10 300 python tools/imag_bench.py --beh 2>&1 | grep -v amdgpu.ids | grep "T_img\|tensorstats\|quantile\|14336x512x512\|sum "
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03u/bench.log 2>&1 || exit 1
tail -1 gpurun_out/r03u/bench.log | cut -c95-180
