#!/bin/bash
mkdir -p gpurun_out/r03d
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_path_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x -k "carry or path or fullsize or update or world" > gpurun_out/r03d/t.log 2>&1; tail -3 gpurun_out/r03d/t.log
for r in 1 2; do
DV3_FUSE_CARRY=0 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03d/bench_off.log 2>&1 || exit 1
echo "off: $(tail -1 gpurun_out/r03d/bench_off.log | cut -c130-175)"
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03d/bench_on.log 2>&1 || exit 1
echo "on : $(tail -1 gpurun_out/r03d/bench_on.log | cut -c130-175)"
done
