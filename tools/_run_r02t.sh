#!/bin/bash
mkdir -p gpurun_out/r02z
for r in 1 2; do
DV3_CONV_L16=0 DV3_CONVT_L16=0 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02z/bench_off.log 2>&1 || exit 1
echo "off: $(tail -1 gpurun_out/r02z/bench_off.log | cut -c95-180)"
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02z/bench_on.log 2>&1 || exit 1
echo "on : $(tail -1 gpurun_out/r02z/bench_on.log | cut -c95-180)"
done
timeout -k 10 300 python tools/wm_bench.py 2>&1 | grep -v amdgpu | grep -i "conv\|world"
