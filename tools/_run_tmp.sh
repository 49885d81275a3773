#!/bin/bash
mkdir -p gpurun_out/r03w
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv" > gpurun_out/r03w/t.log 2>&1; tail -2 gpurun_out/r03w/t.log
for v in 0 1; do echo "== DV3_C3_MFMA=$v"; DV3_C3_MFMA=$v timeout -k 10 300 python tools/wm_bench.py 2>&1 | grep "conv_s2_c3\|world-model"; done
