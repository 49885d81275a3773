#!/bin/bash
mkdir -p gpurun_out/r03f
timeout -k 10 900 python -m pytest tests/ -q -m gpu -x > gpurun_out/r03f/t.log 2>&1; tail -4 gpurun_out/r03f/t.log
timeout -k 10 400 python bench.py > gpurun_out/r03f/bench.log 2>&1 || exit 1
tail -1 gpurun_out/r03f/bench.log | cut -c1-220
timeout -k 10 300 python tools/imag_bench.py --json gpurun_out/r03f/imag.json > gpurun_out/r03f/imag.log 2>&1; grep -v amdgpu gpurun_out/r03f/imag.log | head -16
timeout -k 10 300 python tools/policy_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03f/policy.log
timeout -k 10 200 python tools/scan_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03f/scan.log
