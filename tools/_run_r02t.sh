#!/bin/bash
mkdir -p gpurun_out/r02v
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py tests/test_api_gpu.py -q -m gpu > gpurun_out/r02v/t.log 2>&1; tail -5 gpurun_out/r02v/t.log
timeout -k 10 300 python tools/_policy_probe.py 2>&1 | grep -v amdgpu.ids | grep "E="
timeout -k 10 300 python tools/policy_bench.py 2>&1 | grep -v amdgpu.ids
