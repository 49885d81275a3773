mkdir -p gpurun_out/r02j
timeout -k 10 300 python tools/_policy_probe.py > gpurun_out/r02j/probe.log 2>&1; tail -4 gpurun_out/r02j/probe.log
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py tests/test_path_gpu.py tests/test_api_gpu.py -q -m gpu -x -k "actor_head or sampling_epilogue or layernorm_on_load or next_step or agent_gpu or tiny_run or world_model or behaviour or api_gpu" > gpurun_out/r02j/tests.log 2>&1; echo rc=$? >> gpurun_out/r02j/tests.log; tail -6 gpurun_out/r02j/tests.log
timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02j/imag.log 2>&1; head -10 gpurun_out/r02j/imag.log
DV3_LN_ON_LOAD=0 timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02j/imag_noln.log 2>&1; head -2 gpurun_out/r02j/imag_noln.log | tail -1
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "cfg2 or cfg3" > gpurun_out/r02j/full.log 2>&1; tail -3 gpurun_out/r02j/full.log
