mkdir -p gpurun_out/r02l
timeout -k 10 300 python tools/_policy_probe.py > gpurun_out/r02l/probe.log 2>&1; tail -18 gpurun_out/r02l/probe.log
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_path_gpu.py tests/test_dp_gpu.py -q -m gpu -x -k "actor_normal or tiny_both or dp_gpu or rccl or two_ranks" > gpurun_out/r02l/tests.log 2>&1; echo rc=$? >> gpurun_out/r02l/tests.log; tail -5 gpurun_out/r02l/tests.log
bash tools/_run_pmc.sh r02l_pmc > gpurun_out/r02l/pmc.log 2>&1; tail -40 gpurun_out/r02l/pmc.log
