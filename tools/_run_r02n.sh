mkdir -p gpurun_out/r02n
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "scan_" > gpurun_out/r02n/tests.log 2>&1; echo rc=$? >> gpurun_out/r02n/tests.log; tail -4 gpurun_out/r02n/tests.log
timeout -k 10 200 python tools/scan_bench.py > gpurun_out/r02n/scan.log 2>&1; tail -1 gpurun_out/r02n/scan.log
DV3_FUSE_SCAN=0 timeout -k 10 200 python tools/scan_bench.py > gpurun_out/r02n/scan_unfused.log 2>&1; tail -1 gpurun_out/r02n/scan_unfused.log
timeout -k 10 300 python tools/_policy_probe.py > gpurun_out/r02n/probe.log 2>&1; tail -16 gpurun_out/r02n/probe.log
