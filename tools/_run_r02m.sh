mkdir -p gpurun_out/r02m
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_path_gpu.py -q -m gpu -x -k "scan_ or tiny_run or world_model or behaviour" > gpurun_out/r02m/tests.log 2>&1; echo rc=$? >> gpurun_out/r02m/tests.log; tail -8 gpurun_out/r02m/tests.log
timeout -k 10 200 python tools/scan_bench.py > gpurun_out/r02m/scan.log 2>&1; tail -1 gpurun_out/r02m/scan.log
DV3_FUSE_SCAN=0 timeout -k 10 200 python tools/scan_bench.py > gpurun_out/r02m/scan_unfused.log 2>&1; tail -1 gpurun_out/r02m/scan_unfused.log
timeout -k 10 200 python tools/policy_bench.py > gpurun_out/r02m/policy.log 2>&1; tail -3 gpurun_out/r02m/policy.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02m/bench.json 2> gpurun_out/r02m/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02m/bench.json')); print(d['ms_per_step'], d['value'], d['timers'])"
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "cfg2 or cfg1 or cfg3" > gpurun_out/r02m/full.log 2>&1; tail -3 gpurun_out/r02m/full.log
