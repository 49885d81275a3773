mkdir -p gpurun_out/r02h
timeout -k 10 1000 python -m pytest tests/ -q -m gpu -x --deselect tests/test_fullsize_gpu.py > gpurun_out/r02h/tests.log 2>&1; echo rc=$? >> gpurun_out/r02h/tests.log; tail -8 gpurun_out/r02h/tests.log
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x > gpurun_out/r02h/full.log 2>&1; echo rc=$? >> gpurun_out/r02h/full.log; tail -4 gpurun_out/r02h/full.log
timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02h/imag.log 2>&1; head -10 gpurun_out/r02h/imag.log
timeout -k 10 200 python tools/policy_bench.py > gpurun_out/r02h/policy.log 2>&1; tail -4 gpurun_out/r02h/policy.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02h/bench.json')); print(d['ms_per_step'], d['value'], d['timers'])"
timeout -k 10 200 python tools/scan_bench.py > gpurun_out/r02h/scan.log 2>&1; tail -2 gpurun_out/r02h/scan.log
