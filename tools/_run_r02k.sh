mkdir -p gpurun_out/r02k
timeout -k 10 1100 python -m pytest tests/ -q -m gpu -x --deselect tests/test_fullsize_gpu.py > gpurun_out/r02k/tests.log 2>&1; echo rc=$? >> gpurun_out/r02k/tests.log; tail -6 gpurun_out/r02k/tests.log
timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02k/imag.log 2>&1; head -10 gpurun_out/r02k/imag.log
timeout -k 10 200 python tools/policy_bench.py > gpurun_out/r02k/policy.log 2>&1; tail -4 gpurun_out/r02k/policy.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02k/bench.json 2> gpurun_out/r02k/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02k/bench.json')); print(d['ms_per_step'], d['value'], d['timers']); print(d['roofline']['kernel'], d['roofline']['frac'])"
