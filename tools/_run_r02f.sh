mkdir -p gpurun_out/r02f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_api_gpu.py tests/test_path_gpu.py -q -m gpu -x -k "actor_head or sampling_epilogue or quantile or api_gpu or tiny_run or world_model or behaviour" > gpurun_out/r02f/tests.log 2>&1; echo rc=$? >> gpurun_out/r02f/tests.log; tail -12 gpurun_out/r02f/tests.log
timeout -k 10 200 python tools/imag_bench.py cfg2 > gpurun_out/r02f/imag.log 2>&1; head -9 gpurun_out/r02f/imag.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02f/bench.json')); print(d['ms_per_step'], d['value'], d['timers']); print(json.dumps(d['roofline']['by_kernel'])[:1500])"
