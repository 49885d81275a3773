mkdir -p gpurun_out/r02p
timeout -k 10 1150 python -m pytest tests/ -q -m gpu > gpurun_out/r02p/tests.log 2>&1; echo rc=$? >> gpurun_out/r02p/tests.log; tail -12 gpurun_out/r02p/tests.log
