mkdir -p gpurun_out/r03s
timeout -k 10 1000 python -m pytest tests/ -q -m gpu > gpurun_out/r03s/t.log 2>&1; tail -4 gpurun_out/r03s/t.log
timeout -k 10 400 python bench.py > gpurun_out/r03s/bench.log 2>&1 || exit 1
tail -1 gpurun_out/r03s/bench.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python tools/imag_bench.py 2>&1 | grep -v amdgpu | head -3
timeout -k 10 300 python tools/policy_bench.py 2>&1 | grep -v amdgpu
