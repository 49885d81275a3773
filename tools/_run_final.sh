#!/bin/bash
# final-evidence run for docs: other configs, then the rocprofv3 trace + PMC passes
mkdir -p gpurun_out/r03g
for c in cfg1 cfg3 cfg4 cfg5; do
  timeout -k 10 500 python bench.py --config $c --steps 10 --warmup 3 > gpurun_out/r03g/bench_$c.log 2>&1 || { echo "FAILED $c"; tail -5 gpurun_out/r03g/bench_$c.log; exit 1; }
  tail -1 gpurun_out/r03g/bench_$c.log | cut -c1-200
done
bash tools/_run_pmc.sh r02c > gpurun_out/r02c_pmc.log 2>&1
tail -30 gpurun_out/r02c_pmc.log
