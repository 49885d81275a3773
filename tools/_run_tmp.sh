#!/bin/bash
mkdir -p gpurun_out/r03p
timeout -k 10 900 python -m pytest tests/ -q -m gpu -x > gpurun_out/r03p/t.log 2>&1; tail -3 gpurun_out/r03p/t.log
for c in cfg2 cfg4; do
for v in 1e30 1.4e10; do
DV3_BT_MIN_FLOPS=$v timeout -k 10 500 python bench.py --no-cpu-baseline --config $c --steps 8 --warmup 3 > gpurun_out/r03p/bench_$c.log 2>&1 || { tail -5 gpurun_out/r03p/bench_$c.log; exit 1; }
echo "$c BT_MIN=$v $(tail -1 gpurun_out/r03p/bench_$c.log | cut -c95-180)"
done; done
