#!/bin/bash
mkdir -p gpurun_out/r03n
timeout -k 10 900 python -m pytest tests/ -q -m gpu -x > gpurun_out/r03n/t.log 2>&1; tail -3 gpurun_out/r03n/t.log
for c in cfg2 cfg3 cfg4 cfg5; do
timeout -k 10 500 python bench.py --no-cpu-baseline --config $c --steps 8 --warmup 3 > gpurun_out/r03n/bench_$c.log 2>&1 || { tail -5 gpurun_out/r03n/bench_$c.log; exit 1; }
echo "$c $(tail -1 gpurun_out/r03n/bench_$c.log | cut -c95-180)"
done
