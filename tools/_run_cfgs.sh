#!/bin/bash
# BASELINE configs other than the headline one (with the CPU-oracle leg): one JSON line each under gpurun_out/$1
OUT=gpurun_out/${1:-cfgs}
mkdir -p $OUT
for c in cfg1 cfg3 cfg4 cfg5; do
  timeout -k 10 600 python bench.py --config $c --steps 10 --warmup 3 > $OUT/bench_$c.log 2>&1 || { echo "FAILED $c"; tail -5 $OUT/bench_$c.log; exit 1; }
  tail -1 $OUT/bench_$c.log | cut -c1-200
done
