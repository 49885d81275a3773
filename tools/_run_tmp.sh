for pad in 0 32 64 544; do echo "== pad $pad"; DV3_BENCH_PAD=$pad timeout -k 10 400 python tools/gemm_bench.py --tiles 15,4 --only "cfg4 GRU" --reps 5 2>&1 | grep "big"; done
