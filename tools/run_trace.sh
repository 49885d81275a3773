# rocprofv3 kernel trace of update replays only (bench.py --plain): per-kernel time per update = total / (steps + warmup)
# usage: tools/run_trace.sh TAG [CFG] [--serial]   (--serial: one update after the other, kernels on the whole chip)
TAG=${1:-r04}
CFG=${2:-cfg2}
MODE=${3:-}
SUF=${MODE:+_serial}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$CFG$SUF -- python3 bench.py --plain --config $CFG --steps 10 --warmup 6 $MODE > $OUT/trace_${CFG}${SUF}_bench.json 2> $OUT/trace_$CFG$SUF.err
echo "trace rc=$?"
ST=$(find $OUT/trace_$CFG$SUF -name "*kernel_stats.csv" | head -1)
cp $ST $OUT/kernel_stats_$CFG$SUF.csv
rm -rf $OUT/trace_$CFG$SUF
