# rocprofv3 kernel trace of update replays only (bench.py --plain): per-kernel time per update = total / (steps + warmup)
TAG=${1:-r03}
CFG=${2:-cfg2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$CFG -- python3 bench.py --plain --config $CFG --steps 10 --warmup 2 > $OUT/trace_${CFG}_bench.json 2> $OUT/trace_$CFG.err
echo "trace rc=$?"
ST=$(find $OUT/trace_$CFG -name "*kernel_stats.csv" | head -1)
cp $ST $OUT/kernel_stats_$CFG.csv
rm -rf $OUT/trace_$CFG
