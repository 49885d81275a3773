# rocprofv3 evidence for profiles/: kernel-trace stats, SQ (MFMA busy) counters, FETCH_SIZE and WRITE_SIZE passes
# (separate passes; counters only with --kernel-trace, never with sys/runtime traces)
# usage: tools/run_pmc.sh TAG [--serial]   default: the pipelined timed region of bench.py (every kernel on a 128-CU lane);
# --serial: one update after the other (kernels on the whole chip, except the reverse scan and the deferred weight gradients)
TAG=${1:-r04}
MODE=${2:-}
OUT=gpurun_out/$TAG${MODE:+_serial}
SIMDS=$([ -n "$MODE" ] && echo 1024 || echo 512)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --plain --steps 6 --warmup 6 $MODE"  # update replays only: every traced launch belongs to an update
# counter passes run for minutes without output: keep gpurun's silence watchdog fed
( while true; do date >> $OUT/heartbeat.log; sleep 45; done ) &
HB=$!
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
echo "trace rc=$?"
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
echo "sq rc=$?"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
echo "fetch rc=$?"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
echo "write rc=$?"
find $OUT -name "*.csv" | head -20
SQ=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1)
FE=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
WR=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
ST=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python tools/pmc_mfma.py $SQ $OUT/pmc_mfma.json --simds $SIMDS | head -16
python tools/pmc_traffic.py $FE $WR $OUT/pmc_traffic.json | head -14
cp $ST $OUT/kernel_stats.csv
# the raw counter CSVs are large: keep only the summaries
rm -rf $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
find $OUT/trace -name "*.csv" ! -name "*kernel_stats.csv" -delete
kill $HB
