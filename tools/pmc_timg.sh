# MFMA-busy counters for the imagination rollout ALONE: two rocprofv3 --pmc runs of a command that,
# after one warm update, only replays the T_img hipGraph (R1 and R2 times); the difference of the counter sums divided by
# R2 - R1 is one rollout's, free of the warm-up launches.  Program directly after `--` (no shell / env hop).
TAG=${1:-r03}
OUT=gpurun_out/${TAG}_timg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( while true; do date >> $OUT/heartbeat.log; sleep 45; done ) &
HB=$!
for R in 10 60; do
  timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace \
      --output-format csv -d $OUT/pmc_$R -- python3 tools/imag_bench.py cfg2 --replays $R > $OUT/run_$R.json 2> $OUT/run_$R.err
  echo "replays $R rc=$?"
done
kill $HB
C10=$(find $OUT/pmc_10 -name "*counter_collection.csv" | head -1)
C60=$(find $OUT/pmc_60 -name "*counter_collection.csv" | head -1)
python tools/pmc_timg.py $C10 $C60 $OUT/run_60.json $OUT/pmc_timg.json
rm -rf $OUT/pmc_10 $OUT/pmc_60
