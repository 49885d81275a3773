# A/B of the observe-scan prologue fusions (csrc/scanops.hip) with the development library: world-model update replay time
# usage: bash tools/scan_ab.sh <out-tag> [cfg2] [variants: all none ln_only lnbwd_only cs_only]
OUT=gpurun_out/${1:-scan_ab}
CFG=${2:-cfg2}
shift; shift
VARS=${@:-all none ln_only lnbwd_only cs_only no_grubwd}
mkdir -p $OUT
DEV=$PWD/dreamerv3-torch_amd/dv3hip/libdv3hip_dev.so
for name in $VARS; do
  case $name in
    all) envs="";;
    none) envs="DV3_FUSE_SCAN_ROW=0";;
    ln_only) envs="DV3_FUSE_SCAN_LNBWD=0 DV3_FUSE_SCAN_CS=0 DV3_FUSE_SCAN_GRUBWD=0";;
    lnbwd_only) envs="DV3_FUSE_SCAN_LN=0 DV3_FUSE_SCAN_CS=0 DV3_FUSE_SCAN_GRUBWD=0";;
    cs_only) envs="DV3_FUSE_SCAN_LN=0 DV3_FUSE_SCAN_LNBWD=0 DV3_FUSE_SCAN_GRUBWD=0";;
    no_grubwd) envs="DV3_FUSE_SCAN_GRUBWD=0";;
  esac
  env DV3HIP_LIB=$DEV $envs python tools/wm_bench.py $CFG > $OUT/wm_${CFG}_$name.txt 2>&1
  echo "$CFG $name: $(grep 'world-model update' $OUT/wm_${CFG}_$name.txt)"
done
