# A/B of the observe-scan prologue fusions (csrc/scanops.hip) with the development library: world-model update replay time
OUT=gpurun_out/${1:-scan_ab}
mkdir -p $OUT
DEV=$PWD/dreamerv3-torch_amd/dv3hip/libdv3hip_dev.so
for v in "all:" "none:DV3_FUSE_SCAN_ROW=0" "ln_only:DV3_FUSE_SCAN_LNBWD=0 DV3_FUSE_SCAN_CS=0" "lnbwd_only:DV3_FUSE_SCAN_LN=0 DV3_FUSE_SCAN_CS=0" "cs_only:DV3_FUSE_SCAN_LN=0 DV3_FUSE_SCAN_LNBWD=0"; do
  name=${v%%:*}; envs=${v#*:}
  env DV3HIP_LIB=$DEV $envs python tools/wm_bench.py > $OUT/wm_$name.txt 2>&1
  echo "$name: $(grep 'world-model update' $OUT/wm_$name.txt)"
done
