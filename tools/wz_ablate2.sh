L=$GRAFT_REPO_ROOT/brats2019_amd/lib
for shape in "32 64" "64 32" "128 16"; do
  for d in 11 16 32 64 19; do
    RU_WZ=1 RU_LIB_PATH=$L/libresunet_hip_dbg$d.so RU_SB2_DEBUG=$d python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  done
done
