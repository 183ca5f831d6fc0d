L=$GRAFT_REPO_ROOT/brats2019_amd/lib
for r in 1 2 3; do
  RU_CONV_FLAGS=35 python3 $GRAFT_REPO_ROOT/tools/conv_time.py 16 128 4 20 2>/dev/null
  RU_CONV_FLAGS=35 RU_LIB_PATH=$L/libresunet_hip_dbg2048.so RU_SB2_DEBUG=2048 python3 $GRAFT_REPO_ROOT/tools/conv_time.py 16 128 4 20 2>/dev/null
done
