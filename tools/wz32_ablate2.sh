L=$GRAFT_REPO_ROOT/brats2019_amd/lib
for shape in "32 64" "64 32" "128 16"; do
  python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  for d in 3 19 11 27 7; do
    RU_LIB_PATH=$L/libresunet_hip_dbg$d.so RU_SB2_DEBUG=$d python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  done
done
