#!/bin/bash
# per-kernel A/B of library builds on one box: tools/ab_lib_kernel.sh "<kernel name pattern>" lib1.so lib2.so ...   (serial step profile, avg us of matching kernels)
PAT=$1; shift
for lib in "" "$@"; do
  tag=$(basename "${lib:-default}" .so)
  [ -n "$lib" ] && export RU_LIB_PATH=$lib || unset RU_LIB_PATH
  RU_SIDE_STREAM=0 bash $GRAFT_REPO_ROOT/tools/profile_step.sh ab_$tag 4 > /dev/null 2>&1
  echo "== $tag"; grep -E "$PAT" $GRAFT_REPO_ROOT/gpurun_out/ab_${tag}_train_step_kernel_stats.txt | cut -c1-130
done
