#!/bin/bash
# First GPU call of the next round (about 4 GPU-minutes): everything the last round prepared but could not measure.
#   1. the split between the two mutate routes at the current aligner speed (environment only: nothing to build)
#   2. the branches round4/qscore-compact, round4/lag1 and round4/lag1-wide (= lag1 + the same schedule for two- and four-word bands)
#      and round4/lag1-bufstore (= lag1-wide + traceback stores through a buffer resource) against this tree (each bit-exact on the interpreted kernels, neither has
#      run on a GPU).  Build them in worktrees HERE first and pass their libraries:
#        for b in qscore-compact lag1 lag1-wide lag1-bufstore; do git worktree add /tmp/r4_$b round4/$b; (cd /tmp/r4_$b && python -m badread_amd.build);
#          cp /tmp/r4_$b/badread_amd/csrc/libbrx_hip.so badread_amd/csrc/variants/libbrx_hip_$b.so; done
#      (lag1 changes the traceback-store geometry: brx_make_geom is compiled into the library, the host side needs nothing else)
# Usage (from the repo root, through gpurun):  bash tools/first_call_next_round.sh > gpurun_out/first_call.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="--steps 2"
bash tools/gpu_ab.sh "|$S" \
  "BRX_TAIL_READS=2730|$S" "BRX_TAIL_READS=8192|$S" "BRX_TAIL_READS=10922|$S" \
  "BRX_HEAD_READS=512|$S" "BRX_HEAD_READS=2048|$S" \
  "BRX_LANE_THRESHOLD=1500|$S" "BRX_LANE_THRESHOLD=6000|$S" \
  "BRX_TB_WINDOW=3|$S" "|$S"
for b in qscore-compact lag1 lag1-wide lag1-bufstore; do
  v=badread_amd/csrc/variants/libbrx_hip_$b.so
  if [ -f $v ]; then
    bash tools/gpu_ab.sh "BRX_LIB_PATH=$PWD/$v|$S" "|$S" "BRX_LIB_PATH=$PWD/$v|$S"
    BRX_LIB_PATH=$PWD/$v timeout 100 python -m pytest tests/test_gpu_align.py tests/test_gpu_pipeline.py tests/test_gpu_golden.py -q -x 2>&1 | tail -2
  fi
done
