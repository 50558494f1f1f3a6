#!/bin/bash
# First GPU call of the next round (about 4 GPU-minutes): everything the last round prepared but could not measure.
#   1. the split between the two mutate routes at the current aligner speed (environment only: nothing to build)
#   2. branch round4/qscore-compact against this tree: build it in a worktree HERE first and pass its library:
#        git worktree add /tmp/r4 round4/qscore-compact && (cd /tmp/r4 && python -m badread_amd.build) &&
#        cp /tmp/r4/badread_amd/csrc/libbrx_hip.so badread_amd/csrc/variants/libbrx_hip_qscore.so
# Usage (from the repo root, through gpurun):  bash tools/first_call_next_round.sh > gpurun_out/first_call.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="--steps 2"
bash tools/gpu_ab.sh "|$S" \
  "BRX_TAIL_READS=2730|$S" "BRX_TAIL_READS=8192|$S" "BRX_TAIL_READS=10922|$S" \
  "BRX_HEAD_READS=512|$S" "BRX_HEAD_READS=2048|$S" \
  "BRX_LANE_THRESHOLD=1500|$S" "BRX_LANE_THRESHOLD=6000|$S" \
  "BRX_TB_WINDOW=3|$S" "|$S"
v=badread_amd/csrc/variants/libbrx_hip_qscore.so
if [ -f $v ]; then
  bash tools/gpu_ab.sh "BRX_LIB_PATH=$PWD/$v|$S" "|$S" "BRX_LIB_PATH=$PWD/$v|$S"
  BRX_LIB_PATH=$PWD/$v timeout 100 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py -q -x 2>&1 | tail -2
fi
