#!/bin/bash
# Round 4, GPU call 1: A/B of the four prepared branches (libraries built in worktrees, badread_amd/csrc/variants/) and the mutate-split sweep.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
V=$PWD/badread_amd/csrc/variants
{
bash tools/gpu_ab.sh "|$S" \
  "BRX_LIB_PATH=$V/libbrx_hip_qscore-compact.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_lag1.so|$S" \
  "BRX_LIB_PATH=$V/libbrx_hip_lag1-wide.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_lag1-bufstore.so|$S" "|$S" \
  "BRX_LIB_PATH=$V/libbrx_hip_qscore-compact.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_lag1.so|$S" \
  "BRX_LIB_PATH=$V/libbrx_hip_lag1-wide.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_lag1-bufstore.so|$S"
bash tools/gpu_ab.sh \
  "BRX_TAIL_READS=2730|$S" "BRX_TAIL_READS=8192|$S" "BRX_TAIL_READS=10922|$S" \
  "BRX_HEAD_READS=512|$S" "BRX_HEAD_READS=2048|$S" \
  "BRX_LANE_THRESHOLD=1500|$S" "BRX_LANE_THRESHOLD=6000|$S" "BRX_TB_WINDOW=3|$S" "|$S"
for b in qscore-compact lag1-bufstore; do
  echo "== pytest $b"
  BRX_LIB_PATH=$V/libbrx_hip_$b.so timeout 150 python -m pytest tests/test_gpu_align.py tests/test_gpu_pipeline.py tests/test_gpu_golden.py -q -x 2>&1 | tail -3
done
} > gpurun_out/r4/call1.log 2>&1
tail -40 gpurun_out/r4/call1.log
