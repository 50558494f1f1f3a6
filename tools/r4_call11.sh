#!/bin/bash
# Round 4, GPU call 11: register budgets of the one- and two-word final aligners (occupancy against spills).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
V=$PWD/badread_amd/csrc/variants
{
bash tools/gpu_ab.sh "|$S" "BRX_LIB_PATH=$V/libbrx_hip_fin1_6.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_fin2_5.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_fin16_25.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_fin18.so|$S" \
  "|$S" "BRX_LIB_PATH=$V/libbrx_hip_fin1_6.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_fin2_5.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_fin16_25.so|$S" "BRX_LIB_PATH=$V/libbrx_hip_fin18.so|$S"
} > gpurun_out/r4/call11.log 2>&1
tail -30 gpurun_out/r4/call11.log | cut -c1-220
