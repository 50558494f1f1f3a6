#!/bin/bash
# Round 4, GPU call 2: the read staged in LDS (k_mutate_seg<false>) and the register budgets of the run-to-completion launches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
V=$PWD/badread_amd/csrc/variants
{
echo "== parity first"
timeout 300 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh "|$S" "BRX_STAGE_WORDS=0|$S" "BRX_LIB_PATH=$V/libbrx_hip_stage4096.so|$S" \
  "BRX_RUN_WPS_HEAD=4|$S" "BRX_RUN_WPS_TAIL=2|$S" "BRX_HEAD_READS=512|$S" "BRX_HEAD_READS=512 BRX_RUN_WPS_TAIL=2|$S" \
  "BRX_SEG_WAVES_PER_CU=12|$S" "BRX_SEG_WAVES_PER_CU=16|$S" \
  "|$S" "BRX_STAGE_WORDS=0|$S" "BRX_LIB_PATH=$V/libbrx_hip_stage4096.so|$S"
echo "== phase profile (staged)"
timeout 200 python tools/phase_profile.py 16384 2>&1 | tail -12
echo "== phase profile (BRX_STAGE_WORDS=0)"
BRX_STAGE_WORDS=0 timeout 200 python tools/phase_profile.py 16384 2>&1 | tail -12
} > gpurun_out/r4/call2.log 2>&1
tail -30 gpurun_out/r4/call2.log
