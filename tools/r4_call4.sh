#!/bin/bash
# Round 4, GPU call 4: proposals as three words + cooperative expansion, the one-round parking pass, odd map; defaults HEAD_READS 512, head budget 4 waves.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
V=$PWD/badread_amd/csrc/variants
{
echo "== parity first"
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py tests/test_gpu_align.py -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh "|$S" "BRX_LIB_PATH=$V/libbrx_hip_stageonly.so BRX_HEAD_READS=512|$S" "BRX_LIB_PATH=$V/libbrx_hip_segwps4.so|$S" "BRX_RUN_WPS_HEAD=2|$S" "BRX_SEG_WAVES_PER_CU=12|$S" "BRX_TAIL_READS=2730|$S" "BRX_TAIL_READS=8192|$S" \
  "|$S" "BRX_LIB_PATH=$V/libbrx_hip_stageonly.so BRX_HEAD_READS=512|$S" "BRX_LIB_PATH=$V/libbrx_hip_segwps4.so|$S" "BRX_RUN_WPS_HEAD=2|$S" "BRX_SEG_WAVES_PER_CU=12|$S" "BRX_STAGE_WORDS=0|$S"
echo "== phase profile"
timeout 200 python tools/phase_profile.py 16384 2>&1 | tail -9
} > gpurun_out/r4/call4.log 2>&1
tail -30 gpurun_out/r4/call4.log | cut -c1-300
