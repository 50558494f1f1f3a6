#!/bin/bash
# Round 4, GPU call 3: lookup-order error-model tables + changed map in LDS (no fence per change) against the staging-only library; PMC pass.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
V=$PWD/badread_amd/csrc/variants
{
echo "== parity first"
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py tests/test_gpu_align.py tests/test_gpu_models.py -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh "|$S" "BRX_LIB_PATH=$V/libbrx_hip_stageonly.so|$S" "BRX_RUN_WPS_HEAD=4|$S" "BRX_HEAD_READS=512|$S" "BRX_HEAD_READS=512 BRX_RUN_WPS_HEAD=4|$S" \
  "|$S" "BRX_LIB_PATH=$V/libbrx_hip_stageonly.so|$S" "BRX_RUN_WPS_HEAD=4|$S" "BRX_HEAD_READS=512|$S" "BRX_HEAD_READS=512 BRX_RUN_WPS_HEAD=4|$S" "BRX_HEAD_READS=256 BRX_RUN_WPS_HEAD=4|$S"
echo "== phase profile"
timeout 200 python tools/phase_profile.py 16384 2>&1 | tail -9
} > gpurun_out/r4/call3.log 2>&1
bash tools/profile_round.sh r04c human "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/r4/call3_profile.log 2>&1
tail -30 gpurun_out/r4/call3.log | cut -c1-300
