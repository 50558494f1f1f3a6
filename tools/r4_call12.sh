#!/bin/bash
# Round 4, GPU call 12: cold kernel arguments in memory, small LDS for the in-place kernel, 512-wave k_win_wave -- against the committed library.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
V=$PWD/badread_amd/csrc/variants
{
echo "== parity first"
timeout 300 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh "|$S" "BRX_LIB_PATH=$V/libbrx_hip_prev.so|$S" "|$S" "BRX_LIB_PATH=$V/libbrx_hip_prev.so|$S" "|$S" "BRX_LIB_PATH=$V/libbrx_hip_prev.so|$S"
} > gpurun_out/r4/call12.log 2>&1
tail -30 gpurun_out/r4/call12.log | cut -c1-260
