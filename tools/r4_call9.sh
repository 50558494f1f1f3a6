#!/bin/bash
# Round 4, GPU call 9: lane windows by band class + hot-row index against the previous commit's library (early set off).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
V=$PWD/badread_amd/csrc/variants
{
echo "== parity first"
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh "|$S" "BRX_LIB_PATH=$V/libbrx_hip_prev.so BRX_EARLY_FRAC=0|$S" "|$S" "BRX_LIB_PATH=$V/libbrx_hip_prev.so BRX_EARLY_FRAC=0|$S" "BRX_LANE_WAVES=1024|$S" "BRX_LANE_THRESHOLD=1500|$S" "|$S --steps 6"
} > gpurun_out/r4/call9.log 2>&1
tail -30 gpurun_out/r4/call9.log | cut -c1-300
