#!/bin/bash
# Round 4, GPU call 10: head / tail / wave-count defaults on the final kernels.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
{
bash tools/gpu_ab.sh "|$S" "BRX_HEAD_READS=256|$S" "BRX_HEAD_READS=768|$S" "BRX_TAIL_READS=6554|$S" "BRX_TAIL_READS=10922|$S" "BRX_WAVES_PER_CU=12|$S" "BRX_WAVES_PER_CU=20|$S" "BRX_SEG_WAVES_PER_CU=6|$S" "BRX_SEG_WAVES_PER_CU=10|$S" "|$S" "BRX_LANE_THRESHOLD=6000|$S" "BRX_HEAD_READS=384 BRX_TAIL_READS=6554|$S"
} > gpurun_out/r4/call10.log 2>&1
tail -30 gpurun_out/r4/call10.log | cut -c1-200
