#!/bin/bash
# Round 4, GPU call 13: packed-window passes instead of most of the in-place tail (lane threshold above the tail size).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
{
bash tools/gpu_ab.sh "|$S" "BRX_LANE_THRESHOLD=8192 BRX_TAIL_READS=1024|$S" "BRX_LANE_THRESHOLD=8192 BRX_TAIL_READS=2048|$S" "BRX_LANE_THRESHOLD=8192 BRX_TAIL_READS=512|$S" "BRX_LANE_THRESHOLD=12000 BRX_TAIL_READS=1024|$S" "BRX_LANE_THRESHOLD=6000 BRX_TAIL_READS=1024|$S" "BRX_LANE_THRESHOLD=8192 BRX_TAIL_READS=4096|$S" \
  "|$S" "BRX_LANE_THRESHOLD=8192 BRX_TAIL_READS=1024|$S" "BRX_LANE_THRESHOLD=16384 BRX_TAIL_READS=1024|$S"
} > gpurun_out/r4/call13.log 2>&1
tail -30 gpurun_out/r4/call13.log | cut -c1-330
