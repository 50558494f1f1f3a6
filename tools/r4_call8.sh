#!/bin/bash
# Round 4, GPU call 8: early set with fewer batches in flight; hot-row index of k_fin_qscore; instruction counts of the tree.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
{
echo "== parity first"
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh "|$S" "BRX_EARLY_FRAC=0|$S" "|$S --streams 4" "BRX_EARLY_FRAC=0|$S --streams 4" "|$S --streams 3" "BRX_EARLY_FRAC=0|$S --streams 3" \
  "|$S" "BRX_EARLY_FRAC=0|$S" "|$S --streams 4" "BRX_EARLY_FRAC=0|$S --streams 4"
} > gpurun_out/r4/call8.log 2>&1
bash tools/profile_round.sh r04h human "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" > gpurun_out/r4/call8_profile.log 2>&1
tail -30 gpurun_out/r4/call8.log | cut -c1-300
