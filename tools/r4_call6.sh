#!/bin/bash
# Round 4, GPU call 6: the early set of the final stage (three sets), fraction sweep, arena view.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
{
echo "== parity first"
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py tests/test_gpu_align.py -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh "|$S" "BRX_EARLY_FRAC=0|$S" "BRX_EARLY_FRAC=0.6|$S" "BRX_EARLY_FRAC=0.9|$S" "BRX_EARLY_FRAC=0.7 BRX_TAIL_READS=10922|$S" "|$S --streams 5" \
  "|$S" "BRX_EARLY_FRAC=0|$S" "BRX_EARLY_FRAC=0.6|$S" "BRX_EARLY_FRAC=0.9|$S" "|$S --streams 5" "|$S --streams 7 --scratch-gb 35"
echo "== arena view"
BRX_DEBUG=1 timeout 200 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --streams 1 --reads-per-step 65536 2>&1 | grep -E "final set|set [0-9]:" | head -20
} > gpurun_out/r4/call6.log 2>&1
tail -30 gpurun_out/r4/call6.log | cut -c1-300
