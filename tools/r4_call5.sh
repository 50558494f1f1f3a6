#!/bin/bash
# Round 4, GPU call 5: band classes of the bulk final stage on three streams; more, smaller batches in flight; scan for rare routes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
{
echo "== parity first"
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py tests/test_gpu_align.py -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh "|$S" "BRX_FIN_SPREAD=0|$S" "BRX_TAIL_READS=8192|$S" "|$S --streams 8 --scratch-gb 30" "|$S --streams 12 --scratch-gb 20" "|$S --streams 5" \
  "|$S" "BRX_FIN_SPREAD=0|$S" "BRX_TAIL_READS=8192|$S" "|$S --streams 8 --scratch-gb 30" "BRX_TAIL_READS=8192 BRX_HEAD_READS=384|$S"
echo "== rare routes"
timeout 400 python tools/find_rare_routes.py 70 110 2> gpurun_out/r4/rare_routes.err
tail -3 gpurun_out/r4/rare_routes.err
} > gpurun_out/r4/call5.log 2>&1
tail -30 gpurun_out/r4/call5.log | cut -c1-300
