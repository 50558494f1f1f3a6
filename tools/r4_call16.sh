#!/bin/bash
# Round 4, GPU call 16: final alignments with narrow bands one read per lane (k_fin_lanes): parity, then A/B on configs[4] and configs[3].
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
S="--steps 3"
{
echo "== parity first"
timeout 300 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py -q -x 2>&1 | tail -3
timeout 400 python -m pytest tests/test_gpu_fullsize.py -q -x -k "hifi" 2>&1 | tail -3
bash tools/gpu_ab.sh "|$S --workload hifi" "BRX_FIN_LANES=0|$S --workload hifi" "|$S --workload hifi" "|$S" "BRX_FIN_LANES=0|$S" "|$S"
BRX_DEBUG=1 timeout 200 python bench.py --workload hifi --steps 1 --warmup 0 --cpu-seconds 0 --streams 1 --reads-per-step 65536 2>&1 | grep -E "final set" | head -6
} > gpurun_out/r4/call16.log 2>&1
tail -30 gpurun_out/r4/call16.log | cut -c1-300
