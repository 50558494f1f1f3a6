cd ${GRAFT_REPO_ROOT:-/root/repo}
{
bash tools/gpu_ab.sh "X=1|--steps 3" "X=1|--steps 3 --scratch-gb 36" "X=1|--steps 3 --streams 7 --scratch-gb 34" "X=1|--steps 3" "BRX_DEBUG=1|--steps 1 --warmup 0 --streams 1 --reads-per-step 65536"
echo "== fullsize"; timeout 2400 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "human or hifi or off_default or cli_sizes" 2>&1 | tail -5
} > gpurun_out/r06m.log 2>&1
grep -E "^\[|passed|failed|final set" gpurun_out/r06m.log | cut -c1-400 | head -30
