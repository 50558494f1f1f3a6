cd ${GRAFT_REPO_ROOT:-/root/repo}
{
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh "X=1|--steps 3" "BRX_LIB_PATH=badread_amd/csrc/variants/libbrx_before.so|--steps 3" "X=1|--steps 3" "BRX_LIB_PATH=badread_amd/csrc/variants/libbrx_before.so|--steps 3" "X=1|--steps 3 --workload kpn"
echo "== fullsize"; timeout 2400 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "human" 2>&1 | tail -3
} > gpurun_out/r06n.log 2>&1
grep -E "^\[|passed|failed" gpurun_out/r06n.log | cut -c1-330
