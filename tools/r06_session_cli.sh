cd ${GRAFT_REPO_ROOT:-/root/repo}
export BRX_ROUND_TAG=r06
for i in 1 2 3; do
bash tools/cli_30x.sh 30x 2>&1 | head -1 | cut -c1-1500; cp gpurun_out/r06_cli_30x.json gpurun_out/r06_cli_30x_run$i.json
done
bash tools/cli_30x.sh 30x "--error_model pacbio2021 --qscore_model pacbio2021 --identity 30,3" 2>&1 | head -1 | cut -c1-1500
bash tools/cli_30x.sh 30x "--gzip-device" 2>&1 | head -1 | cut -c1-1500
