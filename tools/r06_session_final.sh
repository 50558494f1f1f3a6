cd ${GRAFT_REPO_ROOT:-/root/repo}
export BRX_ROUND_TAG=r06
bash tools/profile_round.sh r06 human "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" > gpurun_out/r06_profile_round.out 2>&1
bash tools/profile_round.sh r06_hifi hifi "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" > gpurun_out/r06_hifi_profile_round.out 2>&1
bash tools/profile_round.sh r06_kpn kpn "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" > gpurun_out/r06_kpn_profile_round.out 2>&1
python tools/pmc_traffic.py gpurun_out/r06_pmc_per_kernel.csv 65536 --all human > gpurun_out/r06_pmc_traffic.json 2> gpurun_out/r06_pmc_traffic.err
tail -3 gpurun_out/r06_profile_round.out | cut -c1-300
