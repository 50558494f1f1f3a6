#!/bin/bash
# One GPU session (= one gpurun call) as a list of steps; everything goes to gpurun_out/<tag>.log, the tail is printed.
#   bash tools/gpu_session.sh <tag> <step> ...
# Steps:
#   pytest:<args>            python -m pytest <args> -q -x              (timeout 900 s)
#   ab:<ENV=1 ENV2=..>|<bench args>   one bench configuration, one summary line (tools/gpu_ab.sh)
#   debug:<ENV ..>|<bench args>       the same run with BRX_DEBUG=1, the final-stage set lines only
#   pmc:<tag>|<workload>|<counters>   one counter pass over two serial device batches (tools/profile_round.sh layout), summarised per kernel
#   sh:<command>             anything else
# (Round 4's fifteen one-off scripts tools/r4_call*.sh were this with the steps written out; they are in the history at 1ec8716.)
root=${GRAFT_REPO_ROOT:-/root/repo}; cd "$root"
tag=$1; shift
mkdir -p gpurun_out
log=gpurun_out/$tag.log
{
for step in "$@"; do
  kind=${step%%:*}; body=${step#*:}
  echo "== $step"
  case $kind in
    pytest) timeout 900 python -m pytest $body -q -x 2>&1 | tail -6 ;;
    ab) bash tools/gpu_ab.sh "$body" ;;
    debug) envs=${body%%|*}; args=${body#*|}
           env $envs BRX_DEBUG=1 timeout 300 python bench.py --cpu-seconds 0 $args 2>&1 | grep -E "final set" | head -12 ;;
    pmc) t=${body%%|*}; rest=${body#*|}; wl=${rest%%|*}; ctrs=${rest#*|}
         ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$root/gpurun_out/${t}_pmc" -o p -- \
             python "$root/bench.py" --workload $wl --steps 1 --warmup 1 --streams 1 --reads-per-step 65536 --cpu-seconds 0 > "$root/gpurun_out/${t}_pmc.json" 2> "$root/gpurun_out/${t}_pmc.err" )
         python tools/pmc_summary.py gpurun_out/${t}_pmc/*counter_collection.csv > gpurun_out/${t}_pmc_per_kernel.csv 2>> gpurun_out/${t}_pmc.err
         rm -rf gpurun_out/${t}_pmc
         head -40 gpurun_out/${t}_pmc_per_kernel.csv ;;
    sh) bash -c "$body" ;;
    *) echo "unknown step kind $kind" ;;
  esac
done
} > "$log" 2>&1
tail -60 "$log" | cut -c1-420
