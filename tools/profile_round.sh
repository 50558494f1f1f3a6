#!/bin/bash
# Profiling passes of one round, run on the GPU box through gpurun:
#   bash tools/profile_round.sh <tag> [workload] [counter set]...   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# Pass 1: kernel trace + stats of the default bench command.  Passes 2..: PMC counters, each in its own run with
# --kernel-trace only (never with the hip/hsa/runtime trace domains), on ONE device batch at a time (--streams 1: counter
# collection serialises the kernels anyway).
set -u
tag=${1:-rXX}
wl=${2:-human}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/${tag}_trace" -o t -- python "$root/bench.py" --workload $wl --steps 4 --warmup 1 --cpu-seconds 0 > "$out/${tag}_bench_profiled.json" 2> "$out/${tag}_trace.err"
python "$root/tools/export_prof.py" "$(ls "$out/${tag}_trace"/*.db | head -1)" "$out/${tag}_bench_kernel_stats.csv" >> "$out/${tag}_trace.err" 2>&1
i=0
for ctrs in "${@:3}"; do
  i=$((i + 1))
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out/${tag}_pmc$i" -o p -- \
      python "$root/bench.py" --workload $wl --steps 1 --warmup 1 --streams 1 --reads-per-step 65536 --cpu-seconds 0 > "$out/${tag}_pmc$i.json" 2> "$out/${tag}_pmc$i.err"
done
if [ $i -gt 0 ]; then
  python "$root/tools/pmc_summary.py" "$out/${tag}"_pmc*/*counter_collection.csv > "$out/${tag}_pmc_per_kernel.csv" 2>> "$out/${tag}_trace.err"
  python "$root/tools/valu_per_base.py" "$out/${tag}_pmc_per_kernel.csv" "$out/${tag}_pmc1.json" $wl "$tag" > "$out/${tag}_valu_per_base.json" 2>> "$out/${tag}_trace.err"
fi
rm -rf "$out/${tag}"_pmc*/ "$out/${tag}_trace"
tail -c 400 "$out/${tag}_bench_profiled.json"; head -14 "$out/${tag}_bench_kernel_stats.csv"; cat "$out/${tag}_valu_per_base.json" 2>/dev/null | head -30
