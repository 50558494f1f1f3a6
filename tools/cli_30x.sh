#!/bin/bash
# The product command on BASELINE.json configs[3] as a user runs it: `badread simulate --reference GRCh38-like.fa --quantity 30x`
# in a fresh process, FASTQ to /dev/null; wall time of the whole command, of the read loop (run_batches) and of every start-up step
# (startup_timing of badread_amd.simulate under BRX_DRIVER_TIMING: interpreter + imports, reference, models, engine, tables on the device).
#   bash tools/cli_30x.sh [quantity] [extra arguments]      -> gpurun_out/<BRX_ROUND_TAG>_cli_<tag>.json
#   BRX_CLI_RANKS=N bash tools/cli_30x.sh ...   the same job as N ranks under torch.distributed.run with --output-shards (every rank its own file,
#                   here links to /dev/null): the strong-scaling shape of the product command.  On a box with fewer GPUs
#                   than ranks the ranks share device 0 over gloo (BRX_DEVICE=0 BRX_DIST_BACKEND=gloo: an accounting run, not a rate), with
#                   --gpu-streams cut so that their arenas fit.
cd ${GRAFT_REPO_ROOT:-/root/repo}; out=gpurun_out; q=${1:-30x}; extra=${2:-}; ranks=${BRX_CLI_RANKS:-1}; tag=${q}${extra//[^a-z0-9]/}$([ $ranks -gt 1 ] && echo _ranks$ranks)
mkdir -p $out
fa=$(python -c "import sys; sys.path.insert(0,'tools'); import bench; print(bench.reference_fasta('human', bench.default_ref_dir()))")
# the run before the timed one: packs the FASTA once (the packed form stays in the user cache, as for any second run on a genome)
python -m badread_amd simulate --reference $fa --quantity 1x --seed 1 > /dev/null 2> $out/cli_warm.err
# The timed run starts on a GPU that has been idle for a while, as a user's does: the driver clears the memory the process before gave
# back, and a process that maps 240 GB right behind another one waits for that (clone arenas 10-21 s of thread time back to back,
# 2.4 s after 20 s: profiles/r05j_*).  BRX_CLI_IDLE_S=0 measures the back-to-back case.
sleep ${BRX_CLI_IDLE_S:-20}
t0=$(date +%s.%N)
if [ $ranks -gt 1 ]; then
  ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
  shards=/tmp/brx_cli_shards_$$                # the shard files are links to /dev/null: 187 GB of text must not land in memory or on a disk
  for r in $(seq 0 $((ranks - 1))); do ln -sf /dev/null ${shards}.$r.fastq; ln -sf /dev/null ${shards}.$r.parts; done
  envs=""; streams=""
  if [ $ngpu -lt $ranks ]; then envs="BRX_DEVICE=0 BRX_DIST_BACKEND=gloo"; streams="--gpu-streams $(( 6 / ranks > 0 ? 6 / ranks : 1 ))"; fi
  env $envs BRX_T0=$t0 BRX_DRIVER_TIMING=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$ranks --master-addr 127.0.0.1 --master-port 29531 \
      -m badread_amd simulate --reference $fa --quantity $q --seed 42 --output-shards $shards $streams $extra > /dev/null 2> $out/cli_${tag}.err
  rc=$?
  ls -l ${shards}.* > $out/cli_${tag}.shards 2>/dev/null; rm -f ${shards}.*
else
BRX_T0=$t0 BRX_DRIVER_TIMING=1 timeout 400 python -m badread_amd simulate --reference $fa --quantity $q --seed 42 $extra > /dev/null 2> $out/cli_${tag}.err
rc=$?
fi
t1=$(date +%s.%N)
grep -a driver_timing $out/cli_${tag}.err | tail -1 > $out/cli_${tag}.timing
grep -a startup_timing $out/cli_${tag}.err | tail -1 > $out/cli_${tag}.startup
python - <<PY
import ast, json
def parse(path):
    line = open(path).read().strip()
    return ast.literal_eval(line.split(' ', 1)[1]) if line else {}
t, s = parse('$out/cli_${tag}.timing'), parse('$out/cli_${tag}.startup')
wall = $t1 - $t0
loop = t.get('run_batches_seconds', 0.0)
res = {'command': 'python -m badread_amd simulate --reference grch38_like.fa --quantity $q --seed 42 $extra > /dev/null', 'ranks': $ranks, 'rc': $rc, 'wall_seconds': round(wall, 2),
       'bases': t.get('bases'), 'reads': t.get('reads'), 'gbases_per_s_whole_command': round(t.get('bases', 0) / wall / 1e9, 3),
       'gbases_per_s_read_loop': round(t.get('bases', 0) / max(loop, 1e-9) / 1e9, 3),
       'fixed_cost_seconds': round(wall - loop, 2), 'startup_timing': s, 'driver_timing': t}
print(json.dumps(res))
open('$out/${BRX_ROUND_TAG:-r06}_cli_${tag}.json', 'w').write(json.dumps(res, indent=1))
PY
tail -c 400 $out/cli_${tag}.err | tr '\r' '\n' | tail -4
