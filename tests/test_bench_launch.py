"""
bench.py's launch contract: `python bench.py --gpus N` starts N ranks itself, a WORLD_SIZE that
contradicts --gpus is refused, and rank 0 prints ONE JSON line whose value aggregates all ranks.  Runs on the CPU
checker engine over gloo (--cpu-engine: a dry run of launch / sharding / reporting; the line is marked INVALID).
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')
COMMON = ['--cpu-engine', '--steps', '1', '--warmup', '0', '--reads-per-step', '128', '--streams', '2', '--workload', 'human',
          '--ref-scale', '0.002']


def run(args, env=None, tmp=None):
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args + ['--ref-dir', str(tmp)], env=e, capture_output=True, text=True, timeout=600)


def last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_aggregates(tmp_path):
    one = last_json(run(COMMON + ['--gpus', '1'], tmp=tmp_path).stdout)
    two_run = run(COMMON + ['--gpus', '2'], tmp=tmp_path)
    assert two_run.returncode == 0, two_run.stderr[-3000:]
    two = last_json(two_run.stdout)
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2
    assert 'INVALID' in two and two['scaling'] == 'weak'
    assert two['config']['reads_per_step_per_gpu'] == one['config']['reads_per_step_per_gpu'] == 128
    # weak scaling: the job's bases are the sum over ranks (rank 0 alone simulates about half of them)
    total = two['value'] * two['ms_per_step'] * 1e-3 * two['steps']
    rank0 = two['config']['bases_per_step_per_gpu'] * two['steps']
    assert 1.2 * rank0 < total < 4.0 * rank0, (total, rank0)       # 128 reads of 15 +- 13 kb per rank: the other rank's bases are in the sum
    assert two['config']['reference_contigs'] == 24 and two['config']['reference_non_acgt_runs'] == 28


def test_world_size_that_contradicts_gpus_is_refused(tmp_path):
    r = run(COMMON + ['--gpus', '2'], env={'WORLD_SIZE': '3', 'RANK': '0', 'LOCAL_RANK': '0'}, tmp=tmp_path)
    assert r.returncode != 0 and 'refusing' in (r.stderr + r.stdout)


def test_gpus_8_dry_run_over_gloo(tmp_path):
    """The launch the driver uses for its scaling curve (--gpus 8), on the CPU checker engine over gloo: eight ranks start,
    shard the read-index space, reduce their three scalars and rank 0 prints one line whose bases are the sum of all ranks."""
    r = run(['--cpu-engine', '--steps', '1', '--warmup', '0', '--reads-per-step', '16', '--streams', '1', '--workload', 'human',
             '--ref-scale', '0.002', '--gpus', '8'], tmp=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    line = last_json(r.stdout)
    assert line['n_gpus'] == 8 and 'INVALID' in line and line['scaling'] == 'weak'
    total = line['value'] * line['ms_per_step'] * 1e-3 * line['steps']
    rank0 = line['config']['bases_per_step_per_gpu'] * line['steps']
    assert total > 3.0 * rank0, (total, rank0)          # 16 reads of 15 +- 13 kb per rank: eight ranks in the sum
