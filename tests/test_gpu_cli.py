"""
GPU end-to-end: the `badread simulate` command line (python -m badread_amd simulate ...) on the HIP path
against the same driver run on the CPU oracle engine: stdout bytes identical, banner on stderr, and the
documented behaviours of the reference CLI (FASTQ only on stdout, seed determinism, test/test_simulate2.py:160-177).
"""
import io
import os
import subprocess
import sys

import pytest

import helpers as H
from badread_amd import simulate as S
from test_host_simulate import Args, parse_fastq

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SMALL_REF = os.path.join(HERE, 'golden', 'small_ref.fasta')


def run_cli(*extra):
    cmd = [sys.executable, '-m', 'badread_amd', 'simulate', '--reference', SMALL_REF, '--quantity', '15x',
           '--length', '400,300', '--seed', '11'] + list(extra)
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout, r.stderr.decode()


def test_cli_matches_the_oracle_driver_and_is_deterministic():
    out, err = run_cli()
    recs = parse_fastq(out)
    assert sum(len(r[1]) for r in recs) >= 15 * 3621
    assert 'Badread v' in err and 'Target read set size: 54,315 bp' in err and b'Badread' not in out
    out2, _ = run_cli()
    assert out == out2
    args = Args(quantity='15x', mean_frag_length=400.0, frag_length_stdev=300.0, mean_identity=95.0, max_identity=99.0,
                identity_stdev=2.5, error_model='nanopore2023', qscore_model='nanopore2023', seed=11,
                junk_reads=1, random_reads=1, chimeras=1)
    sink = io.BytesIO()
    S.simulate(args, output=io.StringIO(), engine=H.oracle_engine(), stdout=sink, shard=S.Shard())
    assert sink.getvalue() == out


def test_cli_models_and_identity_modes():
    out, _ = run_cli('--error_model', 'random', '--qscore_model', 'ideal', '--identity', '20,3', '--glitches', '0,0,0',
                     '--start_adapter_seq', '', '--end_adapter_seq', '')
    assert len(parse_fastq(out)) > 10
    out, _ = run_cli('--error_model', 'pacbio2021', '--qscore_model', 'pacbio2021', '--identity', '30,3')
    assert len(parse_fastq(out)) > 10


def test_sequence_fragment_drop_in():
    import random
    from badread_amd.error_model import ErrorModel
    from badread_amd.qscore_model import QScoreModel
    em, qm = ErrorModel('nanopore2023', io.StringIO()), QScoreModel('nanopore2023', io.StringIO())
    random.seed(5)
    frag = ''.join(random.choice('ACGT') for _ in range(3000))
    seq, quals, identity, by_q = S.sequence_fragment(frag, 1.0, em, qm)
    assert seq == frag and len(quals) == len(seq) and identity == 1.0          # test/test_simulate.py:45-51
    random.seed(6)
    a = S.sequence_fragment(frag, 0.9, em, qm)
    random.seed(6)
    b = S.sequence_fragment(frag, 0.9, em, qm)
    assert a == b and a[0] != frag and abs(a[2] - 0.9) < 0.05 and 0.0 < a[3] < 1.0
    assert S.sequence_fragment('', 0.9, em, qm) == ('', '', 0.0, 0.0)


def test_reference_identity_tolerances():
    """The reference's acceptance test of sequence_fragment (test/test_simulate.py:57-163) pointed at the HIP path:
    six error models x three target identities x two lengths x 20 trials, per-read and mean tolerances as written
    there; both the reads and the checking alignments go through the C-ABI."""
    eng = H.hip_engine()
    pref, _ = H.small_reference()
    H.configure(eng, pref)
    assert H.identity_tolerance_check(eng) == 6 * 3 * 2 * 20


def test_distributions_match_the_running_reference(tmp_path):
    """SURVEY.md 8d gate 3 on the HIP path: same check as tests/test_golden_oracle.py, engine = libbrx_hip.so."""
    import stat_parity
    stat_parity.check(H.hip_engine(), tmp_path)


def test_gpu_streams_and_batch_size_do_not_change_the_output():
    """Device batches in flight (engine clones on their own HIP streams) and the batch size are scheduling only."""
    base, _ = run_cli('--gpu-streams', '1')
    for extra in (('--gpu-streams', '3', '--gpu-batch', '16'), ('--gpu-streams', '5', '--gpu-batch', '64')):
        out, _ = run_cli(*extra)
        assert out == base, extra


def test_two_ranks_on_one_gpu_give_the_single_rank_bytes(tmp_path):
    """SURVEY.md 8e on the real engine: the CLI under torch.distributed.run with world size 2 -- both ranks on device 0
    (BRX_DEVICE), the 4 B/read exchange and the records to rank 0 over gloo (BRX_DIST_BACKEND: a 1-GPU box cannot run
    RCCL between two ranks of one device) -- writes the bytes of the single-process run.  Exercises run_batches, the
    sharded stop rule and collect_bytes with the HIP engine and device tensors on the sending side."""
    import socket
    single, _ = run_cli('--quantity', '40x')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    out_path = tmp_path / 'ranks.fastq'
    script = tmp_path / 'rank.py'
    script.write_text(
        'import os, sys\n'
        f'sys.path.insert(0, {REPO!r})\n'
        'from badread_amd.__main__ import main\n'
        f'sys.argv = ["badread", "simulate", "--reference", {SMALL_REF!r}, "--quantity", "40x", "--length", "400,300", "--seed", "11"]\n'
        'if int(os.environ.get("RANK", "0")) == 0:\n'
        f'    sys.stdout = open({str(out_path)!r}, "w")\n'
        'main()\n'
        'sys.stdout.flush()\n')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), BRX_DIST_BACKEND='gloo', BRX_DEVICE='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), str(script)], cwd=REPO, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert open(out_path, 'rb').read() == single


def _launch_ranks(tmp_path, world, extra_args, env_extra, out_name='ranks.fastq'):
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    out_path = tmp_path / out_name
    script = tmp_path / 'rank.py'
    argv = ["badread", "simulate", "--reference", SMALL_REF, "--quantity", "40x", "--length", "400,300", "--seed", "11"] + list(extra_args)
    script.write_text(
        'import os, sys\n'
        f'sys.path.insert(0, {REPO!r})\n'
        'from badread_amd.__main__ import main\n'
        f'sys.argv = {argv!r}\n'
        'if int(os.environ.get("RANK", "0")) == 0:\n'
        f'    sys.stdout = open({str(out_path)!r}, "w")\n'
        'main()\n'
        'sys.stdout.flush()\n')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), **env_extra)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), str(script)], cwd=REPO, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return out_path


def test_two_ranks_on_two_gpus_over_rccl_give_the_single_rank_bytes(tmp_path):
    """the `nccl` flavour of the three distributed calls of the driver -- Shard.gather_words (all_gather
    of device tensors), Shard.collect_bytes (send / recv of device tensors: the records travel GPU to GPU over xGMI) and the
    seed broadcast -- with one rank per GPU, against the single-process bytes.  Skips itself on a box with fewer than two
    devices (the builder's and the round-end 1-GPU boxes); the driver's multi-GPU box runs it before the scaling curve."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL refuses two ranks on one device)')
    single, _ = run_cli('--quantity', '40x')
    out_path = _launch_ranks(tmp_path, 2, [], {})
    assert open(out_path, 'rb').read() == single
    # and every rank writing its own shard (no record leaves its GPU for another)
    prefix = str(tmp_path / 'shard')
    _launch_ranks(tmp_path, 2, ['--output-shards', prefix], {}, out_name='unused.fastq')
    import test_host_simulate as THS
    got, parts = THS.reassemble(prefix, 2)
    assert got == single and len(parts[0]) == len(parts[1])


def test_output_shards_with_the_hip_engine_on_one_gpu(tmp_path):
    """--output-shards on the real engine: two ranks on device 0 (exchange over gloo), each writing the records of its own
    reads from its own pinned ring; the files put back batch by batch, rank after rank, are the single-process bytes, and
    with --gzip-device the members each rank packed on the GPU decompress to the same records."""
    import gzip
    import test_host_simulate as THS
    single, _ = run_cli('--quantity', '40x')
    prefix = str(tmp_path / 'shard')
    _launch_ranks(tmp_path, 2, ['--output-shards', prefix], dict(BRX_DIST_BACKEND='gloo', BRX_DEVICE='0'))
    got, _ = THS.reassemble(prefix, 2)
    assert got == single
    prefix = str(tmp_path / 'shardz')
    _launch_ranks(tmp_path, 2, ['--output-shards', prefix, '--gzip-device'], dict(BRX_DIST_BACKEND='gloo', BRX_DEVICE='0'))
    texts = [gzip.decompress(open(f'{prefix}.{r}.fastq.gz', 'rb').read()) for r in range(2)]
    assert sorted(THS.parse_fastq(b''.join(texts))) == sorted(THS.parse_fastq(single))
