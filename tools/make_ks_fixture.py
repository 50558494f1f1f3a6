"""
Generate tests/golden/ks_reference.npz: per-read statistics of >= 10 000 reads simulated by the UNMODIFIED
REFERENCE (/root/reference, `badread simulate`, its own Mersenne-Twister streams), for the statistical parity
gate of SURVEY.md section 8d(3): read length, read identity, error-free length - length, mean qscore per read,
and the qscore histogram.  The reference imports `edlib`; oracle/shim/edlib stands in for it (as in
tools/make_golden.py).  The parity test draws the same number of reads from OUR path (oracle on CPU, HIP on
the GPU: both bit-identical) under the same parameters and runs two-sample KS tests at alpha = 0.01.

Run:  python tools/make_ks_fixture.py [workers]      (needs /root/reference; ~3 minutes on 8 cores)
"""
import collections
import io
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden', 'ks_reference.npz')

# the workload: one 300 kb circular contig + one 20 kb linear contig, reads of mean 2.5 kb (short enough that the
# Python reference manages 10 000 of them in minutes), all other parameters at the reference's defaults
REF_SEED = 20260926
LENGTH = '2500,2000'
IDENTITY = '95,99,2.5'
READS_PER_WORKER = 1400
QUANTITY_PER_WORKER = 3_600_000          # ~1440 reads of mean ~2.5 kb


def reference_sequences():
    rng = np.random.default_rng(REF_SEED)
    seqs = collections.OrderedDict()
    for name, n in (('ring', 300000), ('stick', 20000)):
        seqs[name] = np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, n)].tobytes().decode()
    return seqs, {'ring': True, 'stick': False}


def write_fasta(path):
    seqs, circ = reference_sequences()
    with open(path, 'w') as f:
        for name, s in seqs.items():
            f.write(f'>{name} circular={"true" if circ[name] else "false"}\n')
            for i in range(0, len(s), 80):
                f.write(s[i:i + 80] + '\n')


HEADER = re.compile(rb'length=(\d+) error-free_length=(\d+) read_identity=([0-9.]+)%')


def parse(fastq):
    """-> arrays length, identity (%), error_free - length, mean qscore; histogram of qscores (0..93)"""
    lines = fastq.split(b'\n')
    L, I, D, Q = [], [], [], []
    hist = np.zeros(94, dtype=np.int64)
    for i in range(0, len(lines) - 3, 4):
        m = HEADER.search(lines[i])
        if not m:
            continue
        q = np.frombuffer(lines[i + 3], dtype=np.uint8).astype(np.int64) - 33
        L.append(int(m.group(1))); I.append(float(m.group(3))); D.append(int(m.group(2)) - int(m.group(1)))
        Q.append(float(q.mean()) if len(q) else 0.0)
        hist += np.bincount(q, minlength=94)[:94]
    return np.array(L, np.int32), np.array(I, np.float32), np.array(D, np.int32), np.array(Q, np.float32), hist


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    tmp = tempfile.mkdtemp(prefix='brx_ks_')
    fasta = os.path.join(tmp, 'ref.fasta')
    write_fasta(fasta)
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(REPO, 'oracle', 'shim'), REFERENCE, os.path.join(REPO, 'oracle'), REPO])
    procs = []
    for w in range(workers):
        cmd = [sys.executable, '-c', 'import badread.__main__ as m; m.main()', 'simulate', '--reference', fasta,
               '--quantity', str(QUANTITY_PER_WORKER), '--length', LENGTH, '--identity', IDENTITY, '--seed', str(1000 + w)]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=tmp))
    parts = [parse(p.communicate()[0]) for p in procs]
    L, I, D, Q = (np.concatenate([p[i] for p in parts]) for i in range(4))
    hist = sum(p[4] for p in parts)
    np.savez_compressed(OUT, length=L, identity=I, trimmed=D, mean_q=Q, qhist=hist,
                        meta=np.array([f'Badread 0.4.2 CLI (unmodified, edlib = oracle/shim), reference seed {REF_SEED}, '
                                       f'--length {LENGTH} --identity {IDENTITY}, seeds 1000..{1000 + workers - 1}, '
                                       f'--quantity {QUANTITY_PER_WORKER} each']))
    print(f'{len(L)} reads, {int(L.sum())} bases -> {OUT}')


if __name__ == '__main__':
    main()
