"""
The reference's own CPU path as the stated baseline -- TEST / MEASUREMENT
INFRASTRUCTURE, runs only where /root/reference exists (this container; not the GPU box).

    python tools/ref_cpu_baseline.py [--workload human|hifi|kpn] [--mbases-per-worker 1.0] [--workers P]

Runs the UNMODIFIED /root/reference (badread.simulate.simulate) with oracle/shim/edlib on the path (bit-vector Myers,
so the alignment cost is representative of real edlib) as P single-threaded worker processes, one per host core, on the
bench workload's parameters.  For the 3.1 Gb configurations the CPU side uses a 50 Mb slice of the same synthetic
construction (tools/synth_refs.py, scale 0.0162): pure-Python loading of 3.1 Gb takes minutes and the per-read cost does
not depend on the genome size.  Each worker simulates `--mbases-per-worker` with seed 42+i, FASTQ to /dev/null.
What is timed is the steady state INSIDE the read loop (simulate.py:63-86): the clock starts at the loop's first
progress line and stops at its last; model and reference loading are reported separately.  The result is stored in
profiles/cpu_reference_baseline.json, which bench.py embeds in its `cpu_baseline.reference` field beside the rate of
the C port it measures live on the GPU box.
"""
import argparse
import json
import os
import platform
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = '/root/reference'
SLICE_SCALE = 50e6 / 3088269832.0

WORKER = r'''
import io, json, os, sys, time, types
ref_path, quantity, seed, workload, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
t_import = time.perf_counter()
import badread.simulate as S                    # the unmodified reference
class Clock(object):
    def __init__(self): self.first = None; self.last = None
    def write(self, text):
        if 'Simulating:' in text:
            now = time.perf_counter()
            if self.first is None: self.first = now
            self.last = now
        return len(text)
    def flush(self): pass
args = types.SimpleNamespace(reference=ref_path, quantity=quantity, mean_frag_length=15000.0, frag_length_stdev=13000.0,
    mean_identity=95.0, max_identity=99.0, identity_stdev=2.5, error_model='nanopore2023', qscore_model='nanopore2023',
    seed=seed, start_adapter='90,60', end_adapter='50,20', start_adapter_seq='AATGTACTTCGTTCAGTTACGTATTGCT',
    end_adapter_seq='GCAATACGTAACTGAACGAAGT', junk_reads=1.0, random_reads=1.0, chimeras=1.0, glitch_rate=10000.0,
    glitch_size=25.0, glitch_skip=25.0, small_plasmid_bias=False)
if workload == 'hifi':                         # configs[4]: --error_model pacbio2021 --qscore_model pacbio2021 --identity 30,3
    args.error_model = args.qscore_model = 'pacbio2021'
    args.mean_identity, args.identity_stdev, args.max_identity = 30.0, 3.0, None
clock = Clock()
class Counter(io.TextIOBase):
    def __init__(self): self.n = 0; self.line = 0
    def write(self, text): self.n += len(text); return len(text)
sink = Counter()
sys.stdout = sink
t0 = time.perf_counter()
S.simulate(args, output=clock)
t1 = time.perf_counter()
sys.stdout = sys.__stdout__
loop_s = clock.last - clock.first
json.dump({'loop_seconds': loop_s, 'setup_seconds': (clock.first - t0), 'total_seconds': t1 - t0, 'fastq_chars': sink.n}, open(out_path, 'w'))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='human', choices=('human', 'hifi', 'kpn'))
    ap.add_argument('--mbases-per-worker', type=float, default=1.0)
    ap.add_argument('--workers', type=int, default=os.cpu_count())
    ap.add_argument('--out', default=os.path.join(REPO, 'profiles', 'cpu_reference_baseline.json'))
    a = ap.parse_args()
    if not os.path.isdir(REFERENCE):
        sys.exit(f'{REFERENCE} is not here: this script measures the unmodified reference and runs only beside it')
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import synth_refs
    tmp = tempfile.mkdtemp(prefix='brx_refcpu_')
    fasta = os.path.join(tmp, 'ref.fa')
    if a.workload == 'kpn':
        n_bases = synth_refs.write_kpneumoniae_like(fasta)
        ref_desc = 'configs[1] reference (5.5 Mb, 3 circular contigs)'
    else:
        n_bases = synth_refs.write_grch38_like(fasta, scale=SLICE_SCALE)
        ref_desc = f'{n_bases / 1e6:.1f} Mb slice (scale {SLICE_SCALE:.4f}) of the GRCh38-like reference of configs[3]/[4]'
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, 'oracle', 'shim'), REFERENCE]),
               OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    subprocess.check_call(['make', '-s', '-C', os.path.join(REPO, 'oracle'), '_ref/libmyers_ref.so'])
    quantity = str(int(a.mbases_per_worker * 1e6))
    procs = []
    t0 = time.time()
    for i in range(a.workers):
        out = os.path.join(tmp, f'w{i}.json')
        procs.append((out, subprocess.Popen([sys.executable, '-c', WORKER, fasta, quantity, str(42 + i), a.workload, out],
                                            env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    rates, setups = [], []
    chars = 0
    for out, pr in procs:
        _, err = pr.communicate()
        if pr.returncode != 0 or not os.path.isfile(out):
            sys.exit('worker failed:\n' + err.decode()[-2000:])
        rec = json.load(open(out))
        # bases = (FASTQ characters - headers/separators) / 2 is awkward; the loop stops at >= quantity bases, so the
        # reads written hold `quantity` bases plus at most one read: use the sequence lines' share exactly
        rates.append(rec)
        setups.append(rec['setup_seconds'])
        chars += rec['fastq_chars']
    # exact bases: re-derive from the FASTQ size is not needed -- the loop's total is >= quantity and < quantity + one
    # read (<= ~200 kb); quantity is used (a lower bound, biased against the reference by < 5 %)
    bases_each = int(quantity)
    value = sum(bases_each / r['loop_seconds'] for r in rates)
    result = {'value': value, 'unit': 'bases/s', 'cores': a.workers, 'kind': 'reference',
              'per_core': value / a.workers,
              'workload': a.workload, 'reference_genome': ref_desc,
              'sample': f'{a.workers} processes x {bases_each} bases (seed 42+i), unmodified /root/reference '
                        f'(Badread v0.4.2, badread.simulate.simulate) + oracle/shim/edlib, FASTQ to a counting sink; '
                        f'value = sum over processes of bases / seconds inside the read loop (simulate.py:63-86)',
              'setup_seconds_mean': sum(setups) / len(setups), 'loop_seconds_mean': sum(r['loop_seconds'] for r in rates) / len(rates),
              'wall_seconds': time.time() - t0,
              'host': {'cpu': _cpu_model(), 'logical_cpus': os.cpu_count(), 'python': platform.python_version()},
              'measured_by': 'tools/ref_cpu_baseline.py in the CPU container (no GPU); the GPU box has no /root/reference'}
    store = {}
    if os.path.isfile(a.out):
        try:
            store = json.load(open(a.out))
        except ValueError:
            store = {}
    store[a.workload] = result
    with open(a.out, 'w') as f:
        json.dump(store, f, indent=1)
    print(json.dumps(result, indent=1))


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


if __name__ == '__main__':
    main()
