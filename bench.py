#!/usr/bin/env python3
"""
bench.py -- simulated bases per second of the HIP hot path on N MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], the configuration the metric is quoted on that fits one GPU):
a synthetic "K. pneumoniae-like" reference -- one circular 5.3 Mb chromosome plus two circular
plasmids (200 kb depth=2, 5 kb depth=10), uniform ACGT from numpy default_rng(1) -- default
`badread simulate` parameters (length 15000,13000; identity 95,99,2.5; nanopore2023 error and
qscore models; default adapters, junk/random/chimera 1 %, glitches 10000,25,25), seed 42.
A "step" is ONE pass of the whole hot path (plan -> fragments -> mutate -> align -> qscores ->
FASTQ bytes) over one batch of `--reads-per-step` read indices per GPU (default 131072 reads x 15 kb
~ 2 Gbases, seven times the 50x job).  The batch goes through the C-ABI as `--streams` device
batches (brx_simulate_batch, 16384 reads each by default) that are in flight together, one context
+ HIP stream + host thread each; device batches of consecutive steps follow each other without a
barrier, exactly as the CLI driver runs them (badread_amd.simulate.run_batches).  Inputs (packed
reference, model tables) are resident in HBM before the timed region; the FASTQ bytes stay in HBM
(the PCIe-inclusive rate is reported separately as `value_incl_d2h`).  Weak scaling: every rank
processes its own slices of the read-index space, no collective on the data path.

The JSON line carries `roofline` (dominant kernel, HBM bound, algorithmic bytes = 2.26 B per
simulated base) and `cpu_baseline` (the C oracle -- a single-threaded port of
the same algorithm -- run on all host cores of this box on a bounded sample of the same workload).
"""
import argparse
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '40')      # a hardware queue per stream of every in-flight batch: main + side stream each (HIP's default of 4 serialises them)
import json
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

ALGO_BYTES_PER_BASE = 2.26          # 0.25 B packed reference read + 2 B FASTQ written + header share
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
SEED = 42


def kpneumoniae_like():
    """configs[1] reference: names, sequences, depths, circular flags."""
    import collections
    rng = np.random.default_rng(1)
    seqs = collections.OrderedDict()
    depths, circular = {}, {}
    for name, length, depth in (('chromosome', 5300000, 1.0), ('plasmid_1', 200000, 2.0), ('plasmid_2', 5000, 10.0)):
        seqs[name] = np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, length)].tobytes().decode()
        depths[name], circular[name] = depth, True
    return seqs, depths, circular


def build_workload(io_null):
    from badread_amd.engine import SimParams
    from badread_amd.error_model import ErrorModel
    from badread_amd.fragment_lengths import FragmentLengths
    from badread_amd.identities import Identities
    from badread_amd.qscore_model import QScoreModel
    from badread_amd.reference import PackedReference
    from badread_amd.simulate import adjust_depths
    seqs, depths, circular = kpneumoniae_like()
    pref = PackedReference.from_seqs(seqs, depths, circular)
    frag = FragmentLengths(15000, 13000, io_null)
    ident = Identities(95, 2.5, 99, io_null)
    depths = adjust_depths(pref, frag, False, np.random.RandomState(SEED))
    _, cum = pref.contig_weights(depths)
    mode, a, b, mx = ident.device_mode()
    params = SimParams(frag_mean=15000, frag_stdev=13000, identity_mode=mode, id_a=a, id_b=b, id_max=mx)
    em = ErrorModel('nanopore2023', io_null).tables()
    qm = QScoreModel('nanopore2023', io_null).tables()
    return pref, cum, em, qm, params


def configure(engine, wl):
    pref, cum, em, qm, params = wl
    engine.set_reference(pref, cum)
    engine.set_error_model(em)
    engine.set_qscore_model(qm)
    engine.set_params(params)
    return engine


def cpu_worker(first_read, budget_s, out_path):
    """One process = one core: the oracle over 16-read chunks of the same read-index stream for `budget_s` seconds."""
    import io
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import pyoracle
    eng = configure(pyoracle.OracleEngine(), build_workload(io.StringIO()))
    eng.simulate_batch(SEED, first_read, 2)                      # page everything in before the clock starts
    bases = reads = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        _, st = eng.simulate_batch(SEED, first_read + reads, 16)
        bases += int(st['seq_len'].sum())
        reads += 16
    with open(out_path, 'w') as f:
        json.dump({'bases': bases, 'reads': reads, 'seconds': time.perf_counter() - t0}, f)


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes
    expose 256 hardware threads but limit the container to 16 CPUs' worth of time: /sys/fs/cgroup/cpu.max)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(first_read, budget_s):
    """The oracle (oracle/brx_oracle.c, a scalar C port of the same path: gamma/beta draws, fragment build, mutate
    loop with block-Myers window alignments, final alignment + traceback, qscore lookup, FASTQ record) on EVERY host
    usable core (usable_cores()): one single-threaded process per core, disjoint slices of the same read-index stream, own clock each."""
    import subprocess
    import tempfile
    cores = usable_cores()
    tmp = tempfile.mkdtemp(prefix='brx_cpu_')
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='')
    procs = []
    for i in range(cores):
        out = os.path.join(tmp, f'{i}.json')
        procs.append((out, subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(first_read + i * 100000),
                                             str(budget_s), out], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    bases = reads = 0
    rate = 0.0
    done = 0
    for out, pr in procs:
        pr.wait()
        if os.path.isfile(out):
            rec = json.load(open(out))
            bases += rec['bases']; reads += rec['reads']; rate += rec['bases'] / rec['seconds']; done += 1
    return {'value': rate, 'unit': 'bases/s', 'cores': done, 'kind': 'port',
            'sample': f'{reads} reads / {bases} bases of the same workload and seed, {budget_s:.0f} s of CPU time per core, '
                      f'oracle/brx_oracle.c (gcc -O2) as one single-threaded process per host core; value = sum of the '
                      f'per-process rates'}


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == '--cpu-worker':
        cpu_worker(int(sys.argv[2]), float(sys.argv[3]), sys.argv[4])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reads-per-step', type=int, default=131072,
                    help='read indices per GPU per step; split into --streams device batches')
    ap.add_argument('--scratch-gb', type=float, default=30.0, help='scratch arena per in-flight batch')
    ap.add_argument('--streams', type=int, default=8,
                    help='device batches in flight per GPU (one context + HIP stream + host thread each): the slowest '
                         'read of one device batch overlaps the bulk of the others')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='seconds each host core runs the cpu_baseline leg (0 = skip)')
    ap.add_argument('--d2h', action='store_true', help='also time steps that copy the FASTQ bytes to pinned host memory')
    args = ap.parse_args()

    import io
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a ROCm device: the HIP path has no CPU fallback')
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local))
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from badread_amd.engine import HipEngine
    wl = build_workload(io.StringIO())
    C = max(1, args.streams)
    R = max(64, args.reads_per_step // C)              # reads per device batch (one brx_simulate_batch call)
    engines = [configure(HipEngine(local, scratch_bytes=int(args.scratch_gb * (1 << 30))), wl) for _ in range(C)]
    streams = [torch.cuda.Stream(device=local) for _ in range(C)]
    eng = engines[0]
    for e, st_ in zip(engines, streams):              # prime every context (lazy module load, buffers) -- not a step
        with torch.cuda.stream(st_):
            e.simulate_batch_device(SEED, 2 ** 40, 64, expected_bytes=R * 36000)
    torch.cuda.synchronize()

    def step(index, e=None):
        first = (index * world + rank) * R
        out, stats = (e or eng).simulate_batch_device(SEED, first, R, expected_bytes=R * 36000)
        return out, stats

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(step_indices):
        """The device batches of steps `step_indices`, C in flight: worker i owns context i / stream i and takes
        every C-th device batch; batch b of step k covers read indices ((k*C + b)*world + rank)*R ..."""
        indices = [k * C + b for k in step_indices for b in range(C)]
        acc = [{'bases': 0, 'g1_bases': 0, 'passes': 0, 'stages': {}, 'final_launches': 0, 'misses': 0, 'error': None} for _ in range(C)]

        def worker(i):
            try:
                torch.cuda.set_device(local)
                with torch.cuda.stream(streams[i]):
                    for idx in indices[i::C]:
                        _, stats = step(idx, engines[i])
                        acc[i]['bases'] += int(stats['seq_len'].sum())
                        g_words = engines[i].read_cycles(R)[:, 7]
                        acc[i]['g1_bases'] += int(stats['seq_len'][g_words == 1].sum())
                        acc[i]['passes'] += engines[i].mutate_passes()
                        for name, ms in engines[i].stage_ms().items():
                            acc[i]['stages'][name] = acc[i]['stages'].get(name, 0.0) + ms
                        acc[i]['final_launches'] += engines[i].final_launches()
                        acc[i]['misses'] += engines[i].window_misses()
                    streams[i].synchronize()
            except BaseException as ex:          # surfaced on the main thread
                acc[i]['error'] = ex

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(C)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        for a in acc:
            if a['error'] is not None:
                raise a['error']
        return acc

    run_steps(list(range(args.warmup)) if args.warmup else [])
    sync()
    t0 = time.perf_counter()
    acc = run_steps([args.warmup + k for k in range(args.steps)])
    sync()
    elapsed = time.perf_counter() - t0
    bases = sum(a['bases'] for a in acc)
    final_launches = sum(a['final_launches'] for a in acc)
    stage_sum = {}
    for a in acc:
        for name, ms in a['stages'].items():
            stage_sum[name] = stage_sum.get(name, 0.0) + ms

    t = torch.tensor([elapsed, float(bases)], dtype=torch.float64, device='cuda')
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, bases = float(tmax[0].item()), float(t[1].item())
    value = bases / elapsed

    d2h = None
    if args.d2h and rank == 0:
        host = torch.empty(R * 40000, dtype=torch.uint8).pin_memory()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        b2 = 0
        for k in range(args.steps):
            out, stats = step(args.warmup + k)
            host[:out.numel()].copy_(out, non_blocking=False)
            b2 += int(stats['seq_len'].sum())
        torch.cuda.synchronize()
        d2h = b2 / (time.perf_counter() - t1)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    stages = {k: v / (args.steps * C) for k, v in stage_sum.items()}          # per device batch
    # dominant single kernel: k_fin_align<1,1,1> (final banded Myers alignment + traceback of the reads whose band
    # fits one 32-bit word per lane); stage 'align1' is the HIP-event duration of ONE launch, averaged over launches
    kernel = 'k_fin_align<1,1,1>'
    launches = final_launches / args.steps
    g1_bases_per_step = sum(a['g1_bases'] for a in acc) / args.steps
    launch_ms = stages['align1']
    algo_bytes = ALGO_BYTES_PER_BASE * g1_bases_per_step / launches
    achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
    bases_per_step_rank0 = bases / (args.steps * world)
    traffic = None
    tfile = os.path.join(REPO, 'profiles', 'pmc_traffic.json')
    if os.path.isfile(tfile):
        try:
            rec = json.load(open(tfile))
            if rec.get('reads_per_step') == R and rec.get('kernel') == kernel:
                traffic = rec.get('hbm_bytes_per_launch')
        except (OSError, ValueError):
            pass
    result = {
        'metric': 'simulated bases/sec', 'value': value, 'unit': 'bases/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1000.0 * elapsed / args.steps, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
        'config': {'workload': 'configs[1]: 5.5 Mb K. pneumoniae-like synthetic reference (3 circular contigs), '
                               'nanopore2023 error+qscore models, default badread simulate parameters, seed 42',
                   'reads_per_step_per_gpu': R * C, 'bases_per_step_per_gpu': bases_per_step_rank0,
                   'device_batches_per_step': C, 'reads_per_device_batch': R,
                   'parallelism': f'reads sharded by index over {world} GPU(s), reference replicated, no collectives'},
        'roofline': {'bound': 'hbm', 'kernel': kernel, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'algorithmic_bytes_per_launch': algo_bytes, 'bases_per_launch': g1_bases_per_step / launches,
                     'launch_ms': launch_ms, 'launches_per_step': launches,
                     'note': 'integer-ALU / latency bound path: see DESIGN.md section 5; one launch per device batch; '
                             'launch_ms is the HIP-event duration of one launch while other batches share the GPU'},
        'stage_ms_per_device_batch': stages, 'mutate_passes_per_device_batch': sum(a['passes'] for a in acc) / (args.steps * C),
        'traceback_window_misses_per_step': sum(a['misses'] for a in acc) / args.steps,
    }
    if d2h is not None:
        result['value_incl_d2h'] = d2h
    if world == 1 and args.cpu_seconds > 0:
        result['cpu_baseline'] = cpu_baseline(10_000_000, args.cpu_seconds)
        result['gpu_over_cpu'] = value / result['cpu_baseline']['value']
    else:
        result['cpu_baseline'] = None
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
