"""
Host side of `badread simulate` for the MI355X path.

The reference runs one read at a time in a Python loop (/root/reference/badread/simulate.py:63-86).
Here the loop body -- build_fragment (:91-115), Identities.get_identity, sequence_fragment
(:256-358) with get_qscores, and the FASTQ record (:73-82) -- is one C-ABI call per batch of
read indices (brx_simulate_batch, include/brx.h); this module keeps what stays on the host:

  * start-up in the reference's order: load reference, FragmentLengths, adjust_depths (:516-536),
    Identities, models, adapters (:412-430), target size (:124-145), the same banner text;
  * the stop rule: reads are keyed by index, the host prefix-sums the read lengths of each batch
    in index order and cuts after the first read that reaches the target (:63,84), so the output
    is independent of batch size and of the number of GPUs;
  * sharding: rank r of N takes the r-th slice of every super-batch (no collective on the data
    path; the per-read lengths and, for an ordered stdout, the record bytes are gathered);
  * `sequence_fragment(fragment, target_identity, error_model, qscore_model)`: the narrow Python
    boundary the reference's tests use (test/test_simulate.py:47,83), as a batch of one.

Randomness: every draw on the device is Philox4x32-10 keyed by (seed, read index, stream)
(include/brx_spec.h), so a seed fixes the output bytes, but not to the reference's MT19937 stream.
"""
import os
import random
import sys
import time

import numpy as np

from . import settings
from .engine import RS_BAND, RS_NOFRAG, RS_QMISS, RS_TOO_MANY_SEGS, SimParams
from .error_model import ErrorModel
from .fragment_lengths import FragmentLengths
from .identities import Identities
from .misc import float_to_str, get_random_sequence, str_is_int
from .qscore_model import QScoreModel
from .reference import PackedReference
from .version import __version__

NOFRAG_MESSAGE = ('Error: failed to generate any sequence fragments - are your read lengths '
                  'incompatible with your reference contig lengths?')
ADJUST_SAMPLES = 100000
DEFAULT_MAX_BATCH = 65536
DEFAULT_IN_FLIGHT = 6          # super-batches in flight per GPU (--gpu-streams): what bench.py measures


# ---------------------------------------------------------------------------------------------
# start-up pieces (host only)
# ---------------------------------------------------------------------------------------------
def print_intro(output):
    print(f'\nBadread v{__version__}\nlong read simulation', file=output)


def load_reference(reference, output):
    """The reference's function (simulate.py:494-507), same return value: (seqs, depths, circular, hairpin_left,
    hairpin_right) as Python dicts, after printing the summary.  The driver itself uses load_packed_reference."""
    from .misc import load_fasta
    print(f'\nLoading reference from {reference}', file=output)
    seqs, depths, circular, hp_left, hp_right = load_fasta(reference)
    print(f'  {len(seqs):,} contig{"" if len(seqs) == 1 else "s"}:', file=output)
    for name, seq in seqs.items():
        print(f'    {name}: {len(seq):,} bp, {"circular" if circular[name] else "linear"}, {depths[name]:.2f}x depth', file=output)
    if len(seqs) > 1:
        print(f'  total size: {sum(len(v) for v in seqs.values()):,} bp', file=output)
    return seqs, depths, circular, hp_left, hp_right


def load_packed_reference(reference, output):
    """FASTA(.gz) -> PackedReference through the native packer (libbrx_host.so, csrc/brx_fasta.cpp), with the
    reference's summary lines (simulate.py:494-507)."""
    print(f'\nLoading reference from {reference}', file=output)
    pref = PackedReference.from_fasta(reference)
    print(f'  {len(pref.names):,} contig{"" if len(pref.names) == 1 else "s"}:', file=output)
    for name, length in zip(pref.names, pref.lengths):
        shape = 'circular' if pref.circular[name] else 'linear'
        print(f'    {name}: {length:,} bp, {shape}, {pref.depths[name]:.2f}x depth', file=output)
    if len(pref.names) > 1:
        print(f'  total size: {sum(pref.lengths):,} bp', file=output)
    return pref


def adjust_depths(pref, frag_lengths, small_plasmid_bias, rng):
    """
    Depth compensation for fragments that cannot come from a short contig (simulate.py:516-536):
    circular contigs (unless --small_plasmid_bias) are scaled by total / sum(L for L <= len),
    linear contigs by total / sum(min(len, L)), over 100,000 sampled fragment lengths.
    Returns the adjusted {name: depth}; exits like the reference when no length fits a circular contig.
    """
    lengths = np.sort(frag_lengths.sample_many(ADJUST_SAMPLES, rng))
    prefix = np.concatenate(([0], np.cumsum(lengths)))
    total = int(prefix[-1])
    depths = dict(pref.depths)
    for name, ref_len in zip(pref.names, pref.lengths):
        n_fit = int(np.searchsorted(lengths, ref_len, side='right'))     # lengths <= ref_len
        if pref.circular[name]:
            if small_plasmid_bias:
                continue
            passing = int(prefix[n_fit])
            if passing == 0:
                sys.exit('Error: fragment length distribution incompatible with reference lengths '
                         '- try running with --small_plasmid_bias to avoid this error')
        else:
            passing = int(prefix[n_fit]) + ref_len * (len(lengths) - n_fit)
        depths[name] *= total / passing
    return depths


def get_target_size(ref_size, quantity):
    """'250M' / '25x' / plain integer -> bases (simulate.py:124-145)."""
    text = str(quantity)
    if str_is_int(text):
        return int(text)
    scale = {'x': ref_size, 'g': 1000000000, 'm': 1000000, 'k': 1000}.get(text[-1:].lower())
    if scale is not None:
        try:
            return int(round(float(text[:-1]) * scale))
        except ValueError:
            pass
    sys.exit('Error: could not parse quantity\n'
             '--quantity must be either an absolute value (e.g. 250M) or a relative depth '
             '(e.g. 25x)')


def adapter_parameters(param_str):
    fields = param_str.split(',')
    try:
        if len(fields) == 2:
            return [float(f) / 100 for f in fields]
    except ValueError:
        pass
    sys.exit('Error: adapter parameters must be two comma-separated values between 0 and 1')


def build_random_adapters(args):
    """An integer adapter 'sequence' means a random adapter of that length (simulate.py:422-432)."""
    made = []
    for attr in ('start_adapter_seq', 'end_adapter_seq'):
        value = getattr(args, attr)
        if str_is_int(value):
            setattr(args, attr, get_random_sequence(int(value)))
            made.append(True)
        else:
            made.append(False)
    return tuple(made)


def _print_adapter(label, seq, rate, amount, was_random, output):
    if seq and rate > 0.0 and amount > 0.0:
        print(f'{label} adapter:', file=output)
        print(f'  seq: {seq}{" (randomly generated)" if was_random else ""}', file=output)
        print(f'  rate:   {rate * 100.0:.1f}%', file=output)
        print(f'  amount: {amount * 100.0:.1f}%', file=output)
    else:
        print(f'{label} adapter: none', file=output)


def print_adapter_summary(start_rate, start_amount, start_seq, end_rate, end_amount, end_seq,
                          random_start, random_end, output):
    print('', file=output)
    _print_adapter('Start', start_seq, start_rate, start_amount, random_start, output)
    print('', file=output)
    _print_adapter('End', end_seq, end_rate, end_amount, random_end, output)


def print_glitch_summary(glitch_rate, glitch_size, glitch_skip, output):
    print('', file=output)
    if glitch_rate == 0:
        print('Reads will have no glitches', file=output)
        return
    print('Read glitches:', file=output)
    for text, value in (('rate (mean distance between glitches)', glitch_rate),
                        ('size (mean length of random sequence)', glitch_size),
                        ('skip (mean sequence lost per glitch) ', glitch_skip)):
        print(f'  {text} = {float_to_str(value):>5}', file=output)


def print_other_problem_summary(args, output):
    print('\nOther problems:', file=output)
    print(f'  chimera join rate: {args.chimeras}%', file=output)
    print(f'  junk read rate:    {args.junk_reads}%', file=output)
    print(f'  random read rate:  {args.random_reads}%', file=output)


def print_progress(count, bp, target, output):
    percent = min(int(1000.0 * bp / target) / 10, 100.0) if target else 100.0
    print(f'\rSimulating: {count:,} read{" " if count == 1 else "s"}  {bp:,} bp  {percent:.1f}%',
          file=output, flush=True, end='')


def sim_params_from_args(args, frag_lengths, identities, start_rate, start_amount, end_rate, end_amount):
    mode, id_a, id_b, id_max = identities.device_mode()
    return SimParams(frag_mean=frag_lengths.mean, frag_stdev=frag_lengths.stdev,
                     identity_mode=mode, id_a=id_a, id_b=id_b, id_max=id_max,
                     start_rate=start_rate, start_amount=start_amount, end_rate=end_rate, end_amount=end_amount,
                     start_adapter=args.start_adapter_seq, end_adapter=args.end_adapter_seq,
                     junk_rate=args.junk_reads / 100, random_rate=args.random_reads / 100,
                     chimera_rate=args.chimeras / 100,
                     glitch_rate=args.glitch_rate, glitch_size=args.glitch_size, glitch_skip=args.glitch_skip)


# ---------------------------------------------------------------------------------------------
# sharding and the stop rule
# ---------------------------------------------------------------------------------------------
class Shard(object):
    """This process's place among the ranks of one node: rank r of world N (torch.distributed or single)."""

    def __init__(self, rank=0, world=1, dist=None, owns_group=False):
        self.rank, self.world, self.dist = rank, world, dist
        self.owns_group = owns_group          # from_env() created the process group: finish() takes it down again

    def finish(self):
        """Leave the job together.  Every rank leaves run_batches at the same batch (stop rule, bad read, failed sink alike);
        a rank that then tears its connections down while a peer is still flushing makes gloo abort the peer, so the ranks
        meet once more and the group this object created is destroyed in order."""
        if self.world > 1 and self.dist is not None and self.dist.is_initialized():
            self.gather_words(np.zeros(1, dtype=np.uint32), [1] * self.world)
            if self.owns_group:
                self.dist.destroy_process_group()
                self.owns_group = False

    def abandon(self):
        """Leave WITHOUT meeting the peers (this rank failed alone): close the connections so that a peer's next exchange fails
        at once instead of waiting for this rank until the backend's timeout."""
        if self.world > 1 and self.dist is not None and self.owns_group and self.dist.is_initialized():
            try:
                self.dist.destroy_process_group()
            except Exception:              # the group may be in the middle of the collective that failed
                pass
            self.owns_group = False

    @classmethod
    def from_env(cls):
        world = int(os.environ.get('WORLD_SIZE', '1'))
        if world <= 1:
            return cls()
        import torch.distributed as dist
        owns = not dist.is_initialized()
        if owns:
            import torch
            # BRX_DIST_BACKEND=gloo keeps the exchange on the host (CPU tensors) whatever the engine computes on: several ranks
            # can then share ONE GPU (BRX_DEVICE picks it), which is how the tests run the HIP engine under world > 1 on a 1-GPU box
            backend = os.environ.get('BRX_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            if torch.cuda.is_available():
                from .engine import rank_device_index
                torch.cuda.set_device(rank_device_index(torch.cuda.device_count()))
            dist.init_process_group(backend=backend)
        return cls(dist.get_rank(), dist.get_world_size(), dist, owns_group=owns)

    def slice_of(self, first, count):
        """Contiguous slice of the super-batch [first, first+count) owned by this rank."""
        per = -(-count // self.world)
        lo = min(self.rank * per, count)
        hi = min(lo + per, count)
        return first + lo, hi - lo

    def _device(self):
        import torch
        return torch.device('cuda', torch.cuda.current_device()) if self.dist.get_backend() == 'nccl' else torch.device('cpu')

    def gather_words(self, words, counts):
        """Every rank's uint32 array (rank i contributes counts[i] words, known to all ranks from the batch plan), in
        rank order, on every rank: the ONLY collective of the driver -- 4 bytes per read (its length and two status
        bits) plus one word per rank, which is what every rank needs to apply the stop rule identically."""
        if self.world == 1:
            return [words]
        import torch
        dev = self._device()
        width = max(max(counts), 1)
        mine = torch.zeros(width, dtype=torch.int32, device=dev)
        if len(words):
            mine[:len(words)] = torch.from_numpy(np.ascontiguousarray(words).view(np.int32)).to(dev)
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return [p[:n].cpu().numpy().view(np.uint32) for p, n in zip(parts, counts)]

    def collect_bytes(self, mine, sizes, staging):
        """Record bytes to rank 0, point to point: rank r > 0 sends its `sizes[r]` bytes (a device tensor over RCCL /
        xGMI in production, a CPU tensor over gloo in the tests) and rank 0 receives them in rank order into `staging`.
        No rank other than 0 ever holds another rank's records.  Yields (rank, uint8 tensor) on rank 0."""
        if self.world == 1 or self.rank == 0:
            if sizes[0]:
                yield 0, mine[:sizes[0]]
            for r in range(1, self.world):
                if sizes[r]:
                    buf = staging(sizes[r])
                    self.dist.recv(buf, src=r)
                    yield r, buf
        elif sizes[self.rank]:
            part = mine[:sizes[self.rank]].contiguous()
            if self.dist.get_backend() != 'nccl' and part.device.type != 'cpu':
                part = part.cpu()                # host exchange (gloo) of an engine that computes on the GPU
            self.dist.send(part, dst=0)


FLAG_NOFRAG, FLAG_BAD = 1 << 31, 1 << 30          # status bits packed beside the read length in the 4 B/read gather
BAD_STATUS = RS_TOO_MANY_SEGS | RS_BAND | RS_QMISS


def cut_point(seq_lens, running_total, target_size):
    """
    Index (within this batch) of the read at which the reference's loop stops, or None.  Empty
    reads (seq_len 0) are skipped and never stop the loop (simulate.py:70-71).
    """
    seq_lens = np.asarray(seq_lens, dtype=np.int64)
    reached = np.flatnonzero((running_total + np.cumsum(seq_lens) >= target_size) & (seq_lens > 0))
    return int(reached[0]) if len(reached) else None


def plan_batch(remaining_bases, mean_length, world, max_batch):
    """Reads in the next super-batch: a little over the expected need, rounded to 64 per rank."""
    expect = remaining_bases / max(mean_length, 1.0)
    want = int(expect * 1.05) + 8 * world
    per_rank = min(max(-(-want // world), 1), max_batch)
    per_rank = -(-per_rank // 64) * 64 if per_rank >= 64 else per_rank
    return per_rank * world


class _HostRing(object):
    """Pinned host buffers for the FASTQ bytes on their way out: the device-to-host copy of
    a finished batch lands in page-locked memory (full PCIe rate, no staging copy by the runtime) and a writer thread
    hands it to the sink while the GPU works on the next batches.  A buffer returns to the ring when the writer is done
    with it; `depth` buffers bound the memory and apply back-pressure when the sink is slower than the GPU."""

    def __init__(self, torch, pinned, depth=3):
        import queue
        import threading
        self.torch, self.pinned = torch, pinned
        self.free = queue.Queue()
        for _ in range(depth):
            self.free.put(None)                     # allocated (and grown) on first use
        self.todo = queue.Queue()
        self.error = None
        self.defer_errors = False                   # multi-rank runs: a failed sink is reported through the per-batch exchange, so that every rank stops at the same batch
        self.sink = None
        self.sink_seconds = self.alloc_seconds = 0.0
        self.copy_stream = None                     # device -> pinned copies of the batches' bytes (write)
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def prealloc(self, nbytes):
        """Page-lock the ring's buffers NOW, on a thread of their own, while the first device batch computes (pinning 3 x 2.5 GB
        is ~1 s that the consumer thread spent at its first writes: profiles/r04_cli_30x.json, ring_alloc 0.95 s)."""
        import threading
        if not self.pinned or nbytes <= 0:
            return
        slots = []
        while True:                                  # take the empty slots; whatever is in use already stays as it is
            try:
                slots.append(self.free.get_nowait())
            except Exception:
                break

        def make():
            for slot in slots:
                if slot is None:
                    try:
                        slot = self.torch.empty(int(nbytes), dtype=self.torch.uint8, pin_memory=True)
                    except Exception:                # not fatal: stage() allocates what it needs
                        slot = None
                self.free.put(slot)
        threading.Thread(target=make, daemon=True).start()

    def _run(self):
        while True:
            item = self.todo.get()
            if item is None:
                return
            buf, n, landed, _source = item
            try:
                if landed is not None:              # the copy into `buf` runs on the ring's own stream: sleep until it has landed
                    while not landed.query():       # (event.synchronize() spins a core on this stack)
                        time.sleep(0.0002)
                if self.error is None and n:
                    t0 = time.perf_counter()
                    self.sink(memoryview(buf.numpy())[:n])
                    self.sink_seconds += time.perf_counter() - t0
            except BaseException as ex:             # surfaced by the next stage() / write() of the producer, or by flush()
                self.error = ex
            self.free.put(buf)

    def stage(self, nbytes):
        """A host buffer of at least nbytes (blocks while all buffers are with the writer).  A sink that has failed
        (BrokenPipe behind `| head`, a full disk) stops the run here, at the next batch, not after the whole target."""
        if self.error is not None and not self.defer_errors:
            raise self.error
        buf = self.free.get()
        if buf is None or buf.numel() < nbytes:
            cap = max(int(nbytes * 1.25), 1 << 20)
            t0 = time.perf_counter()
            buf = self.torch.empty(cap, dtype=self.torch.uint8, pin_memory=self.pinned)
            self.alloc_seconds += time.perf_counter() - t0
        return buf

    def write(self, tensor):
        """Copy a uint8 tensor (device or host) into a ring buffer and queue it for the sink.  A device tensor is copied
        ASYNCHRONOUSLY on the ring's own stream: the consumer thread goes on to the next batch (stop rule, refill of the pipeline)
        while the bytes cross PCIe, and the writer thread waits for the copy, not the consumer -- on configs[4] the consumer spent
        5.0 of the read loop's 9.1 s inside blocking copies (profiles/r04_cli_30x_hifi.json).  The source tensor travels with
        the queue entry so that its memory is not reused before the copy has read it."""
        n = int(tensor.numel())
        buf = self.stage(n)
        landed = None
        if n and self.pinned and tensor.is_cuda and not os.environ.get('BRX_SYNC_COPY_OUT'):
            torch = self.torch
            if self.copy_stream is None:
                # (a high-priority stream for the copy was measured: BRX_COPY_PRIORITY=-1 made configs[4]'s read loop 9.8 s against 8.8 s at the
                #  default priority, profiles/r05h; the copy itself runs at 57 GB/s on an idle GPU: tools/d2h_probe.py)
                prio = int(os.environ.get('BRX_COPY_PRIORITY', '0'))
                self.copy_stream = torch.cuda.Stream(device=tensor.device, priority=prio)
            self.copy_stream.wait_stream(torch.cuda.current_stream(tensor.device))     # the tensor's producer has been waited for on the current stream
            with torch.cuda.stream(self.copy_stream):
                buf[:n].copy_(tensor, non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(self.copy_stream)
        elif n:
            buf[:n].copy_(tensor, non_blocking=False)
        self.todo.put((buf, n, landed, tensor if landed is not None else None))

    def flush(self, reraise=True):
        self.todo.put(None)
        self.thread.join()
        if reraise and self.error is not None:
            raise self.error


class _ArenaPrefetch(object):
    """The job's scratch arenas, allocated by threads of their own from the moment the first engine exists -- beside the model
    tables, the genome's upload and the first batch -- instead of when the read loop asks for them.  On a freshly leased GPU a
    40 GB arena is ~4 s of `hipMalloc` (the driver provisions and clears the memory: 21 s of thread time for the five clones of a
    30x human job, profiles/r05_cli_30x.json); started 1.4 s before the loop, the engines are there 1.4 s earlier.  take() hands
    out the next arena (waiting for its thread) or None."""

    def __init__(self, torch, device, nbytes, count):
        import threading
        self.nbytes, self.count = int(nbytes), int(count)
        self.slots = [None] * self.count
        self.errors = []
        self.lock = threading.Lock()
        self.next = 0
        self.seconds = 0.0

        def alloc(i):
            t0 = time.perf_counter()
            try:
                torch.cuda.set_device(device)
                self.slots[i] = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
            except BaseException as ex:                  # the taker allocates what it needs itself (and reports what fails then)
                self.errors.append(ex)
            with self.lock:
                self.seconds += time.perf_counter() - t0
        self.threads = [threading.Thread(target=alloc, args=(i,), daemon=True) for i in range(self.count)]
        for th in self.threads:
            th.start()

    def take(self):
        with self.lock:
            i = self.next
            if i >= self.count:
                return None
            self.next += 1
        self.threads[i].join()
        t, self.slots[i] = self.slots[i], None
        return t

    def release_rest(self):
        while self.take() is not None:
            pass

    @staticmethod
    def for_job(engine, target_size, mean_length, error_rate, in_flight, world):
        """Arenas for the batches in flight of THIS job on THIS device, or None (not a GPU engine, a job of one small batch,
        BRX_ARENA_PREFETCH=0).  How many: what the job will use and the free memory holds (_BatchPool.engines_that_fit's rule)."""
        # Measured (profiles/r05i_arena_prefetch.json, r05k_*): the read loop then runs undisturbed -- 18.8-19.0 s for the 30x human job
        # (20.9-23.7 without), 5.8-6.1 s = 15-16 Gbases/s for the configs[4] flavour (9.2 s).  What the allocations cost depends on what
        # the GPU did before: right behind another process that gave 40+ GB back they are ~0.13 s per GB (the driver is still clearing),
        # hold a lock the genome's upload waits for (0.2 -> 6.2 s) and the whole command takes what it took without this (27.2 s against
        # 26.1-27.0 s); on a GPU that has been idle for 20 s they are nearly free and the command is 22.0 s against 23.4 s, 8.7 s
        # against 11.6 s for configs[4].  Never slower, so on by default; BRX_ARENA_PREFETCH=0 is the round-4 behaviour.
        if getattr(getattr(engine, 'device', None), 'type', '') != 'cuda' or os.environ.get('BRX_ARENA_PREFETCH', '1') in ('', '0'):
            return None
        from .engine import arena_estimate
        torch = engine.torch
        first_batch = plan_batch(target_size, mean_length, world, DEFAULT_MAX_BATCH) // max(world, 1)
        if first_batch < 4096:
            return None                                  # a small job: one arena, sized by presize
        nbytes = engine.arena_bytes(first_batch, mean_length, error_rate) if hasattr(engine, 'arena_bytes') else arena_estimate(first_batch, mean_length, error_rate)
        out_bytes = int(first_batch * (2.1 * mean_length + 400.0))
        batches = -(-int(target_size) // max(int(first_batch * mean_length * max(world, 1)), 1))
        n = max(1, min(int(in_flight), batches))
        free, _ = torch.cuda.mem_get_info(engine.device)
        reserve = int(float(os.environ.get('BRX_DRIVER_RESERVE_GB', '24')) * (1 << 30))
        while n > 1 and n * nbytes + (2 * n + 2) * out_bytes + reserve > free:
            n -= 1
        return _ArenaPrefetch(torch, engine.device, nbytes, n)


class _BatchPool(object):
    """`in_flight` engines (the given one + clones sharing its device tables), each driven by its own host thread on
    its own HIP stream, so that several batches overlap on the GPU.

    Batches are CONSUMED in index order (the stop rule), but they do not FINISH in index order: read lengths differ, and so do
    batch times.  Round 2 released an engine only when its batch was consumed, so an engine that had finished batch k + 3
    idled until batches k .. k + 2 were through -- every round of six batches took as long as its slowest member, and a 30x job
    through this driver ran at 60-70 % of the rate bench.py measures on the same kernels.  Now the worker thread itself takes
    a free engine when it starts a batch, copies the batch's FASTQ bytes device-to-device out of the engine's buffer when the
    batch is done (2 GB at HBM speed: ~1 ms) and gives the engine back at once; `depth` = in_flight + 2 batches may be outstanding,
    the surplus holding only their bytes."""

    def __init__(self, engine, in_flight, arenas=None, device_gzip=False):
        import concurrent.futures
        import queue
        self.arenas = arenas
        # --gzip-device: the worker of a batch that the stop rule cannot cut (submit(..., pack=True): the job still needs several
        # batches' worth of bases behind it) packs ITS batch on ITS engine's stream as soon as the batch is done and hands over
        # the gzip members instead of the text; the consumer packs only the job's last batches itself.  Round 5 packed every batch
        # on the consumer thread: 9.3 s for the 185 GB of text of the configs[4] job, which bound that flavour (10.4 s of read
        # loop against 6.0 s).
        self.device_gzip = bool(device_gzip)
        self.engines = [engine]
        self.streams = [None]
        torch = getattr(engine, 'torch', None)           # absent on the tests' CPU checker engines
        self.on_gpu = torch is not None and getattr(engine, 'device', None) is not None and engine.device.type == 'cuda'
        if in_flight > 1 and hasattr(engine, 'clone'):
            self.streams = [torch.cuda.Stream(device=engine.device) if self.on_gpu else None for _ in range(in_flight)]
            self.engines += [None] * (in_flight - 1)
        elif self.on_gpu:
            self.streams = [torch.cuda.Stream(device=engine.device)]
        self.free = queue.Queue()
        self.free.put(0)                                 # engine 0 exists already: it takes the first batch
        self.create_seconds = 0.0
        self.create_errors = []                          # clones that could not be made: the job runs on fewer engines
        import threading
        self.lock = threading.Lock()                     # the counters below are updated by several worker threads
        # Mapping a clone's 40 GB arena takes ~0.6-1.2 s per engine whoever does it and whenever (the driver clears the memory:
        # 14-29 ms per GB measured, also for ONE 200 GB allocation on an idle device -- 5.9 s), so the clones are made by their own
        # threads while engine 0 already computes: a job's first seconds run on fewer engines instead of on none.  Clone i is
        # started when the job submits its (i + 1)-th batch: a one-batch job (a bacterial genome at 100x) maps no clone at all.
        def make(i):                                     # a clone maps tens of GB of scratch: its own thread, beside the first batches
            t0 = time.perf_counter()
            try:
                if self.on_gpu:
                    torch.cuda.set_device(self.engines[0].device)
                spare = self.arenas.take() if self.arenas is not None else None       # allocated since the first engine exists, or None
                self.engines[i] = self.engines[0].clone(scratch_tensor=spare) if spare is not None else self.engines[0].clone()
            except BaseException as ex:                  # this slot never joins the queue; engine 0 and the other clones carry on
                with self.lock:
                    self.create_errors.append(ex)
                    self.create_seconds += time.perf_counter() - t0
                return
            with self.lock:
                self.create_seconds += time.perf_counter() - t0
            self.free.put(i)
        self.makers = [threading.Thread(target=make, args=(i,), daemon=True) for i in range(1, len(self.engines))]
        self.started = 0                                 # makers started so far
        self.submitted = 0
        self.depth = len(self.engines) + (2 if len(self.engines) > 1 else 0)
        self.job_seconds = self.job_count = 0.0          # wall time of the device batches (bench.py --d2h reports the average)
        self.wait_engine_seconds = 0.0
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=len(self.engines)) if len(self.engines) > 1 else None

    def __len__(self):
        return self.depth

    @staticmethod
    def engines_that_fit(torch, engine, in_flight, out_bytes):
        """How many engines the device's FREE memory holds: every clone maps an arena as large as the first engine's, every engine
        owns an output buffer, and in_flight + 2 copies of a batch's bytes may wait for the consumer.  A reserve stays free for what
        the runtime allocates on its own: the private segments of the planning kernels (1.6 KB per lane, per hardware queue: ~0.9 GB
        each over the ~20 queues of a six-engine job), the gzip stage, other ranks' records on rank 0.  A job that asks for more
        engines than fit runs on fewer (a 96-batch job of six 65536-read engines with 40 GB arenas died with
        HSA_STATUS_ERROR_OUT_OF_RESOURCES at the edge of the 288 GB, where no allocation of ours failed)."""
        torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info(engine.device)
        reserve = int(float(os.environ.get('BRX_DRIVER_RESERVE_GB', '24')) * (1 << 30))
        scratch = engine.scratch_bytes() if hasattr(engine, 'scratch_bytes') else 0
        n = in_flight
        while n > 1 and (n - 1) * scratch + (2 * n + 2) * int(out_bytes) + reserve > free:
            n -= 1
        return n

    COPY_STEP = 1 << 28

    @classmethod
    def _copy_of(cls, torch, out):
        """The batch's bytes out of the engine's buffer, in a block whose CAPACITY is a multiple of 256 MB.  A plain clone() asks
        torch's caching allocator for the batch's exact size -- 2.05-2.10 GB, a different one every time -- and a cached block that is
        a few MB short serves nobody: it stays cached (per stream) and a new one is mapped.  Over the 94 batches of the 30x job
        the cache crept through the 24 GB this driver leaves to the runtime, and a queue that needed its private segment then
        died with HSA_STATUS_ERROR_OUT_OF_RESOURCES at 88 % of the job (`Available Free mem : 0 MB`; no allocation of ours had
        failed).  With the capacity in steps every freed block fits the next batch."""
        n = int(out.numel())
        if n <= cls.COPY_STEP // 16:
            return out.clone()
        cap = -(-n // cls.COPY_STEP) * cls.COPY_STEP
        buf = torch.empty(cap, dtype=torch.uint8, device=out.device)
        buf[:n].copy_(out)
        return buf[:n]

    def submit(self, seed, first, n_mine, pack=False):
        self.submitted += 1
        while self.started < min(self.submitted - 1, len(self.makers)):      # the k-th batch in the pipeline is what clone k - 1 is for
            self.makers[self.started].start()
            self.started += 1

        def job():
            import torch
            t_q = time.perf_counter()
            i = self.free.get()
            t_job = time.perf_counter()
            with self.lock:
                self.wait_engine_seconds += t_job - t_q
            try:
                stream = self.streams[i]
                eng = self.engines[i]
                if n_mine == 0:
                    return torch.zeros(0, dtype=torch.uint8), np.zeros(0, dtype=eng.stats_dtype)
                if not self.on_gpu:
                    out, stats = eng.simulate_batch(seed, first, n_mine, allow_nofrag=True)
                    return torch.from_numpy(np.ascontiguousarray(out).copy()), stats.copy()
                torch.cuda.set_device(eng.device)
                with torch.cuda.stream(stream):
                    out, stats = eng.simulate_batch_device(seed, first, n_mine, allow_nofrag=True)
                    stats = stats.copy()
                    packed = None
                    if pack and self.device_gzip and len(stats) and hasattr(eng, 'gzip_device'):
                        from .output import fastq_blocks
                        live = np.flatnonzero(stats['rec_len'] > 0)
                        if len(live) and not (stats['status'] & (RS_NOFRAG | BAD_STATUS)).any():
                            nbytes = int(stats['rec_off'][live[-1]] + stats['rec_len'][live[-1]])
                            blocks = fastq_blocks(stats['rec_off'], stats['rec_len'], stats['seq_len'], nbytes)
                            packed = (nbytes, eng.gzip_device(out[:nbytes], blocks))      # a new tensor, made on this batch's stream
                    out = self._copy_of(torch, out) if packed is None else None   # the engine's buffer is free again; the copy runs on this batch's stream ...
                    stream.synchronize()                     # ... and is complete before the engine is handed to the next batch
                    return (out, stats) if not self.device_gzip else (out, stats, packed)
            finally:
                with self.lock:
                    self.job_seconds += time.perf_counter() - t_job
                    self.job_count += 1
                self.free.put(i)
        if self.pool is None:
            class _Done(object):
                def __init__(self, v): self.v = v
                def result(self): return self.v
            return _Done(job())
        return self.pool.submit(job)

    def close(self):
        for th in self.makers[:self.started]:            # clones no batch asked for were never started
            th.join()
        if self.pool is not None:
            self.pool.shutdown(wait=True)
        for eng in self.engines[1:]:
            if eng is not None:
                eng.close()


def run_batches(engine, seed, target_size, mean_length, write, output, shard=None, max_batch=None, in_flight=1, device_gzip=False,
                local_write=None, local_parts=None, expected_error=None, arenas=None):
    """
    The `while total_size < target_size` loop (simulate.py:63-86) over super-batches of read indices.
    `write(bytes_like)` receives the FASTQ bytes in read order on rank 0 only.  Returns (read count, total bases).

    Up to `in_flight` super-batches run at once (one engine clone, stream and host thread each).  They cover
    consecutive index ranges and are CONSUMED in index order, where the stop rule is applied, so the output does
    not depend on `in_flight` or on the batch sizes; batches issued beyond the stopping read are discarded.  The
    issue schedule depends only on consumed totals, so every rank of a multi-GPU run plans the same batches.

    Between ranks: one all_gather of 4 bytes per read (length + two status bits) and one word per rank (bytes kept)
    per super-batch, then the kept record bytes travel point to point to rank 0 only.  On every rank the bytes leave the
    GPU through a ring of pinned host buffers drained by a writer thread (_HostRing).

    device_gzip: every rank turns the bytes it keeps into gzip members ON ITS GPU (brx_gzip_device, blocks cut at the
    lines of the records: badread_amd.output.fastq_blocks) before they go anywhere; `write` then receives gzip data.

    local_write (--output-shards): EVERY rank hands the bytes it keeps to its own `local_write` through its own pinned ring
    (its own PCIe link, its own file) and nothing travels to rank 0; `local_parts(n)` receives the byte count of every
    super-batch, which is what puts the ranks' files back into read order (batch by batch, rank after rank).  The
    exchange of 4 bytes per read and the stop rule are the same: the same reads are kept.

    A sink that fails on one rank of a multi-rank run (a full disk, a closed pipe) is reported in the same exchange -- one
    word per rank -- so that every rank leaves the loop at the same batch instead of waiting in a collective.
    """
    import collections
    import torch
    shard = shard or Shard()
    max_batch = max_batch or DEFAULT_MAX_BATCH
    count = total = 0
    next_read = 0
    expected_mean = float(mean_length)
    if shard.rank == 0:
        print_progress(count, total, target_size, output)
    timing = run_batches.last_timing = collections.Counter()     # seconds of the consumer thread per activity (bench.py --d2h)
    t0 = time.perf_counter()
    t_job = t0
    first_arena = None
    if hasattr(engine, 'presize'):               # the arena of the first engine (the clones copy its size) for the batches this job will issue
        first_batch = plan_batch(target_size, expected_mean, shard.world, max_batch) // shard.world
        first_arena = arenas.take() if arenas is not None else None
        if first_arena is not None:              # allocated beside the start-up (simulate._ArenaPrefetch)
            engine.adopt_scratch(first_arena)
        elif expected_error is None:
            engine.presize(first_batch, expected_mean)
        else:                                    # the job's identity law: arenas for Q30 reads are half those of 95 % reads
            engine.presize(first_batch, expected_mean, expected_error)
        out_bytes = int(first_batch * (engine.expected_record_bytes() if hasattr(engine, 'expected_record_bytes') else 2.1 * expected_mean + 400.0))
    else:
        out_bytes = 0
    asked = fit = max(1, int(in_flight))
    if arenas is not None and first_arena is not None:
        fit = min(fit, arenas.count)             # decided when the arenas were requested, by the same rule, from the memory that was free then
    elif fit > 1 and hasattr(engine, 'scratch_bytes') and getattr(getattr(engine, 'device', None), 'type', '') == 'cuda':
        fit = _BatchPool.engines_that_fit(engine.torch, engine, fit, out_bytes)
    if shard.world > 1:                          # the issue schedule depends on the depth of the pipeline: the same on every rank
        fit = min(int(x[0]) for x in shard.gather_words(np.array([fit], dtype=np.uint32), [1] * shard.world))
    if fit < asked and shard.rank == 0:
        print(f'  {fit} of the {asked} batches in flight asked for fit into the free device memory', file=output)
    pool = _BatchPool(engine, fit, arenas, device_gzip=device_gzip)
    timing['create_engines'] = time.perf_counter() - t0
    ring = None
    if local_write is not None or shard.rank == 0:
        ring = _HostRing(torch, pinned=pool.on_gpu)
        ring.sink = local_write if local_write is not None else write
        ring.defer_errors = shard.world > 1
        # (Pinning the ring's buffers on a thread of their own beside the first batch looked free and was not: 3 x 2.6 GB of
        #  hipHostMalloc ran against the clone threads' 40 GB arenas and every clone took 1.1 s instead of 0.05 -- 5.6 s of
        #  wait_for_engine, profiles/r05b_cli_30x.json.  BRX_RING_PREALLOC=1 keeps the experiment reachable.)
        if pool.on_gpu and out_bytes and os.environ.get('BRX_RING_PREALLOC') and target_size > 3 * max_batch * expected_mean:
            ring.prealloc(int(1.25 * out_bytes))
    sink_failed_on = None

    def staging(nbytes):
        """Where rank 0 receives another rank's records (device memory under RCCL): a tensor of ITS OWN per receive.  The ring
        copies device tensors to the host asynchronously on its copy stream and keeps the tensor with the queue entry until the
        copy has landed; one reused staging buffer would be overwritten by the next rank's `recv` (ordered behind the current
        stream only) while that copy still reads it.  The caching allocator makes this as cheap as the reuse was."""
        return torch.empty(int(nbytes), dtype=torch.uint8, device=shard._device())

    gz_engine = None
    if device_gzip:
        if not hasattr(engine, 'gzip_device'):
            sys.exit('Error: --gzip-device needs the GPU engine')
        from .output import fastq_blocks
        gz_engine = engine.clone(1 << 20) if hasattr(engine, 'clone') else engine      # its own context: the others are busy on their threads
    pending = collections.deque()          # (None, future, first_of_super_batch, n_super, first, n_mine)
    fatal = bad_read = None

    def fill():
        """Keep the pipeline full: what is outstanding is assumed to deliver its expected number of bases.  Called at
        points that depend only on consumed totals, so every rank issues the same batches."""
        nonlocal next_read
        while len(pending) < len(pool):
            outstanding = sum(p[3] for p in pending) * expected_mean
            remaining = target_size - total - outstanding
            if remaining <= 0 and pending:
                break
            n_super = plan_batch(max(remaining, 1), expected_mean, shard.world, max_batch)
            first, n_mine = shard.slice_of(next_read, n_super)
            # --gzip-device: a batch with at least three batches' worth of bases still to come behind it is kept whole
            fut = pool.submit(seed, first, n_mine, pack=device_gzip and remaining > 4.0 * n_super * expected_mean)
            pending.append((None, fut, next_read, n_super, first, n_mine))
            next_read += n_super

    try:
        while total < target_size:
            fill()
            _, fut, base, n_super, first, n_mine = pending.popleft()
            t0 = time.perf_counter()
            res = fut.result()
            out, stats, prepacked = res if len(res) == 3 else (res[0], res[1], None)
            timing['wait_for_batch'] += time.perf_counter() - t0
            timing['batches'] += 1
            # ---- the 4 B/read exchange: length (0 for skipped reads) | NOFRAG << 31 | BAD << 30 ----
            words = (stats['seq_len'].astype(np.uint32) * (stats['rec_len'] > 0)).astype(np.uint32)
            assert not len(words) or int(words.max()) < FLAG_BAD
            words |= np.where(stats['status'] & RS_NOFRAG, FLAG_NOFRAG, 0).astype(np.uint32)
            words |= np.where(stats['status'] & BAD_STATUS, FLAG_BAD, 0).astype(np.uint32)
            per_rank = [Shard(r, shard.world).slice_of(base, n_super)[1] for r in range(shard.world)]
            if shard.world > 1:                     # one more word per rank: "my sink has failed"
                mine = np.append(words, np.uint32(1 if ring is not None and ring.error is not None else 0))
                parts = shard.gather_words(mine, [c + 1 for c in per_rank])
                failed = [r for r, part in enumerate(parts) if part[-1]]
                if failed:
                    sink_failed_on = failed[0]
                    break
                allw = np.concatenate([part[:-1] for part in parts])
            else:
                allw = words
            lens = (allw & (FLAG_BAD - 1)).astype(np.int64)
            cut = cut_point(lens, total, target_size)
            wrong = np.flatnonzero(allw & (FLAG_NOFRAG | FLAG_BAD))
            stop_at = None
            if len(wrong) and (cut is None or wrong[0] < cut):
                stop_at = int(wrong[0])
                if allw[stop_at] & FLAG_NOFRAG:
                    fatal = True
                else:
                    bad_read = base + stop_at
            last = stop_at - 1 if stop_at is not None else (cut if cut is not None else n_super - 1)
            # bytes of my reads with global batch position <= last
            my_lo = first - base
            keep = int(np.clip(last - my_lo + 1, 0, n_mine))
            my_bytes = int(stats['rec_off'][keep - 1] + stats['rec_len'][keep - 1]) if keep else 0
            packed = False
            if prepacked is not None and my_bytes == prepacked[0]:          # the whole batch is kept: its worker has packed it already
                out = prepacked[1]
                my_bytes = int(out.numel())
                packed = True
                timing['batches_packed_by_their_worker'] += 1
            elif prepacked is not None:
                # the stop rule cut a batch that was packed ahead (reads far longer than the job expected): unpack it on the host, once
                import gzip as _gzip
                text = _gzip.decompress(bytes(prepacked[1].cpu().numpy()))
                out = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).to(prepacked[1].device)
            if gz_engine is not None and my_bytes and not packed:
                t0 = time.perf_counter()
                blocks = fastq_blocks(stats['rec_off'][:keep], stats['rec_len'][:keep], stats['seq_len'][:keep], my_bytes)
                out = gz_engine.gzip_device(out[:my_bytes], blocks)       # a new tensor: the engine's buffer is free again
                my_bytes = int(out.numel())
                packed = True
                timing['device_gzip'] += time.perf_counter() - t0
            sizes = [my_bytes]
            if shard.world > 1 and local_write is None:
                sizes = [int(x[0]) for x in shard.gather_words(np.array([my_bytes], dtype=np.uint32), [1] * shard.world)]
                assert my_bytes < 2 ** 32
            if pool.on_gpu and my_bytes and not packed:
                out = out[:my_bytes]                # `out` is this batch's own copy (made by its worker): nothing to wait for
            used = lens[:last + 1]
            count += int((used > 0).sum())
            total += int(used.sum())
            stop = fatal or bad_read is not None
            if count and not stop:
                expected_mean = max(total / count, 1.0)
            if not stop and total < target_size:
                fill()                              # the freed engine starts its next batch while this one's bytes leave
            t0 = time.perf_counter()
            if local_write is not None:             # every rank: its own bytes through its own ring to its own file
                if my_bytes:
                    ring.write(out[:my_bytes])
                if local_parts is not None:
                    local_parts(my_bytes)
            else:
                for _, part in shard.collect_bytes(out, sizes, staging):
                    ring.write(part)                # rank 0: through pinned memory to the writer thread
            timing['copy_out'] += time.perf_counter() - t0
            if shard.rank == 0:
                print_progress(count, total, target_size, output)
            if stop:
                break
    finally:
        for _, fut, *_ in pending:          # speculative batches past the stopping read
            try:
                fut.result()
            except Exception:
                pass
        if getattr(getattr(engine, 'device', None), 'type', '') == 'cuda':       # what the device looked like when the last batch was through (BRX_DRIVER_TIMING)
            try:
                free_b, _total_b = torch.cuda.mem_get_info(engine.device)
                timing['device_free_gb_at_end'] = free_b / float(1 << 30)
                timing['torch_reserved_gb_at_end'] = torch.cuda.memory_reserved(engine.device) / float(1 << 30)
                timing['torch_allocated_gb_at_end'] = torch.cuda.memory_allocated(engine.device) / float(1 << 30)
            except Exception:
                pass
        t0 = time.perf_counter()
        pool.close()
        if arenas is not None:                  # arenas no clone asked for (a job that stopped early, ranks that agreed on fewer engines)
            arenas.release_rest()
            timing['arena_prefetch_thread_seconds'] = arenas.seconds
            if arenas.errors:                   # a prefetch that failed left its clone to allocate by itself: say so (it may be what ran out of memory later)
                timing['arena_prefetch_failures'] = len(arenas.errors)
                if shard.rank == 0:
                    print(f'  {len(arenas.errors)} of the {arenas.count} scratch arenas could not be allocated ahead '
                          f'({type(arenas.errors[0]).__name__}: {arenas.errors[0]})', file=output)
        if gz_engine is not None and gz_engine is not engine:
            gz_engine.close()
        timing['close_engines'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if ring is not None:
            # a sink error must not mask the exception that is already propagating; in a multi-rank run it is reported below
            ring.flush(reraise=sys.exc_info()[0] is None and sink_failed_on is None and not ring.defer_errors)
            timing['sink'] = ring.sink_seconds
            timing['ring_alloc'] = ring.alloc_seconds
        timing['flush'] = time.perf_counter() - t0
        timing['retries'] = sum(getattr(e, 'retries', 0) for e in pool.engines if e is not None)
        timing['device_batch_seconds_avg'] = pool.job_seconds / max(pool.job_count, 1.0)
        timing['wait_for_engine'] = pool.wait_engine_seconds
        timing['engines'] = len(pool.engines)
        timing['run_batches_seconds'] = time.perf_counter() - t_job
        timing['create_clones_thread_seconds'] = pool.create_seconds
        if pool.create_errors and shard.rank == 0:
            print(f'\n  {len(pool.create_errors)} of the {len(pool.engines) - 1} extra batch engines could not be created '
                  f'({pool.create_errors[0]}): the job ran on fewer', file=output)
    if shard.rank == 0:
        print('\n', file=output)
    if ring is not None and ring.error is not None and (sink_failed_on is not None or ring.defer_errors):
        raise ring.error                     # this rank's own sink
    if sink_failed_on is not None:
        sys.exit(f'Error: the output of rank {sink_failed_on} failed; every rank stopped at the same batch')
    if fatal:
        sys.exit(NOFRAG_MESSAGE)
    if bad_read is not None:
        sys.exit(f'Error: read {bad_read} exceeded an internal limit of the GPU path (status bits in its statistics); '
                 'no output was written for it or any later read')
    return count, total


def expected_error_rate(identities):
    """Mean errors per base of the job's identity law (beta: 1 - mean; qscore-normal: E[10^(-q / 10)]): what the traceback stores of
    a batch grow with (engine.presize)."""
    import math
    if identities.type == 'beta':
        return min(max(1.0 - float(identities.mean), 0.0), 1.0)
    sigma = float(identities.stdev) * math.log(10.0) / 10.0
    return min(10.0 ** (-float(identities.mean) / 10.0) * math.exp(0.5 * sigma * sigma), 1.0)


class _PartsLog(object):
    """PREFIX.<rank>.parts (and .zparts): a line per batch once the batch's bytes have gone through the sink.  batch() is called by the
    consumer thread in batch order, wrote() by the ring's writer thread after every write of the sink; a batch's bytes are written
    by writes of its own (the ring is handed one batch at a time), so when the bytes written reach a batch's end the sink's output
    counter stands at the end of that batch's last gzip member."""

    def __init__(self, parts_file, zparts_file=None, sink_bytes_out=None):
        import collections
        import threading
        self.parts_file, self.zparts_file, self.sink_bytes_out = parts_file, zparts_file, sink_bytes_out
        self.lock = threading.Lock()
        self.ends = collections.deque()
        self.issued = self.written = self.z_before = 0

    def batch(self, n):
        with self.lock:
            self.parts_file.write(f'{n}\n')
            self.issued += int(n)
            self.ends.append(self.issued)
            self._settle()

    def wrote(self, n):
        with self.lock:
            self.written += int(n)
            self._settle()

    def _settle(self):
        while self.ends and self.written >= self.ends[0]:
            self.ends.popleft()
            if self.zparts_file is not None:
                z = int(self.sink_bytes_out())
                self.zparts_file.write(f'{z - self.z_before}\n')
                self.z_before = z


# ---------------------------------------------------------------------------------------------
# the driver
# ---------------------------------------------------------------------------------------------
def simulate(args, output=sys.stderr, engine=None, stdout=None, shard=None):
    """Same contract as badread.simulate.simulate(args, output): FASTQ to stdout, the rest to `output`."""
    marks = [('enter_simulate', time.time())]           # wall-clock marks of the start-up (BRX_DRIVER_TIMING: tools/cli_30x.sh)

    def mark(name):
        marks.append((name, time.time()))
    shard = shard or Shard.from_env()
    quiet = _Null() if shard.rank != 0 else output
    print_intro(quiet)
    seed = args.seed if args.seed is not None else int.from_bytes(os.urandom(7), 'little')
    if shard.world > 1 and args.seed is None:
        seed = _broadcast_seed(shard, seed)
    random.seed(seed)
    host_rng = np.random.RandomState(seed % (2 ** 32))
    pref = load_packed_reference(args.reference, quiet)
    mark('reference_loaded')
    frag_lengths = FragmentLengths(args.mean_frag_length, args.frag_length_stdev, quiet)
    depths = adjust_depths(pref, frag_lengths, args.small_plasmid_bias, host_rng)
    identities = Identities(args.mean_identity, args.identity_stdev, args.max_identity, quiet)
    if engine is None:
        from .engine import default_engine
        engine = default_engine()
    mark('engine_created')
    # the genome goes to the device BEFORE the arenas are asked for (round 6): the allocations of 200+ GB hold the runtime's lock the
    # upload would otherwise wait for -- 0.2 s of copy became 2-6 s behind them right after another process (profiles/r05i, r05k)
    _, cum_weight = pref.contig_weights(depths)
    engine.set_reference(pref, cum_weight)
    mark('reference_on_device')
    # the job's arenas from now on, beside everything below (models, tables, the first batch): _ArenaPrefetch
    arenas = _ArenaPrefetch.for_job(engine, get_target_size(pref.n_bases, args.quantity), float(args.mean_frag_length),
                                    expected_error_rate(identities), getattr(args, 'gpu_streams', None) or DEFAULT_IN_FLIGHT, shard.world)
    # a model file that is not in the cache is aligned (align_kmers, error_model.py:179-229) on THIS engine
    error_model = ErrorModel(args.error_model, quiet, aligner=lambda qs, ts: engine.align_batch(qs, ts)[0])
    qscore_model = QScoreModel(args.qscore_model, quiet)
    mark('models_loaded')
    print_glitch_summary(args.glitch_rate, args.glitch_size, args.glitch_skip, quiet)
    start_rate, start_amount = adapter_parameters(args.start_adapter)
    end_rate, end_amount = adapter_parameters(args.end_adapter)
    random_start, random_end = build_random_adapters(args)
    print_adapter_summary(start_rate, start_amount, args.start_adapter_seq, end_rate, end_amount,
                          args.end_adapter_seq, random_start, random_end, quiet)
    print_other_problem_summary(args, quiet)
    target_size = get_target_size(pref.n_bases, args.quantity)
    print(f'\nTarget read set size: {target_size:,} bp\n', file=quiet)

    engine.set_error_model(error_model.tables())
    engine.set_qscore_model(qscore_model.tables())
    engine.set_params(sim_params_from_args(args, frag_lengths, identities, start_rate, start_amount,
                                           end_rate, end_amount))
    mark('tables_on_device')
    sink = stdout if stdout is not None else getattr(sys.stdout, 'buffer', None)
    gzip_level = getattr(args, 'gzip_level', None)
    device_gzip = bool(getattr(args, 'gzip_device', False))
    if device_gzip and gzip_level is not None:
        sys.exit('Error: --gzip and --gzip-device exclude each other')
    if gzip_level is not None and sink is not None and shard.rank == 0:
        from .output import GzipSink
        sink = GzipSink(sink, gzip_level)          # multi-threaded gzip members (libbrx_host.so)
    if sink is not None:
        def write(part):
            sink.write(memoryview(part))
    else:                                   # a text-only stdout (e.g. captured in tests)
        def write(part):
            sys.stdout.write(bytes(part).decode('latin-1'))
    # --output-shards PREFIX: every rank writes PREFIX.<rank>.fastq[.gz] itself (and the bytes per batch to PREFIX.<rank>.parts),
    # so that N ranks leave through N PCIe links and N files instead of rank 0's one of each (simulate.py:77-82 is one print loop)
    #   PREFIX.<rank>.parts: one line per batch = the bytes this rank handed to its sink for that batch -- FASTQ text for the plain
    #   file and for --gzip (whose file holds that text as gzip members), compressed bytes for --gzip-device (the members are made
    #   on the GPU, batch by batch).  With --gzip, PREFIX.<rank>.zparts holds the compressed bytes per batch beside it (every batch
    #   ends a gzip member), so that the .gz files too can be put back in read order without decompressing them.
    local_write = local_parts = None
    files = []
    prefix = getattr(args, 'output_shards', None)
    if prefix:
        zipped = device_gzip or gzip_level is not None
        open_error = None
        try:                                # a rank that cannot open its files must not leave the others in their first exchange
            shard_file = open(f'{prefix}.{shard.rank}.fastq' + ('.gz' if zipped else ''), 'wb')
            files.append(shard_file)
            parts_file = open(f'{prefix}.{shard.rank}.parts', 'w')
            files.append(parts_file)
            zparts_file = None
            if gzip_level is not None:
                zparts_file = open(f'{prefix}.{shard.rank}.zparts', 'w')
                files.append(zparts_file)
        except OSError as ex:
            open_error = ex
        failed = [int(x[0]) for x in shard.gather_words(np.array([1 if open_error else 0], dtype=np.uint32), [1] * shard.world)]
        if any(failed):
            for f in files:
                f.close()
            shard.finish()
            sys.exit(f'Error: could not open the output shards of rank(s) {[r for r, x in enumerate(failed) if x]}'
                     + (f': {open_error}' if open_error else ''))
        shard_sink = shard_file
        if gzip_level is not None:
            from .output import GzipSink
            shard_sink = GzipSink(shard_file, gzip_level)
        log = _PartsLog(parts_file, zparts_file, (lambda: shard_sink.bytes_out) if zparts_file is not None else None)

        def local_write(part):
            shard_sink.write(memoryview(part))
            log.wrote(len(part))

        local_parts = log.batch
    try:
        try:
            result = run_batches(engine, seed, target_size, float(args.mean_frag_length), write, quiet, shard,
                                 in_flight=getattr(args, 'gpu_streams', None) or DEFAULT_IN_FLIGHT, device_gzip=device_gzip,
                                 local_write=local_write, local_parts=local_parts, expected_error=expected_error_rate(identities), arenas=arenas)
            if prefix and hasattr(shard_sink, 'flush') and shard_sink is not shard_file:
                shard_sink.flush()
        finally:
            for f in files:
                f.close()
    except (SystemExit, OSError):
        shard.finish()                      # exits every rank takes at the same batch: NOFRAG, a bad read, a failed sink
        raise
    except BaseException:
        # Not an exit the ranks take together (an engine error, a sink raising something else: ADVICE r4): the peers are in, or on
        # their way to, an exchange this rank will never join.  Leave without meeting them -- under torchrun the failing rank takes
        # the job down; a peer left alone fails in its exchange instead of waiting for the backend's timeout.
        shard.abandon()
        raise
    shard.finish()
    if sink is not None and hasattr(sink, 'flush'):
        sink.flush()
    if os.environ.get('BRX_DRIVER_TIMING') and shard.rank == 0:      # seconds of the consumer thread per activity + rate, for tools/cli_30x.sh
        t = dict(run_batches.last_timing)
        t['bases'], t['reads'] = result[1], result[0]
        print('driver_timing ' + repr({k: round(float(v), 3) for k, v in t.items()}), file=sys.stderr)
        mark('done')
        t0 = float(os.environ.get('BRX_T0', marks[0][1]))      # epoch seconds at which the shell started the command, if it says so
        steps = {'interpreter_and_imports': marks[0][1] - t0}
        steps.update({name: at - before for (name, at), (_, before) in zip(marks[1:], marks[:-1])})
        print('startup_timing ' + repr({k: round(float(v), 3) for k, v in steps.items()}), file=sys.stderr)
    return result


class _Null(object):
    def write(self, *_):
        return 0

    def flush(self):
        pass


def _broadcast_seed(shard, seed):
    import torch
    dev = torch.device('cuda', torch.cuda.current_device()) if shard.dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([seed], dtype=torch.int64, device=dev)
    shard.dist.broadcast(t, src=0)
    return int(t.item())


# ---------------------------------------------------------------------------------------------
# sequence_fragment: the narrow Python boundary (simulate.py:256-358)
# ---------------------------------------------------------------------------------------------
_seq_engine_state = {}


def _fragment_alphabet(fragment):
    """ACGT -> 0..3, N -> 4, other symbols of this fragment -> 5.. (all treated as 'not in the model')."""
    sym = bytearray(b'ACGTN' + b'N' * 11)
    lut = np.full(256, 4, dtype=np.uint8)
    for code, ch in enumerate(b'ACGT'):
        lut[ch] = code
    raw = np.frombuffer(fragment.encode('latin-1'), dtype=np.uint8)
    nxt = 5
    for b in np.unique(raw):
        if chr(b) not in 'ACGTN':
            if nxt < 16:
                lut[b], sym[nxt] = nxt, b
                nxt += 1
    return lut[raw], np.frombuffer(bytes(sym), dtype=np.uint8)


def sequence_fragment(fragment, target_identity, error_model, qscore_model, engine=None, seed=None, read_index=0):
    """
    Drop-in for badread.simulate.sequence_fragment: returns (seq, quals, actual_identity,
    identity_by_qscores).  Runs brx_sequence_fragments on a batch of one; the random streams are
    keyed by `seed` (default: 64 bits from Python's `random`, so random.seed() still fixes the result).
    """
    if engine is None:
        from .engine import default_engine
        engine = default_engine()
    # (re)configure the engine unless it holds exactly these tables already: the engine remembers what it was last
    # given, so a simulate() or a direct set_* call in between is noticed
    held = getattr(engine, '_configured', {})
    if held.get('em') is not error_model.tables():
        engine.set_error_model(error_model.tables())
    if held.get('qm') is not qscore_model.tables():
        engine.set_qscore_model(qscore_model.tables())
    _seq_engine_state['keep'] = (error_model, qscore_model)
    if seed is None:
        seed = random.getrandbits(64)
    if len(fragment) == 0:
        return '', '', 0.0, 0.0
    codes, sym = _fragment_alphabet(fragment)
    res, stats = engine.sequence_fragments(seed, read_index, [codes], [float(target_identity)])
    seq_codes, quals = res[0]
    st = stats[0]
    seq = sym[seq_codes].tobytes().decode('latin-1')
    n_cols, padded_len = int(st['n_cols']), int(st['padded_len'])
    actual_identity = int(st['n_match']) / n_cols if n_cols else 0.0
    # identity_by_qscores averages over the PADDED read, like get_qscores (qscore_model.py:71-73)
    identity_by_qscores = 1.0 - float(st['qerr_sum']) / padded_len if padded_len else 0.0
    return seq, quals.tobytes().decode('latin-1'), actual_identity, identity_by_qscores


assert settings.ALIGNMENT_INTERVAL == 25 and settings.ALIGNMENT_SIZE == 1000     # hard-coded in csrc/brx_kernels.h
