"""
Host output stage.  The reference prints FASTQ text and leaves compression to a
`| gzip` pipe -- one core of deflate behind a simulator that emits GB/s.  `GzipSink` compresses every batch of FASTQ
bytes on all usable host cores through libbrx_host.so (csrc/brx_gzip.cpp: independent gzip members of 1 MB, which
`gzip -d` and every gzip reader treat as one stream) before it reaches stdout; `--gzip LEVEL` on the command line.
"""
import ctypes
import os

import numpy as np

from .reference import host_library


def usable_cores():
    """Affinity mask, capped by the cgroup CPU quota when there is one."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return cores


class GzipSink(object):
    def __init__(self, sink, level=6, threads=None, block_bytes=1 << 20):
        self.sink = sink
        self.level = int(level)
        self.threads = int(threads or usable_cores())
        self.block_bytes = int(block_bytes)
        self.lib = host_library()
        self.lib.brx_gzip_bound.restype = ctypes.c_size_t
        self.lib.brx_gzip_bound.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
        self.lib.brx_gzip_parallel.restype = ctypes.c_int
        self.lib.brx_gzip_parallel.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                               ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        self._buf = np.zeros(1 << 20, dtype=np.uint8)
        self.bytes_in = self.bytes_out = 0

    def write(self, part):
        data = np.ascontiguousarray(np.frombuffer(memoryview(part), dtype=np.uint8))
        if data.size == 0:
            return
        need = int(self.lib.brx_gzip_bound(data.size, self.block_bytes))
        if self._buf.size < need:
            self._buf = np.zeros(need, dtype=np.uint8)
        got = ctypes.c_size_t(0)
        rc = self.lib.brx_gzip_parallel(data.ctypes.data, data.size, self.level, self.threads, self.block_bytes,
                                        self._buf.ctypes.data, self._buf.size, ctypes.byref(got))
        if rc != 0:
            raise RuntimeError(f'brx_gzip_parallel failed ({rc})')
        self.sink.write(memoryview(self._buf[:got.value]))
        self.bytes_in += data.size
        self.bytes_out += got.value


# ---------------------------------------------------------------------------------------------
# --gzip-device: the blocks of brx_gzip_device (include/brx.h) cut at the lines of the records
# ---------------------------------------------------------------------------------------------
LONG_READ = 2048          # from this sequence length on, a record is two blocks: header + bases | '+' + qualities
SMALL_BLOCK = 1 << 16     # records of shorter reads are merged into blocks of about this size
MAX_BLOCK = 1 << 26       # longer lines are split (brx_gzip_device takes blocks of up to 128 MB)


def fastq_blocks(rec_off, rec_len, seq_len, n_bytes):
    """Block boundaries (uint64, first 0, last n_bytes) for the first n_bytes of a batch's FASTQ text, from the per-read
    statistics of the reads it holds (record = header line, sequence line, '+' line, quality line: rec_len = header +
    2 seq_len + 5).  A Huffman code per block: bases cost ~2 bits under their own code and qualities ~5, against ~4.5 for
    both under a shared one, so records of long reads are cut between the sequence line and the '+' line; the records of
    short reads (where a block's 150-byte code table would not pay) are merged into mixed blocks."""
    rec_off = np.asarray(rec_off, dtype=np.int64)
    rec_len = np.asarray(rec_len, dtype=np.int64)
    seq_len = np.asarray(seq_len, dtype=np.int64)
    live = (rec_len > 0) & (rec_off + rec_len <= n_bytes)
    rec_off, rec_len, seq_len = rec_off[live], rec_len[live], seq_len[live]
    long_read = seq_len >= LONG_READ
    cuts = [np.array([0, n_bytes], dtype=np.int64)]
    if long_read.any():
        start = rec_off[long_read]
        cuts.append(start)                                                   # the record starts a block ...
        cuts.append(start + rec_len[long_read] - seq_len[long_read] - 3)     # ... '+' starts the next ...
        cuts.append(start + rec_len[long_read])                              # ... and the record ends it
    short = ~long_read
    if short.any():
        # runs of short reads: a cut wherever the bytes since the run's last cut pass SMALL_BLOCK
        ends = rec_off[short] + rec_len[short]
        run_start = np.r_[True, rec_off[short][1:] != ends[:-1]]             # a long read (or a gap) lies before this read
        run_id = np.cumsum(run_start) - 1
        run_base = rec_off[short][run_start][run_id]
        bucket = (ends - run_base - 1) // SMALL_BLOCK
        last_of_bucket = np.r_[(bucket[1:] != bucket[:-1]) | (run_id[1:] != run_id[:-1]), True]
        cuts.append(ends[last_of_bucket])
    cuts = np.unique(np.concatenate(cuts))
    cuts = cuts[(cuts >= 0) & (cuts <= n_bytes)]
    gaps = np.diff(cuts)
    if len(gaps) and int(gaps.max()) > MAX_BLOCK:                             # a line of more than 64 MB: split it evenly
        extra = [np.arange(a + MAX_BLOCK, b, MAX_BLOCK, dtype=np.int64) for a, b in zip(cuts[:-1][gaps > MAX_BLOCK], cuts[1:][gaps > MAX_BLOCK])]
        cuts = np.unique(np.concatenate([cuts] + extra))
    return cuts.astype(np.uint64)
