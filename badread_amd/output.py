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
