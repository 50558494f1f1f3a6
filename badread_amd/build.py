"""
Build libbrx_hip.so (hipcc, gfx950) and libbrx_host.so (g++, zlib) in-tree (`python -m badread_amd.build`).

Flags that matter:
  --offload-arch=gfx950   the only target (CDNA4 / MI355X); no other architectures, no fallbacks
  -ffp-contract=off       part of the numerical spec: device doubles must round like the oracle's
The .so is git-ignored but travels to the GPU box with the tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.realpath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(CSRC, 'libbrx_hip.so')
SOURCES = ['brx_hip.hip']
DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))) + \
    [os.path.join('..', '..', 'include', 'brx.h'), os.path.join('..', '..', 'include', 'brx_spec.h')]


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def source_hash():
    """First 16 hex digits of the SHA-256 over the kernel sources (csrc/*.h, *.hip, include/*.h): measurements that depend on
    the kernels (profiles/valu_per_base.json) record it, and bench.py flags them as stale when the tree has moved on."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    inc = os.path.join(HERE, '..', 'include')
    files += sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h'))
    for path in files:
        h.update(os.path.basename(path).encode() + b'\0')
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    build_host(force)
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
           '-Wall', '-Wno-unused-function', '-Wno-unused-parameter', '-Wno-unused-variable']
    if verbose:
        cmd.append('-Rpass-analysis=kernel-resource-usage')
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ['-o', OUT]
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


HOST_OUT = os.path.join(CSRC, 'libbrx_host.so')
HOST_DEPS = ['brx_fasta.cpp', 'brx_gzip.cpp', os.path.join('..', '..', 'include', 'brx_host.h'), os.path.join('..', '..', 'include', 'brx.h')]


def build_host(force=False):
    """libbrx_host.so: the CPU-side helpers (FASTA packer, parallel gzip of the output); g++ + zlib, no HIP."""
    if not force and os.path.exists(HOST_OUT) and \
            all(os.path.getmtime(os.path.join(CSRC, d)) <= os.path.getmtime(HOST_OUT) for d in HOST_DEPS):
        return HOST_OUT
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wextra',
           os.path.join(CSRC, 'brx_fasta.cpp'), os.path.join(CSRC, 'brx_gzip.cpp'), '-o', HOST_OUT, '-lz', '-pthread']
    subprocess.check_call(cmd, cwd=CSRC)
    return HOST_OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
    print(HOST_OUT)
