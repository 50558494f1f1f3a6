"""
The traceback-store geometry of the aligner (brx_make_geom / brx_stored / brx_jrep / brx_tb_units in
badread_amd/csrc/brx_align.h are __host__ __device__) checked on the CPU: tests/native/geom_check.hip is compiled
with hipcc and run here.  It proves, over ~1500 random and corner-case geometries and every window setting, that
stored cells never share an address, that addresses stay inside the sized store, and that the windowed store keeps
exactly the superblocks meeting rows c(j) +- H.
"""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_traceback_store_geometry_on_the_host(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    exe = str(tmp_path / 'geom_check')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O2', '-std=c++17', os.path.join(HERE, 'native', 'geom_check.hip'), '-o', exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith('ok'), r.stdout[-2000:] + r.stderr[-2000:]
