"""Small checks of the measurement tools that run without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

ASM = """
\t.text
_Z6kernelPj:                            ; @_Z6kernelPj
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tv_mov_b32_e32 v1, 0
.LBB0_1:                                ; =>This Loop Header
\tv_add_u32_e32 v1, 1, v1
\tv_and_b32_e32 v2, 3, v1
.LBB0_2:                                ;   inner loop
\tv_xor_b32_e32 v2, v2, v1
\tds_read_b32 v3, v2
\ts_waitcnt lgkmcnt(0)
\ts_add_i32 s2, s2, -1
\ts_cmp_lg_u32 s2, 0
\ts_cbranch_scc1 .LBB0_2
\tglobal_store_dword v0, v1, s[0:1]
\ts_add_i32 s3, s3, -1
\ts_cmp_lg_u32 s3, 0
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
"""


def test_isa_loops_counts_the_units_of_every_backward_branch(tmp_path):
    path = tmp_path / 'k.s'
    path.write_text(ASM)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'isa_loops.py'), str(path), '_Z6kernelPj', '1'],
                         capture_output=True, text=True, check=True).stdout
    lines = [ln for ln in out.splitlines() if ln.startswith('  .LBB')]
    assert len(lines) == 2
    outer = next(ln for ln in lines if ln.split()[0] == '.LBB0_1')
    inner = next(ln for ln in lines if ln.split()[0] == '.LBB0_2')
    assert 'VALU 3' in outer and 'LDS 1' in outer and 'VMEM 1' in outer and 'stores 1' in outer and 'inner loops 1' in outer
    assert 'VALU 1' in inner and 'LDS 1' in inner and 'SALU 2' in inner and 'inner loops 0' in inner
