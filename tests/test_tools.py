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


ASM_LINES = """
\t.file\t1 "/x/csrc" "brx_a.h"
\t.file\t2 "/x/csrc" "brx_b.h"
_Z6kernelPj:                            ; @_Z6kernelPj
\t.loc\t1 10 3
\tv_mov_b32_e32 v1, 0
\tv_add_u32_e32 v1, 1, v1
\t.loc\t2 20 1
\tglobal_load_dword v2, v0, s[0:1]
\tv_xor_b32_e32 v2, v2, v1
\t.loc\t1 10 9
\tv_and_b32_e32 v2, 3, v1
\ts_endpgm
.Lfunc_end0:
"""


def test_isa_lines_charges_instructions_to_their_source_lines(tmp_path):
    path = tmp_path / 'k.s'
    path.write_text(ASM_LINES)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'isa_lines.py'), str(path), '_Z6kernelPj'],
                         capture_output=True, text=True, check=True).stdout
    rows = {ln.split()[0]: ln for ln in out.splitlines() if ln.startswith('brx_')}
    assert 'VALU    3' in rows['brx_a.h:10'] and 'VMEM   0' in rows['brx_a.h:10']
    assert 'VALU    1' in rows['brx_b.h:20'] and 'VMEM   1' in rows['brx_b.h:20']


def test_pmc_traffic_all_carries_every_kernel_that_may_rank_first(tmp_path):
    """bench.py's roofline.traffic is looked up by the kernel the RUN ranks first; two kernels are within a few per cent of each
    other this round, so profiles/pmc_traffic.json holds all of them (tools/pmc_traffic.py --all)."""
    import json
    rows = ['kernel,counter,sum,dispatches']
    for name, fetch, write, disp in (('void k_mutate_seg<false, false, 4>', 1000.0, 500.0, 128), ('void k_mutate_seg<true, false, 4>', 4000.0, 2000.0, 5),
                                     ('k_win_lane', 300.0, 200.0, 128), ('void k_fin_align<1, 1, 1>', 900.0, 600.0, 3), ('k_build', 1.0, 1.0, 3)):
        rows.append(f'"{name}",FETCH_SIZE,{fetch},{disp}')
        rows.append(f'"{name}",WRITE_SIZE,{write},{disp}')
    path = tmp_path / 'pmc.csv'
    path.write_text('\n'.join(rows) + '\n')
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'pmc_traffic.py'), str(path), '65536', '--all', 'human'],
                         capture_output=True, text=True, check=True).stdout
    rec = json.loads(out)
    assert rec['reads_per_step'] == 65536 and rec['workload'] == 'human' and len(rec['csrc_sha16']) == 16
    k = rec['kernels']
    assert set(k) == {'k_mutate_seg<false>', 'k_mutate_seg<true>', 'k_win_lane', 'k_fin_align<1,1,1>'}
    assert abs(k['k_mutate_seg<false>']['hbm_bytes_per_launch'] - (2 * 1000.0 + 500.0) * 1024 / 128) < 1e-6      # per pass: every dispatch
    assert abs(k['k_mutate_seg<true>']['hbm_bytes_per_launch'] - (2 * 4000.0 + 2000.0) * 1024 / 4) < 1e-6          # the priming launch is not a full-size one
    assert abs(k['k_fin_align<1,1,1>']['hbm_bytes_per_launch'] - (2 * 900.0 + 600.0) * 1024 / 2) < 1e-6
