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
    for name, fetch, write, disp in (('void k_mut_post<2>', 1000.0, 500.0, 128), ('void k_mutate_seg<false, 4>', 4000.0, 2000.0, 5),
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
    assert set(k) == {'k_mut_post', 'k_mutate_seg', 'k_win_lane', 'k_fin_align<1,1,1>'}
    assert abs(k['k_mut_post']['hbm_bytes_per_launch'] - (2 * 1000.0 + 500.0) * 1024 / 128) < 1e-6      # per pass: every dispatch
    assert abs(k['k_mutate_seg']['hbm_bytes_per_launch'] - (2 * 4000.0 + 2000.0) * 1024 / 4) < 1e-6          # the priming launch is not a full-size one
    assert abs(k['k_fin_align<1,1,1>']['hbm_bytes_per_launch'] - (2 * 900.0 + 600.0) * 1024 / 2) < 1e-6


TRACE = """Kernel_Name,Start_Timestamp,End_Timestamp,Queue_Id
"k_plan_count(BrxDev, RS*)",1000000,1100000,4
"k_mut_apply(BrxDev)",1200000,3200000,4
"void k_fin_align<16, 8, 65535>(BrxDev)",3300000,9300000,2
"void k_fin_align<1, 1, 1>(BrxDev)",5000000,12000000,4
"void k_fin_quad<1>(BrxDev)",5100000,6100000,1
"k_emit(BrxDev)",12100000,12500000,4
"""


def test_batch_timeline_says_whether_the_widest_class_ends_the_batch(tmp_path):
    """tools/batch_timeline.py (the timeline, not the argument): per kernel first start / last end of the
    LAST batch of a kernel trace, and whether k_fin_align<16,8,...> ends after the bulk set's final alignments."""
    import json
    path = tmp_path / 'kernel_trace.csv'
    warm = TRACE.replace('Kernel_Name,Start_Timestamp,End_Timestamp,Queue_Id\n', '')
    # two batches: the tool reports the second one (everything from the last k_plan_count on)
    second = []
    for ln in warm.strip().splitlines():
        name, rest = ln.rsplit('",', 1)
        a, b, q = rest.split(',')
        second.append(f'{name}",{int(a) + 20000000},{int(b) + 20000000},{q}')
    path.write_text(TRACE + '\n'.join(second) + '\n')
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'batch_timeline.py'), str(path)], capture_output=True, text=True, check=True).stdout
    d = json.loads(out)
    assert abs(d['batch_ms'] - 11.5) < 1e-6
    assert d['kernels']['k_fin_align<16, 8, 65535>'] == {'first_start_ms': 2.3, 'last_end_ms': 8.3, 'busy_ms': 6.0, 'dispatches': 1, 'queues': ['2']}
    assert d['widest_class_ends_ms'] == 8.3 and d['bulk_final_alignments_end_ms'] == 11.0 and d['widest_class_on_critical_path'] is False


def test_parts_log_and_expected_error_rate():
    """simulate._PartsLog: a .parts line when a batch is issued, a .zparts line once the batch's bytes are through the sink (whatever
    the interleaving of consumer and writer thread); simulate.expected_error_rate: the arena's slabs are sized by it."""
    import io
    sys.path.insert(0, REPO)
    from badread_amd import simulate as S
    from badread_amd.identities import Identities
    parts, zparts = io.StringIO(), io.StringIO()
    z = {'out': 0}
    log = S._PartsLog(parts, zparts, lambda: z['out'])
    log.batch(100)                      # issued before anything was written
    z['out'] += 40; log.wrote(60)       # first write of the batch: not complete yet
    assert zparts.getvalue() == ''
    z['out'] += 30; log.wrote(40)       # complete: 70 compressed bytes
    log.batch(0)                        # a batch this rank keeps nothing of
    z['out'] += 5; log.wrote(50); log.batch(50)      # written before the consumer logged it
    assert parts.getvalue() == '100\n0\n50\n' and zparts.getvalue() == '70\n0\n5\n'
    null = io.StringIO()
    assert abs(S.expected_error_rate(Identities(95.0, 2.5, 99.0, null)) - 0.05) < 1e-12
    q30 = S.expected_error_rate(Identities(30.0, 3.0, None, null))
    assert 0.001 < q30 < 0.0014          # E[10^(-q/10)] for q ~ N(30, 3): a little above 10^-3


def test_arena_prefetch_hands_out_what_its_threads_allocated():
    """simulate._ArenaPrefetch (the job's arenas requested when the first engine exists): take() gives each arena once, in order,
    waiting for its thread; a failed allocation is a None the taker replaces; nothing is left after release_rest()."""
    sys.path.insert(0, REPO)
    from badread_amd import simulate as S

    class FakeCuda(object):
        def set_device(self, device):
            pass

    class FakeTorch(object):
        uint8 = 'u8'
        cuda = FakeCuda()

        def __init__(self):
            self.made = 0

        def empty(self, n, dtype=None, device=None):
            self.made += 1
            if self.made == 2:
                raise MemoryError('no room')
            return ('arena', n, device)

    torch = FakeTorch()
    pre = S._ArenaPrefetch(torch, 'dev0', 1000, 3)
    got = [pre.take(), pre.take(), pre.take(), pre.take()]
    assert got[3] is None and sum(1 for g in got[:3] if g == ('arena', 1000, 'dev0')) == 2 and sum(1 for g in got[:3] if g is None) == 1
    assert len(pre.errors) == 1
    pre.release_rest()
    assert pre.take() is None
    # not a GPU engine, or switched off: no prefetch at all

    class Cpu(object):
        device = None
    assert S._ArenaPrefetch.for_job(Cpu(), 10 ** 12, 15000.0, 0.05, 6, 1) is None
