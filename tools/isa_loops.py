"""
Static view of a kernel's loops in the gfx950 assembly hipcc emits (-S --cuda-device-only): for every backward branch, the
instructions between its target label and the branch, split by unit (VALU / SALU / VMEM / LDS / other).  Used to see what a
column of an aligner or an iteration of the mutate loop costs before any GPU time is spent.
    python tools/isa_loops.py brx.s '_Z11k_fin_alignILi1ELi1ELi1E' [min_instructions]
"""
import re
import sys


def kernel_lines(path, prefix):
    out, on = [], False
    for line in open(path):
        if not on:
            if line.startswith(prefix) and line.rstrip().split(':')[0].startswith(prefix) and ':' in line:
                on = True
            continue
        if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
            break
        out.append(line.rstrip('\n'))
    return out


def unit(op):
    if op.startswith('v_'):
        return 'VALU'
    if op.startswith(('s_waitcnt', 's_nop', 's_sleep', 's_barrier')):
        return 'wait'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_'):
        return 'SALU'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'VMEM'
    if op.startswith('ds_'):
        return 'LDS'
    return 'other'


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = kernel_lines(path, prefix)
    labels, insts = {}, []
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((';', '.', '//')) and not re.match(r'^\.LBB\d+_\d+:', s):
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        op = s.split()[0]
        if re.match(r'^[a-z_0-9]+$', op):
            insts.append((op, s))
    loops = []
    for i, (op, s) in enumerate(insts):
        if op.startswith(('s_cbranch', 's_branch')):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    print(f'{prefix}: {len(insts)} instructions, {len(loops)} backward branches')
    for a, b, tgt in sorted(loops, key=lambda t: t[0]):
        body = insts[a:b + 1]
        if len(body) < floor:
            continue
        cnt = {}
        for op, _ in body:
            cnt[unit(op)] = cnt.get(unit(op), 0) + 1
        inner = sum(1 for x, y, _ in loops if a < x and y < b)
        stores = sum(1 for op, _ in body if op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store')))
        print(f'  {tgt:12s} [{a:6d}..{b:6d}] {len(body):5d} insts  ' + '  '.join(f'{k} {v}' for k, v in sorted(cnt.items())) +
              f'  stores {stores}  inner loops {inner}')


if __name__ == '__main__':
    main()
