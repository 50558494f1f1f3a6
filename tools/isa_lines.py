#!/usr/bin/env python3
"""Static attribution of a kernel's instructions to source lines: hipcc -S --cuda-device-only -gline-tables-only, then
    python tools/isa_lines.py brx_g.s '_Z12k_mutate_segILb0ELb0ELi4E' [first_inst last_inst]
prints VALU / VMEM / LDS / SALU counts per (file, line) of the innermost .loc, largest first.  (Inlined callees are charged to
their own lines; a static count says nothing about trip counts -- pair it with tools/isa_loops.py.)"""
import collections
import re
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from isa_loops import kernel_lines, unit  # noqa: E402


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
    files = {}
    for line in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).rsplit('/', 1)[-1]
    cur, n = ('?', 0), 0
    per = collections.defaultdict(collections.Counter)
    for ln in kernel_lines(path, prefix):
        s = ln.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((';', '.', '//')):
            continue
        op = s.split()[0]
        if not re.match(r'^[a-z_0-9]+$', op):
            continue
        if lo <= n <= hi:
            per[cur][unit(op)] += 1
        n += 1
    rows = sorted(per.items(), key=lambda kv: -kv[1]['VALU'])
    tot = collections.Counter()
    for _, c in rows:
        tot.update(c)
    print('total', dict(tot))
    for (f, l), c in rows[:int(sys.argv[5]) if len(sys.argv) > 5 else 45]:
        print(f'{f}:{l:<6d} VALU {c["VALU"]:4d}  VMEM {c["VMEM"]:3d}  LDS {c["LDS"]:3d}  SALU {c["SALU"]:4d}')


if __name__ == '__main__':
    main()
