#!/usr/bin/env python3
"""Per-kernel resource table of the product library (VGPRs, AGPRs, scratch bytes, LDS, occupancy), from hipcc's
-Rpass-analysis=kernel-resource-usage.  Usage: python tools/kernel_resources.py [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', 'badread_amd', 'csrc')


def table(extra=()):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
           '-Rpass-analysis=kernel-resource-usage', *extra, os.path.join(CSRC, 'brx_hip.hip'), '-o', '/tmp/brx_resources.so']
    err = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark: (?:\s*)([A-Za-z \[\]/]+?): (.+?) \[-Rpass', line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == 'Function Name':
            cur = {'name': subprocess.run(['c++filt', val], capture_output=True, text=True).stdout.strip().split('(')[0]}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    return rows


if __name__ == '__main__':
    rows = table(sys.argv[1:])
    print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
    for r in rows:
        print(f"{r['name'][:44]:44s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>4s}")
