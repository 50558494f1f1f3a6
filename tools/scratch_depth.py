"""
Where does a kernel's scratch (register spill) traffic sit?  Reads the gfx950 assembly of libbrx_hip.so's source
(hipcc -S --cuda-device-only) and lists, per kernel, its scratch_load / scratch_store instructions by the loop depth of the basic
block they are in (LLVM's `Depth=` comments): a reload per READ (depth 1 in the per-read loop of a kernel) costs nothing, one per
column or per proposal round does.
    python tools/scratch_depth.py [kernel-substring ...]      -> table on stdout, JSON with --json
"""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly():
    out = os.path.join(tempfile.mkdtemp(prefix='brx_isa_', dir=os.path.join(REPO, 'gpurun_out') if os.path.isdir(os.path.join(REPO, 'gpurun_out')) else None), 'brx.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only',
                           '-Wno-unused-function', '-Wno-unused-parameter', '-Wno-unused-variable', '-Wno-unused-command-line-argument',
                           os.path.join(REPO, 'badread_amd', 'csrc', 'brx_hip.hip'), '-o', out])
    return out


def scan(path, wanted):
    res, kern, depth = {}, None, 0
    head = re.compile(r'^(_Z\w+|k_\w+):\s')
    for line in open(path):
        m = head.match(line)
        if m:
            name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().split('(')[0]
            kern = name if (not wanted or any(w in name for w in wanted)) and (name.startswith('void k_') or name.startswith('k_')) else None
            depth = 0
            if kern:
                res[kern] = {'loads': {}, 'stores': {}}
            continue
        if kern is None:
            continue
        if line.startswith('.LBB'):
            d = re.search(r'Depth=(\d+)', line)
            nested = re.search(r'in Loop: Header=\S+ Depth=(\d+)', line)
            depth = int(d.group(1)) if d else int(nested.group(1)) if nested else 0
            continue
        if 's_endpgm' in line:
            kern = None
            continue
        s = line.strip()
        if s.startswith('scratch_load'):
            res[kern]['loads'][depth] = res[kern]['loads'].get(depth, 0) + 1
        elif s.startswith('scratch_store'):
            res[kern]['stores'][depth] = res[kern]['stores'].get(depth, 0) + 1
    return {k: v for k, v in res.items() if v['loads'] or v['stores']}


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    r = scan(assembly(), args)
    if '--json' in sys.argv:
        print(json.dumps(r, indent=1, sort_keys=True))
    else:
        for k, v in sorted(r.items()):
            print(k)
            for what in ('loads', 'stores'):
                print('   %-6s' % what, '  '.join('depth %d: %d' % (d, n) for d, n in sorted(v[what].items())))
