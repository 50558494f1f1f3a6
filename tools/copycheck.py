"""Similarity of each host module to the same-named reference file (difflib ratio on stripped lines)."""
import difflib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/badread'


def norm(path):
    out = []
    for line in open(path, errors='replace'):
        s = line.strip()
        if s and not s.startswith('#'):
            out.append(s)
    return out


for name in sorted(os.listdir(os.path.join(REPO, 'badread_amd'))):
    mine = os.path.join(REPO, 'badread_amd', name)
    ref = os.path.join(REF, name)
    if name.endswith('.py') and os.path.isfile(ref):
        r = difflib.SequenceMatcher(None, norm(mine), norm(ref), autojunk=False).ratio()
        rc = difflib.SequenceMatcher(None, open(mine).read(), open(ref, errors='replace').read(), autojunk=False).ratio() if os.path.getsize(mine) < 40000 else -1
        print(f'{name:24s} line-ratio {r:.2f}  char-ratio {rc:.2f}')
