"""
The reference's OWN test files, unmodified and read in place from /root/reference/test, run against this package
(tools/run_reference_tests.py: `badread` aliased to `badread_amd`, `edlib` to the oracle's shim, the CPU checker
as the engine because this container has no GPU).  Skipped where /root/reference does not exist (the GPU box).

What must pass: everything in the files below that exercises the mirrored host interface -- misc helpers, fragment
length and identity laws, target size, reference loading, ErrorModel / QScoreModel loading and sampling, align_kmers,
and test_simulate.py (sequence_fragment: perfect fragments and the identity tolerances of all six error models).
What cannot: tests that call Python internals which no longer exist as Python because they run inside the kernels
(build_fragment, get_fragment, add_glitches, adapters, get_qscores: their semantics are pinned by the replay goldens
instead, tests/test_golden_oracle.py) and the model-BUILDING commands (out of scope).
"""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE_TESTS = '/root/reference/test'

MUST_PASS_FILES = ['test_fragment_lengths.py', 'test_identities.py', 'test_target_size.py', 'test_references.py',
                   'test_simulate.py']
# per file: (-k expression of what is in scope, minimum number of passing tests)
PARTIAL = {
    'test_misc.py': ('not LoadSequences and not test_fastq and not wrong_type', 30),
    'test_error_model.py': ('not MakeErrorModel', 15),
    'test_qscore_model.py': ('not MakeQScoreModel and not GetQScores and not Bugs', 10),
}


def run(paths, k=None):
    cmd = [sys.executable, os.path.join(REPO, 'tools', 'run_reference_tests.py'), '-q', '--no-header', '-p', 'no:cacheprovider',
           '--rootdir', '/tmp'] + (['-k', k] if k else []) + paths
    r = subprocess.run(cmd, cwd='/tmp', capture_output=True, text=True, timeout=1500)
    tail = r.stdout.strip().split('\n')[-1]
    counts = {m.group(2): int(m.group(1)) for m in re.finditer(r'(\d+) (passed|failed|error|errors|deselected)', tail)}
    return r, counts


@pytest.mark.skipif(not os.path.isdir(REFERENCE_TESTS), reason='the reference is not on this machine')
def test_reference_test_files_that_must_pass_unchanged():
    r, counts = run([os.path.join(REFERENCE_TESTS, f) for f in MUST_PASS_FILES])
    assert r.returncode == 0 and counts.get('failed', 0) == 0 and counts.get('passed', 0) >= 35, r.stdout[-3000:]


@pytest.mark.skipif(not os.path.isdir(REFERENCE_TESTS), reason='the reference is not on this machine')
@pytest.mark.parametrize('name', sorted(PARTIAL))
def test_reference_test_files_in_scope_parts(name):
    k, at_least = PARTIAL[name]
    r, counts = run([os.path.join(REFERENCE_TESTS, name)], k)
    assert r.returncode == 0 and counts.get('failed', 0) == 0 and counts.get('passed', 0) >= at_least, r.stdout[-3000:]
