"""
Run the REFERENCE'S OWN test files (/root/reference/test/test_*.py, unmodified, read in place) against this package:
`badread` is aliased to `badread_amd`, `edlib` to the oracle's shim, and -- there being no GPU here -- the engine
behind sequence_fragment / the model loader's aligner is the CPU checker.  This shows which of the reference's tests
read unchanged against the drop-in host interface, and which do not (functions that moved onto the device and have
no Python body here: get_fragment, add_glitches, ...).

    python tools/run_reference_tests.py [pytest args ...]        (needs /root/reference)
"""
import importlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = '/root/reference'
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'oracle', 'shim'), os.path.join(REPO, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def install_alias(engine='oracle'):
    import badread_amd
    sys.modules['badread'] = badread_amd
    for name in ('misc', 'simulate', 'error_model', 'qscore_model', 'fragment_lengths', 'identities', 'settings',
                 'version', '__main__', 'reference', 'engine'):
        sys.modules['badread.' + name] = importlib.import_module('badread_amd.' + name)
        setattr(badread_amd, name, sys.modules['badread.' + name])
    if engine == 'oracle':
        import pyoracle
        import badread_amd.engine as E
        import badread_amd.error_model as EM
        eng = pyoracle.OracleEngine()
        E.default_engine = lambda: eng
        E.hip_align_batch = pyoracle.oracle_align_batch
        EM.default_aligner = lambda: pyoracle.oracle_align_batch


def main():
    import pytest
    if not os.path.isdir(os.path.join(REFERENCE, 'test')):
        sys.exit('needs /root/reference')
    install_alias()
    args = sys.argv[1:] or ['-q', '-x', '--no-header', '-p', 'no:cacheprovider', '--rootdir', '/tmp',
                            os.path.join(REFERENCE, 'test')]
    return pytest.main(args)


if __name__ == '__main__':
    sys.exit(main())
