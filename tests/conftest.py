import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


# A set of the final stage with fewer than BRX_LANES_MIN_READS (2048) reads for the by-lane aligner keeps them on whole waves: the few dozen
# reads of a test case would never reach k_fin_lanes.  The suite runs with the rule off; the full-size tests (tests/test_gpu_fullsize.py)
# take the variable away again and run the shipped default.
os.environ.setdefault('BRX_LANES_MIN_READS', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason='no ROCm device in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
