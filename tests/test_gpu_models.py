"""
The packaged error-model tables (badread_amd/model_cache/*.error.npz) hold, for every alternative of every k-mer,
the ops of the inner alignment that error_model.align_kmers needs (/root/reference/badread/error_model.py:179-229,
edlib call at :202).  They were produced with the CPU checker (tools/make_model_cache.py); a model given as a FILE
is aligned at load time on the GPU (one brx_align_batch for the whole model).  Here every packaged pair goes
through the HIP kernel again and must give the cached ops -- 425 984 alternatives per model.
"""
import io

import numpy as np
import pytest

from badread_amd.error_model import ErrorModel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['nanopore2023', 'pacbio2021', 'nanopore2018'])
def test_packaged_alignments_equal_the_hip_aligner(name):
    em = ErrorModel(name, io.StringIO())
    cached = em._ops
    assert cached is not None and len(cached) == 4 ** em.kmer_size
    em._aligner = None                      # default: badread_amd.engine.hip_align_batch
    em._align_all()
    n = 0
    for r, (a_row, b_row) in enumerate(zip(cached, em._ops)):
        assert len(a_row) == len(b_row)
        for a, b in zip(a_row, b_row):
            assert np.array_equal(np.asarray(a, dtype=np.uint8), np.asarray(b, dtype=np.uint8)), (name, em._kmers[r])
            n += 1
    assert n > 100000
