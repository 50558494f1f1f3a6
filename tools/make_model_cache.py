"""
Regenerate badread_amd/model_cache/*.npz from the reference's model files.

The reference ships its error/qscore models as gzipped text (badread/error_models/*.gz,
badread/qscore_models/*.gz).  This package ships DERIVED tables instead (parsed probabilities plus,
for error models, the inner k-mer alignment ops of every alternative), so that nothing has to read
the reference's install at run time.  Run in a container that has the reference:

    python tools/make_model_cache.py [--models nanopore2023 pacbio2021 ...] [--aligner oracle|hip]

`--aligner oracle` (default) uses the CPU checker (oracle/myers_ref.c) because this container has
no GPU; tests/test_gpu_models.py re-aligns every pair with the HIP kernel on a GPU box and asserts
the cached ops are identical.
"""
import argparse
import io
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from badread_amd.error_model import ErrorModel, BUILTIN_ERROR_MODELS, find_builtin_file  # noqa: E402
from badread_amd.qscore_model import QScoreModel, BUILTIN_QSCORE_MODELS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--models', nargs='*', default=list(BUILTIN_ERROR_MODELS))
    ap.add_argument('--aligner', default='oracle', choices=['oracle', 'hip'])
    args = ap.parse_args()
    if args.aligner == 'oracle':
        sys.path.insert(0, os.path.join(REPO, 'oracle'))
        from pyoracle import oracle_align_batch as aligner
    else:
        from badread_amd.engine import hip_align_batch as aligner
    out_dir = os.path.join(REPO, 'badread_amd', 'model_cache')
    os.makedirs(out_dir, exist_ok=True)
    null = io.StringIO()
    for name in args.models:
        if name in BUILTIN_ERROR_MODELS:
            path = find_builtin_file('error_models', name)
            em = ErrorModel(path, output=null, aligner=aligner, use_cache=False)
            em.save_npz(os.path.join(out_dir, f'{name}.error.npz'))
        if name in BUILTIN_QSCORE_MODELS:
            from badread_amd.error_model import find_builtin_file as fb
            qm = QScoreModel.__new__(QScoreModel)
            qm.scores, qm.probabilities, qm.kmer_size, qm.type, qm._tables = {}, {}, 1, None, None
            qm.load_from_file(fb('qscore_models', name), null)
            qm.save_npz(os.path.join(out_dir, f'{name}.qscore.npz'))
        print(name, 'done')


if __name__ == '__main__':
    main()
