"""
TEST INFRASTRUCTURE: one process = one slice of a batch through the CPU oracle on a bench workload (tests/test_gpu_fullsize.py
runs one per host core so that EVERY read of a 16384-read batch of the 3.1 Gb configurations is compared).
    python tests/oracle_slice_worker.py WORKLOAD REF_DIR SEED FIRST COUNT OUT.npz
"""
import io
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)

if __name__ == '__main__':
    workload, ref_dir, seed, first, count, out = sys.argv[1], (None if sys.argv[2] == '-' else sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    import bench
    import pyoracle
    eng = bench.configure(pyoracle.OracleEngine(), bench.build_workload(io.StringIO(), workload, ref_dir))
    data, stats = eng.simulate_batch(seed, first, count)
    np.savez(out, data=np.asarray(data), stats=stats)
