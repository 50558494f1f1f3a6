"""
Hidden constants of the simulator.  The VALUES are a behavioural contract with the reference
(/root/reference/badread/settings.py:24-51): the mutate loop re-aligns every ALIGNMENT_INTERVAL
applied changes over an ALIGNMENT_SIZE window, the CLI refuses means at or below the MIN_* bounds,
the synthetic qscore models use the ranges below, and each chimera junction gets an end / start
adapter with the CHIMERA_* chances.  The HIP kernels hard-code the first two
(badread_amd/csrc/brx_kernels.h: BRX_ALIGN_INTERVAL, BRX_ALIGN_SIZE) and 0.25 for the chimera
adapters; badread_amd/simulate.py asserts they agree at import time.
"""

ALIGNMENT_INTERVAL, ALIGNMENT_SIZE = 25, 1000

MIN_MEAN_READ_LENGTH, MIN_MEAN_READ_IDENTITY, MIN_MEAN_READ_QSCORE = 100, 50, 5

CHIMERA_START_ADAPTER_CHANCE = CHIMERA_END_ADAPTER_CHANCE = 0.25

# qscore ranges (inclusive) of the two synthetic models
RANDOM_QSCORE_RANGE = (1, 20)                 # 'random': every op, k = 1
IDEAL_QSCORE_RANKS = (                        # 'ideal': rank 1 = X and I; ranks 2..6 = all-'=' windows of 1,3,5,7,9
    (1, 3), (4, 7), (8, 20), (21, 30), (31, 40), (41, 50),
)

RANDOM_QSCORE_MIN, RANDOM_QSCORE_MAX = RANDOM_QSCORE_RANGE
for _rank, (_lo, _hi) in enumerate(IDEAL_QSCORE_RANKS, start=1):
    globals()[f'IDEAL_QSCORE_RANK_{_rank}_MIN'] = _lo
    globals()[f'IDEAL_QSCORE_RANK_{_rank}_MAX'] = _hi
del _rank, _lo, _hi
