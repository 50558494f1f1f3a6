"""
Hidden constants of the simulator.  The VALUES are a behavioural contract with the reference
(/root/reference/badread/settings.py:24-51): the mutate loop re-aligns every ALIGNMENT_INTERVAL
applied changes over an ALIGNMENT_SIZE window, the CLI refuses means below the MIN_* bounds, and
the synthetic qscore models use the ranges below.  The HIP kernels hard-code 25 / 1000
(badread_amd/csrc/brx_kernels.hip: BRX_ALIGN_INTERVAL, BRX_ALIGN_SIZE) and the host asserts they
agree at import time.
"""

ALIGNMENT_INTERVAL = 25
ALIGNMENT_SIZE = 1000

MIN_MEAN_READ_LENGTH = 100
MIN_MEAN_READ_IDENTITY = 50
MIN_MEAN_READ_QSCORE = 5

RANDOM_QSCORE_MIN, RANDOM_QSCORE_MAX = 1, 20

IDEAL_QSCORE_RANK_1_MIN, IDEAL_QSCORE_RANK_1_MAX = 1, 3
IDEAL_QSCORE_RANK_2_MIN, IDEAL_QSCORE_RANK_2_MAX = 4, 7
IDEAL_QSCORE_RANK_3_MIN, IDEAL_QSCORE_RANK_3_MAX = 8, 20
IDEAL_QSCORE_RANK_4_MIN, IDEAL_QSCORE_RANK_4_MAX = 21, 30
IDEAL_QSCORE_RANK_5_MIN, IDEAL_QSCORE_RANK_5_MAX = 31, 40
IDEAL_QSCORE_RANK_6_MIN, IDEAL_QSCORE_RANK_6_MAX = 41, 50

CHIMERA_START_ADAPTER_CHANCE = 0.25
CHIMERA_END_ADAPTER_CHANCE = 0.25
