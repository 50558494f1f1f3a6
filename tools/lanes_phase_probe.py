"""Where a cycle of k_mut_lanes goes (BRX_PROFILE=1: shader-clock time per step, summed over the waves of one device batch alone on the chip).
    BRX_PROFILE=1 python tools/lanes_phase_probe.py [workload] [reads]"""
import io
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ['BRX_PROFILE'] = '1'
os.environ.setdefault('BRX_LANES_CYCLES', '0')      # every cycle in k_mut_lanes and no head chain: the in-place kernel writes the same counters
os.environ.setdefault('BRX_HEAD_READS', '0')
import bench  # noqa: E402
from badread_amd.engine import HipEngine  # noqa: E402

wl_name = sys.argv[1] if len(sys.argv) > 1 else 'human'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
wl = bench.build_workload(io.StringIO(), wl_name, bench.default_ref_dir())
eng = bench.configure(HipEngine(0, scratch_bytes=int(40 * (1 << 30))), wl)
eng.simulate_batch_device(42, 2 ** 40, 64)
eng.simulate_batch_device(42, 0, n)
ph = eng.phase_cycles(n).astype('float64')
tot = ph.sum(axis=0)
ph = ph[ph[:, 7] > 0]
tot = ph.sum(axis=0)
waves = len(ph)
names = ['apply', 'refill', 'park', 'whole_wave_windows', 'lane_aligner', 'cycles']
clk_mhz = 100.0                         # s_memtime ticks at 100 MHz
rec = {'workload': wl_name, 'reads': n, 'waves': waves, 'cycles_per_wave': tot[7] / max(waves, 1),
       'us_per_cycle': {nm: tot[i] / clk_mhz / max(tot[7], 1) for i, nm in enumerate(names[:5])}}
rec['us_per_cycle']['total'] = sum(rec['us_per_cycle'].values())
per_wave = ph[:, :5].sum(axis=1)
order = per_wave.argsort()[::-1]
rec['ticks_per_wave'] = {'mean': float(per_wave.mean()), 'median': float(sorted(per_wave)[len(per_wave) // 2]), 'max': float(per_wave.max())}
rec['slowest_waves'] = [{'cycles': int(ph[i, 7]), 'ticks': float(per_wave[i]), **{nm: float(ph[i, j]) for j, nm in enumerate(names[:5])}} for i in order[:6]]
rec['by_cycles'] = {}
for lo, hi in ((0, 20), (20, 40), (40, 80), (80, 10 ** 9)):
    sel = (ph[:, 7] >= lo) & (ph[:, 7] < hi)
    if sel.any():
        rec['by_cycles'][f'{lo}-{hi}'] = {'waves': int(sel.sum()), 'ticks_per_cycle': float(per_wave[sel].sum() / ph[sel, 7].sum()),
                                            'whole_wave_share': float(ph[sel, 3].sum() / per_wave[sel].sum()), 'aligner_share': float(ph[sel, 4].sum() / per_wave[sel].sum())}
print(json.dumps(rec, indent=1))
