"""
The per-kernel table of DESIGN.md section 5 from one round's counter passes (profiles/<tag>_pmc_per_kernel.csv, tools/profile_round.sh):
wave-instructions per simulated base, share of the path's wave-cycles, fraction of them issuing / parked in s_waitcnt, time per
launch alone on the chip (kernels are serialised under counter collection) and FETCH x 2 + WRITE bytes per simulated base.
    python tools/kernel_table.py profiles/r06_pmc_per_kernel.csv profiles/valu_per_base.json [workload]
"""
import csv
import json
import sys


def main(path, valu_json, workload='human'):
    bases = float(json.load(open(valu_json))[workload]['bases_counted'])
    rows = {}
    for r in csv.DictReader(open(path)):
        k = rows.setdefault(r['kernel'], {})
        # SQ_WAVE_CYCLES is collected in two passes: keep the larger dispatch count's sum once
        if r['counter'] in k and r['counter'] == 'SQ_WAVE_CYCLES':
            k[r['counter']] = max(k[r['counter']], float(r['sum']))
        else:
            k[r['counter']] = float(r['sum'])
        k['ms'] = float(r['mean_duration_ms']); k['n'] = int(r['dispatches']); k['vgpr'] = r['vgpr']
    total_wc = sum(k.get('SQ_WAVE_CYCLES', 0.0) for k in rows.values())
    tot = {'valu': 0.0, 'f': 0.0, 'w': 0.0}
    print('| kernel | VALU per base | share of wave-cycles | issuing | parked in s_waitcnt (SQ_WAIT_ANY) | ms per launch (alone) | fetch x2 + write, B per base |')
    print('|---|---|---|---|---|---|---|')
    for name, k in sorted(rows.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0.0)):
        wc = k.get('SQ_WAVE_CYCLES', 0.0)
        if wc <= 0:
            continue
        valu = k.get('SQ_INSTS_VALU', 0.0) / bases
        f = 2.0 * k.get('FETCH_SIZE', 0.0) * 1024.0 / bases       # KB, 64-byte requests counted for 128 (MI355X_MICROARCH.md)
        w = k.get('WRITE_SIZE', 0.0) * 1024.0 / bases
        tot['valu'] += valu; tot['f'] += f; tot['w'] += w
        if wc / total_wc < 0.003 and valu < 0.05:
            continue
        print('| `%s` | %.2f | %.1f %% | %.2f | %.2f | %.2f | %.1f + %.1f |' % (
            name, valu, 100.0 * wc / total_wc, k.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, k.get('SQ_WAIT_ANY', 0.0) / wc, k['ms'], f, w))
    print('| **whole path** | **%.1f** | | | | | **%.0f** (%.0f + %.0f) |' % (tot['valu'], tot['f'] + tot['w'], tot['f'], tot['w']))


if __name__ == '__main__':
    main(*sys.argv[1:4])
