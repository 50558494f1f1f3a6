"""Timeline of ONE device batch from a rocprofv3 kernel trace (--kernel-trace --output-format csv of
`bench.py --steps 1 --warmup 1 --streams 1 --reads-per-step 65536`): for every kernel name, first start and last end relative to
the batch's first kernel, summed busy time and dispatches -- what runs beside what inside a batch (is the
widest band class of the head set, k_fin_align<16,8,65535> on its own stream, on the batch's critical path?).
    python tools/batch_timeline.py <kernel_trace.csv> > profiles/rNN_batch_timeline.json"""
import csv
import json
import sys


def main(path):
    rows = []
    for r in csv.DictReader(open(path, newline='')):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id') or r.get('Stream_Id') or ''))
    rows.sort()
    # the LAST batch of the run: everything from the last k_plan_count on (the timed step follows the warm-up step)
    starts = [i for i, r in enumerate(rows) if r[2].startswith('k_plan_count')]
    rows = rows[starts[-1]:] if starts else rows
    t0 = rows[0][0]
    per = {}
    for s, e, name, q in rows:
        p = per.setdefault(name, {'first_start_ms': (s - t0) / 1e6, 'last_end_ms': 0.0, 'busy_ms': 0.0, 'dispatches': 0, 'queues': set()})
        p['last_end_ms'] = max(p['last_end_ms'], (e - t0) / 1e6)
        p['busy_ms'] += (e - s) / 1e6
        p['dispatches'] += 1
        p['queues'].add(q)
    out = {'batch_ms': (max(r[1] for r in rows) - t0) / 1e6, 'kernels': {}}
    for name, p in sorted(per.items(), key=lambda kv: kv[1]['first_start_ms']):
        if p['busy_ms'] < 0.05:
            continue
        out['kernels'][name] = {'first_start_ms': round(p['first_start_ms'], 2), 'last_end_ms': round(p['last_end_ms'], 2),
                                'busy_ms': round(p['busy_ms'], 2), 'dispatches': p['dispatches'], 'queues': sorted(p['queues'])}
    wide = out['kernels'].get('k_fin_align<16, 8, 65535>')
    bulk = [v for k, v in out['kernels'].items() if k.startswith(('k_fin_align<1, 1, 1>', 'k_fin_align<2, 2, 2>', 'k_fin_quad'))]
    if wide and bulk:
        out['widest_class_ends_ms'] = wide['last_end_ms']
        out['bulk_final_alignments_start_ms'] = min(v['first_start_ms'] for v in bulk)
        out['bulk_final_alignments_end_ms'] = max(v['last_end_ms'] for v in bulk)
        out['widest_class_on_critical_path'] = wide['last_end_ms'] > out['bulk_final_alignments_end_ms']
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1])
