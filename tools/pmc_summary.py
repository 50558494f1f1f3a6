"""Per-kernel totals of rocprofv3 --pmc counters: python tools/pmc_summary.py <counter_collection.csv>... > out.csv

One row per (kernel, counter): dispatches, sum, mean per dispatch, mean duration (ms).  Several CSVs
(separate PMC passes) may be given; rows are concatenated.
"""
import collections
import csv
import sys


def main(paths):
    w = csv.writer(sys.stdout)
    w.writerow(['counter', 'kernel', 'dispatches', 'sum', 'mean', 'mean_duration_ms', 'vgpr', 'lds_bytes'])
    for path in paths:
        acc = collections.OrderedDict()
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                name = row['Kernel_Name'].split('(')[0].replace('void ', '')
                key = (row['Counter_Name'], name)
                a = acc.setdefault(key, [0, 0.0, 0.0, row['VGPR_Count'], row['LDS_Block_Size']])
                a[0] += 1
                a[1] += float(row['Counter_Value'])
                a[2] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6
        for (counter, name), (n, total, dur, vgpr, lds) in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
            w.writerow([counter, name, n, f'{total:.1f}', f'{total / n:.1f}', f'{dur / n:.3f}', vgpr, lds])


if __name__ == '__main__':
    main(sys.argv[1:])
