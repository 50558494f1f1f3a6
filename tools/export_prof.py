"""Export the per-kernel summary (rocprofv3 --kernel-trace --stats) from a rocpd .db to CSV for profiles/."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
with open(out, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'calls', 'total_duration_us', 'average_us', 'percentage'])
    for name, calls, total, avg, pct in rows:
        w.writerow([name.split('(')[0], calls, f'{total:.3f}', f'{avg:.3f}', f'{pct:.4f}'])
print(f'wrote {len(rows)} kernels to {out}')
