#!/usr/bin/env python3
"""per-kernel duration statistics (the table `rocprofv3 --stats` prints) from a rocpd results database:
    python tools/kernel_stats_from_db.py <results.db> <out.csv> [steps]
columns: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs [, CallsPerStep, MsPerStep]"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 0
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sfx = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0][len('rocpd_kernel_dispatch_'):]
rows = list(cur.execute(f"""select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                            from rocpd_kernel_dispatch_{sfx} d join rocpd_info_kernel_symbol_{sfx} s on s.id = d.kernel_id
                            group by s.kernel_name order by 3 desc"""))
total = sum(r[2] for r in rows)
with open(out, 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'] + (['CallsPerStep', 'MsPerStep'] if steps else []))
    for name, n, tot, mn, mx in rows:
        w.writerow([name, n, tot, round(tot / n, 1), round(100.0 * tot / total, 3), mn, mx] +
                   ([round(n / steps, 2), round(tot / steps / 1e6, 4)] if steps else []))
print(f'{len(rows)} kernels, {total / 1e6:.2f} ms of kernel time' + (f', {total / steps / 1e6:.2f} ms per step' if steps else ''))
