#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc results database (rocpd sqlite): per kernel name, mean duration and mean counter values.
usage: pmc_summary.py <results.db> [name-substring]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sfx = [t for t in tabs if t.startswith('rocpd_pmc_event_')][0][len('rocpd_pmc_event_'):]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
q = f"""select s.kernel_name, d.id, d.end - d.start, p.name, e.value
        from rocpd_kernel_dispatch_{sfx} d
        join rocpd_info_kernel_symbol_{sfx} s on s.id = d.kernel_id
        left join rocpd_pmc_event_{sfx} e on e.event_id = d.event_id
        left join rocpd_info_pmc_{sfx} p on p.id = e.pmc_id"""
acc = collections.OrderedDict()
for name, did, dur, cname, val in cur.execute(q):
    if flt and flt not in name:
        continue
    short = name.split('(')[0][-70:]
    k = acc.setdefault(short, dict(disp=set(), dur={}, ctr=collections.defaultdict(float)))
    k['disp'].add(did)
    k['dur'][did] = dur
    if cname:
        k['ctr'][cname] += val or 0.0
for name, k in acc.items():
    n = len(k['disp'])
    print(f'{name}  n={n}  avg {sum(k["dur"].values()) / n / 1e3:.1f} us')
    for c, v in sorted(k['ctr'].items()):
        print(f'    {c:28s} {v / n:16.0f}')
