#!/usr/bin/env python3
"""per-dispatch PMC values of a rocprofv3 --pmc run:  python tools/pmc_dump.py <results.db> [name-filter]"""
import collections
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sfx = [t for t in tabs if t.startswith('rocpd_pmc_event_')][0][len('rocpd_pmc_event_'):]
cols = [r[1] for r in cur.execute(f"pragma table_info(rocpd_pmc_event_{sfx})")]
pcols = [r[1] for r in cur.execute(f"pragma table_info(rocpd_info_pmc_{sfx})")]
q = f"""select s.kernel_name, d.id, d.start, d.end - d.start, p.name, e.value from rocpd_kernel_dispatch_{sfx} d
        join rocpd_info_kernel_symbol_{sfx} s on s.id = d.kernel_id
        left join rocpd_pmc_event_{sfx} e on e.event_id = d.event_id
        left join rocpd_info_pmc_{sfx} p on p.id = e.pmc_id"""
rows = collections.OrderedDict()
for name, did, st, dur, pname, val in cur.execute(q):
    r = rows.setdefault(did, [name, st, dur, collections.OrderedDict()])
    if pname:
        r[3][pname] = r[3].get(pname, 0.0) + (val or 0.0)
for name, st, dur, vals in sorted(rows.values(), key=lambda r: r[1]):
    if flt in name:
        print(f'{name[:70]:70s} {dur / 1e3:9.1f} us ', ' '.join(f'{k}={v:.0f}' for k, v in vals.items()))
