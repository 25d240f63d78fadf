#!/usr/bin/env python3
"""Timeline of ONE train step from a rocprofv3 --kernel-trace rocpd database: per queue (= plan lane) busy time, how long 1 / 2 / 3+
kernels run at once, idle gaps, and the time-ordered list of phases (kernel families by the lane-0 stream).
    python tools/step_timeline.py <results.db> [out.json]"""
import collections
import json
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sfx = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0][len('rocpd_kernel_dispatch_'):]
cols = [r[1] for r in cur.execute(f"pragma table_info(rocpd_kernel_dispatch_{sfx})")]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = list(cur.execute(f"""select s.kernel_name, d.start, d.end, {('d.' + qcol) if qcol else '0'} from rocpd_kernel_dispatch_{sfx} d
                            join rocpd_info_kernel_symbol_{sfx} s on s.id = d.kernel_id order by d.start"""))
ad = [i for i, r in enumerate(rows) if 'adamw' in r[0]]
# one step = after the (n-3)rd optimizer launch to the (n-1)st (the optimizer is two launches per step)
lo, hi = ad[-5] + 1, ad[-3] + 1
step = rows[lo:hi]
t0, t1 = step[0][1], max(r[2] for r in step)
print(f'step: {len(step)} dispatches, {(t1 - t0) / 1e6:.3f} ms wall, {sum(r[2] - r[1] for r in step) / 1e6:.3f} ms summed kernel time')


def short(n):
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)
    n = re.sub(r'^_Z\d+', '', n)
    return re.split(r'I[A-Za-z0-9_]*E*v|E[v0-9]|Pv|PK', n)[0][:28]


# concurrency histogram
ev = []
for n, s, e, q in step:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
lvl, last, hist = 0, t0, collections.Counter()
for t, dlt in ev:
    hist[lvl] += t - last
    last = t
    lvl += dlt
print('time with k kernels running:', {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
byq = collections.defaultdict(float)
for n, s, e, q in step:
    byq[q] += e - s
print('busy per queue (ms):', {q: round(v / 1e6, 3) for q, v in sorted(byq.items(), key=lambda kv: -kv[1])})
# time each kernel family runs ALONE (no other kernel overlapping)
alone = collections.Counter()
act = {}
evs = sorted([(s, 0, i) for i, (n, s, e, q) in enumerate(step)] + [(e, 1, i) for i, (n, s, e, q) in enumerate(step)])
last = t0
for t, kind, i in evs:
    if len(act) == 1:
        alone[short(next(iter(act.values())))] += t - last
    last = t
    if kind == 0:
        act[i] = step[i][0]
    else:
        act.pop(i, None)
print('time running ALONE by kernel (ms):')
for k, v in alone.most_common(25):
    print(f'   {k:30s} {v / 1e6:.3f}')
if len(sys.argv) > 2:
    json.dump(dict(wall_ms=(t1 - t0) / 1e6, concurrency={k: v / 1e6 for k, v in hist.items()}, alone={k: v / 1e6 for k, v in alone.items()}),
              open(sys.argv[2], 'w'), indent=1)
