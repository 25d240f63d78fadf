#!/usr/bin/env python3
"""HBM traffic of ONE train step per kernel, from two rocprofv3 PMC passes over a short bench.py run:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d F -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-times
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d W -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-times
    python tools/pmc_step_traffic.py F/<run>/<pid>_results.db W/<run>/<pid>_results.db profiles/r01_pmc_step_traffic

FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies the 128-byte requests of 16-B/lane streaming reads at 64 B);
units KB.  The last complete step (between two optimizer launches) is summed per kernel symbol."""
import collections
import json
import re
import sqlite3
import sys


def load(path):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    sfx = [t for t in tabs if t.startswith('rocpd_pmc_event_')][0][len('rocpd_pmc_event_'):]
    q = f"""select s.kernel_name, d.id, d.start, e.value from rocpd_kernel_dispatch_{sfx} d
            join rocpd_info_kernel_symbol_{sfx} s on s.id = d.kernel_id
            left join rocpd_pmc_event_{sfx} e on e.event_id = d.event_id"""
    rows = collections.OrderedDict()
    for name, did, st, val in cur.execute(q):
        r = rows.setdefault(did, [name, st, 0.0])
        r[2] += val or 0.0
    return sorted(rows.values(), key=lambda r: r[1])


def short(n):
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)
    m = re.match(r'([a-z_0-9]+?)(_kernel)?(I.*?E{1,3}v|E)', n)
    return n[:64]


def last_step(rows):
    ad = [i for i, r in enumerate(rows) if 'adamw' in r[0]]
    lo, hi = (ad[-3] + 1 if len(ad) >= 3 else 0), ad[-1] + 1
    acc = collections.OrderedDict()
    for n, _, v in rows[lo:hi]:
        a = acc.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += v
    return acc


f, w = last_step(load(sys.argv[1])), last_step(load(sys.argv[2]))
out = []
for k in f:
    out.append(dict(kernel=k, launches=f[k][0], fetch_mb=round(f[k][1] * 2 / 1e3, 1), write_mb=round(w.get(k, [0, 0.0])[1] / 1e3, 1)))
out.sort(key=lambda r: -(r['fetch_mb'] + r['write_mb']))
tot_f, tot_w = sum(r['fetch_mb'] for r in out), sum(r['write_mb'] for r in out)
nt = [r for r in out if 'gemm_nt_' in r['kernel']]      # gemm_nt_kernel, gemm_nt_pp_kernel, gemm_nt_r3_kernel
summary = dict(step_fetch_gb=round(tot_f / 1e3, 2), step_write_gb=round(tot_w / 1e3, 2),
               gemm_nt=dict(launches=sum(r['launches'] for r in nt), fetch_mb=round(sum(r['fetch_mb'] for r in nt), 1),
                            write_mb=round(sum(r['write_mb'] for r in nt), 1)), kernels=out)
json.dump(summary, open(sys.argv[3] + '.json', 'w'), indent=1)
with open(sys.argv[3] + '.md', 'w') as fh:
    fh.write('# HBM traffic of one train step per kernel (rocprofv3 PMC, see tools/pmc_step_traffic.py)\n\n')
    fh.write(f'ga_convnext_tiny_768, bf16, B=256: FETCH_SIZE x2 = {tot_f / 1e3:.1f} GB, WRITE_SIZE = {tot_w / 1e3:.1f} GB per step '
             f'({(tot_f + tot_w) / 1e3:.1f} GB).\n\n| kernel symbol | launches | fetch x2 MB | write MB |\n|---|---|---|---|\n')
    for r in out[:40]:
        fh.write(f"| `{short(r['kernel'])}` | {r['launches']} | {r['fetch_mb']} | {r['write_mb']} |\n")
print(json.dumps({k: v for k, v in summary.items() if k != 'kernels'}))
