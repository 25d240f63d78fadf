#!/usr/bin/env python3
"""Audit of inline-asm register loads (guide 5.7 item 1): between an asm `buffer_load_dword*` (no `lds`) and the `s_waitcnt vmcnt`
that retires it, no instruction may read or write its destination registers (a compiler copy / re-use there is silent
corruption).  Usage: python tools/asm_load_audit.py file.s kernel_name_substring
With --stores: the 16-byte buffer-store hazard instead: within the 2 wait states behind a `buffer_store_dwordx3/x4`
no VALU instruction may write its data registers (hipcc pads this pair only for stores without an SGPR offset)."""
import re
import sys

STORES = '--stores' in sys.argv
argv = [a for a in sys.argv if a != '--stores']
src = open(argv[1]).read()
pat = argv[2]
bad = 0
if STORES:
    for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):[^\n]*\n(.*?)^\.Lfunc_end', src, flags=re.M | re.S):
        name = m.group(1)
        ins = [t.strip() for t in m.group(2).splitlines()
               if t.strip() and not t.strip().startswith((';', '.')) and not t.strip().endswith(':')]
        for i, t in enumerate(ins):
            sm = re.match(r'buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\]', t)
            if not sm:
                continue
            data = set(range(int(sm.group(1)), int(sm.group(2)) + 1))
            states = 0
            for u in ins[i + 1:]:
                if states >= 2:
                    break
                op = u.split()[0]
                if op == 's_nop':
                    states += int(u.split()[1]) + 1
                    continue
                if op.startswith('v_'):
                    dm = re.match(r'\S+\s+v\[(\d+):(\d+)\]', u) or re.match(r'\S+\s+v(\d+)', u)
                    if dm:
                        d = set(range(int(dm.group(1)), int(dm.group(dm.lastindex)) + 1))
                        if d & data:
                            print(f'{name}: `{u}` writes the data of `{t}` {states} wait state(s) behind it')
                            bad += 1
                states += 1
    print('violations:', bad)
    sys.exit(1 if bad else 0)
for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):[^\n]*\n(.*?)^\.Lfunc_end', src, flags=re.M | re.S):
    name, body = m.group(1), m.group(2)
    pending = []      # list of (set of vgpr indices), in issue order; other vm ops are counted as None
    for ln, line in enumerate(body.splitlines()):
        t = line.strip()
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        op = t.split()[0]
        regs = set()
        for a, b in re.findall(r'v\[(\d+):(\d+)\]', t):
            regs.update(range(int(a), int(b) + 1))
        for a in re.findall(r'(?<![\w\[])v(\d+)\b', t):
            regs.add(int(a))
        if op == 's_waitcnt' and 'vmcnt' in t:
            n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
            keep = pending[len(pending) - n:] if n else []
            pending = keep
            continue
        live = set().union(*[p for p in pending if p]) if pending else set()
        hit = regs & live
        if hit:
            print(f'{name}: line {ln}: `{t}` touches in-flight load registers {sorted(hit)[:8]}')
            bad += 1
        if op.startswith('buffer_load') or op.startswith('global_load'):
            if ' lds' in t or t.endswith('lds'):
                pending.append(None)
            else:
                dm = re.match(r'\S+\s+v\[(\d+):(\d+)\]', t) or re.match(r'\S+\s+v(\d+)', t)
                d = set(range(int(dm.group(1)), int(dm.group(dm.lastindex)) + 1))
                pending.append(d)
        elif op.startswith('buffer_store') or op.startswith('global_store') or op.startswith('global_atomic') or op.startswith('buffer_atomic'):
            pending.append(None)
print('violations:', bad)
sys.exit(1 if bad else 0)
