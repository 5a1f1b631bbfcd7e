#!/usr/bin/env python3
"""Per-basic-block instruction histogram of one kernel in an AMDGPU .s file (tuning aid).
usage: isa_blocks.py file.s mangled-name-substring"""
import sys, re
from collections import Counter
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); i = s.index(':', i); j = s.index('.Lfunc_end', i)
blocks, cur, name = [], Counter(), 'entry'
for line in s[i:j].splitlines():
    t = line.strip()
    if not t or t.startswith((';', '.p2align', '.amd', '.sec', '.type', '.glob')): continue
    if t.endswith(':') or re.match(r'^\.LBB\S+:', t):
        blocks.append((name, cur)); cur, name = Counter(), t.split(':')[0]; continue
    if t.startswith('.'): continue
    op = t.split()[0]
    cur[op] += 1
    if op.startswith(('s_cbranch', 's_branch')): cur['->' + t.split()[-1]] += 0
blocks.append((name, cur))
for name, c in blocks:
    n = sum(c.values())
    if n < 20: continue
    grp = Counter()
    for k, v in c.items():
        if k.startswith('v_mfma'): grp['mfma'] += v
        elif k.startswith('v_accvgpr'): grp['accmov'] += v
        elif k.startswith(('v_exp', 'v_rcp', 'v_log', 'v_sqrt', 'v_rsq')): grp['trans'] += v
        elif k.startswith('v_'): grp['valu'] += v
        elif k.startswith('ds_'): grp['lds'] += v
        elif k.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): grp['vmem'] += v
        elif k.startswith('s_nop'): grp['nop'] += v
        elif k.startswith('s_waitcnt'): grp['wait'] += v
        elif k.startswith('s_'): grp['salu'] += v
    print(f"{name:14s} n={n:5d} ", dict(grp), [k for k in c if k.startswith('->')])

if len(sys.argv) > 3:  # run-length trace of one block by category
    want = sys.argv[3]
    on, prev, run = False, None, 0
    def cat(k):
        if k.startswith('v_mfma'): return 'MFMA'
        if k.startswith('v_accvgpr'): return 'acc'
        if k.startswith(('v_exp', 'v_rcp', 'v_log', 'v_sqrt', 'v_rsq')): return 'TR'
        if k.startswith('v_'): return 'v'
        if k.startswith('ds_'): return 'LDS'
        if k.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): return 'MEM'
        if k.startswith('s_waitcnt'): return 'wait'
        if k.startswith('s_nop'): return 'nop'
        return 's'
    out = []
    for line in s[i:j].splitlines():
        t = line.strip()
        if t.startswith(want + ':'): on = True; continue
        if on and (t.endswith(':') and t.startswith('.LBB')): break
        if not on or not t or t.startswith((';', '.')): continue
        c = cat(t.split()[0])
        if c == prev: run += 1
        else:
            if prev: out.append(f"{prev}{run}" if run > 1 else prev)
            prev, run = c, 1
    out.append(f"{prev}{run}")
    print(' '.join(out))
