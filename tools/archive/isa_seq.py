#!/usr/bin/env python3
"""Compressed instruction sequence (memory ops, MFMA, waits, barriers, branches) of one kernel in a .s file."""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ':')
j = s.index('s_endpgm', i)
seq = []
for l in s[i:j].split('\n'):
    l = l.strip()
    m = re.match(r'(global_load_dwordx4|global_load_dword\w*|v_mfma_\w+|s_waitcnt|s_barrier|ds_write_b\d+|ds_read_b\d+|s_cbranch\w+|global_store\w+|global_atomic\w+|\.LBB\w+:|buffer_\w+|scratch_\w+)', l)
    if m:
        k = m.group(1)
        if k == 's_waitcnt':
            k = l
        if seq and seq[-1][0] == k:
            seq[-1][1] += 1
        else:
            seq.append([k, 1])
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 100
print('\n'.join(k if c == 1 else "%s x%d" % (k, c) for k, c in seq[:lim]))
