#!/usr/bin/env python3
"""Instruction mix of one kernel of /tmp/ssp.s (tools/isa_mix.sh): between the first and the last MFMA, and after it."""
import sys
from collections import Counter
s = open('/tmp/ssp.s').read()
name = sys.argv[1]
i = s.index(name + ':'); j = s.index('s_endpgm', i)
body = s[i:j].split('\n')
mf = [k for k, l in enumerate(body) if 'v_mfma' in l]
def mix(lo, hi, tag):
    c = Counter()
    for l in body[lo:hi]:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.'): continue
        c[l.split()[0]] += 1
    valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
    print(tag, 'MFMA', c.get('v_mfma_f32_32x32x2_f32', 0), 'VALU', valu, 'LDS', sum(v for k, v in c.items() if k.startswith('ds_')),
          'SALU', sum(v for k, v in c.items() if k.startswith('s_')), 'scratch', sum(v for k, v in c.items() if k.startswith('scratch_')))
    print('   ', ', '.join('%s %d' % kv for kv in c.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 24)))
mix(mf[0], mf[-1] + 1, 'MFMA LOOP:')
mix(mf[-1] + 1, len(body), 'AFTER    :')
