#!/usr/bin/env python3
"""Static instruction mix of the kernels in a hipcc -S listing: python tools/isa_mix.py listing.s <name filter> [<name filter> ...]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\S+:', l)]
for idx, (i, name) in enumerate(starts):
    if not any(f in name for f in sys.argv[2:]):
        continue
    j = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    ins = [l.strip().split()[0] for l in lines[i + 1:j] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter()
    for k in ins:
        c['salu' if k.startswith('s_') else 'valu' if k.startswith('v_') else 'vmem' if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'lds' if k.startswith('ds_') else 'other'] += 1
    print(name[:90], len(ins), dict(c))
    print('   ', collections.Counter(ins).most_common(16))
