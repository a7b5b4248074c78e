"""Static instruction mix of the kernels in a hipcc -S listing: python tools/dbg/isa_mix.py file.s [name-substring ...]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
wanted = sys.argv[2:]
cur = None
stats = collections.OrderedDict()
for line in txt:
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1); stats[cur] = collections.Counter(); continue
    if line.startswith('.Lfunc_end'):
        cur = None; continue
    if cur is None: continue
    t = line.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    op = t.split()[0]
    c = stats[cur]
    if op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_load') or op.startswith('s_buffer'): c['smem'] += 1
    elif op.startswith('s_waitcnt') or op.startswith('s_nop'): c['wait'] += 1
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): c['branch'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): c['vmem'] += 1
    else: c['other'] += 1
    c['total'] += 1
    c['op:' + op] += 1
import subprocess
for k, c in stats.items():
    if wanted and not all(w in k for w in wanted): continue
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()[:100]
    print(name)
    print('   ', {x: c[x] for x in ('total', 'valu', 'salu', 'smem', 'vmem', 'lds', 'wait', 'branch')})
    top = sorted(((v, o[3:]) for o, v in c.items() if o.startswith('op:')), reverse=True)[:22]
    print('   ', ' '.join(f'{o}:{v}' for v, o in top))
