"""Durations (us) of every dispatch of the wg_* / named kernels in a rocprofv3 kernel trace, in launch order: python tools/dbg/trace_series.py trace.csv [substr]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else 'wg_'
d = collections.defaultdict(list)
for r in rows:
    m = re.search(r'(\w+<[^>]*>)', r['Kernel_Name'])
    if m and want in m.group(1):
        d[m.group(1)].append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
for k, v in d.items():
    v.sort()
    print(k, len(v), 'mean of last 50: %.1f' % (sum(x[1] for x in v[-50:]) / max(1, len(v[-50:]))))
    print('   ', [round(x[1]) for x in v[-50:]])
