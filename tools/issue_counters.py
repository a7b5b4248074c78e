"""profiles/<tag>_issue_counters.txt: instructions issued per wavefront (rocprofv3 --pmc SQ_*) and launch durations of the step kernels of
the bench workloads — the evidence behind DESIGN.md §4.2 / §4.5 ("a launch of these kernels is bound by the instructions its wavefronts
issue").  Run on the GPU box by tools/collect_issue_counters.sh; reads the pass directories it wrote.
usage: python tools/issue_counters.py <dir> <tag> <workload> ..."""
import collections, csv, glob, os, re, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = re.compile(r'((?:wf|wg|cy|rs)_[a-z_]+kernel<[^>]*>)')


def counters(workload):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(src, f'pmc_{workload}_[ab]', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            m = KERNEL.search(r['Kernel_Name'])
            if m:
                out[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    return out


def durations(workload):
    out = collections.defaultdict(list)
    for path in glob.glob(os.path.join(src, f'trace_{workload}', '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            m = KERNEL.search(r['Kernel_Name'])
            if m:
                out[m.group(1)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return out


lines = [f'# {tag}: per-wavefront issue counters and launch durations of the step kernels, B = 65536, one 50-step random-policy episode each',
         '# (rocprofv3 --pmc, eight SQ counters per pass, two passes; durations from a separate --kernel-trace run of the same command)',
         '# columns: launches, mean launch us, wavefronts per launch, then per WAVEFRONT (mean over the launches): vector, scalar, LDS, vector-memory',
         '# instructions, issue-active / waitcnt-parked / issue-stalled quad-cycles and the wavefront\'s lifetime in quad-cycles', '']
for workload in sys.argv[3:]:
    c, d = counters(workload), durations(workload)
    lines.append(f'== {workload}')
    for kernel in sorted(c):
        v = c[kernel]
        mean = lambda name: (sum(v[name]) / len(v[name])) if v.get(name) else float('nan')
        waves = mean('SQ_WAVES')
        per = lambda name: mean(name) / waves if waves == waves and waves > 0 else float('nan')
        dur = d.get(kernel, [])
        lines.append(f'{kernel:46s} n={len(dur):4d} us={sum(dur) / max(len(dur), 1):8.1f} waves={waves:8.0f}  valu={per("SQ_INSTS_VALU"):7.1f} salu={per("SQ_INSTS_SALU"):7.1f} '
                     f'lds={per("SQ_INSTS_LDS"):6.1f} vmem={per("SQ_INSTS_VMEM_RD") + per("SQ_INSTS_VMEM_WR"):6.1f}  active={per("SQ_ACTIVE_INST_ANY"):8.1f} '
                     f'parked={per("SQ_WAIT_ANY"):8.1f} stalled={per("SQ_WAIT_INST_ANY"):8.1f} lifetime={per("SQ_WAVE_CYCLES"):8.1f}')
    lines.append('')
path = os.path.join(ROOT, 'profiles', f'{tag}_issue_counters.txt')
open(path, 'w').write('\n'.join(lines))
print('\n'.join(lines))
