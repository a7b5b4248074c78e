"""profiles/hbm_traffic.json from the per-dispatch counter CSVs of the rocprofv3 --pmc passes (tools/collect_profiles.sh):
HBM bytes per step launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB; FETCH_SIZE reports half the bytes read on gfx950, MI355X_MICROARCH.md §HBM,
re-checked on this access pattern by tools/ubench/traffic_calib.hip), averaged over the step launches of one 50-step episode.
usage: python tools/traffic_summarise.py <dir with pmc_<domain>_<COUNTER>/...counter_collection.csv> <round tag>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_fingerprint

src, tag = sys.argv[1], sys.argv[2]
STEP_KERNELS = {'wildfire': ('wf_roles_kernel', ), 'cybersecurity': ('cy_roles_kernel', ), 'rideshare': ('rs_env_kernel', 'rs_offsets_kernel', 'rs_emit_kernel')}


def per_step(domain, counter):
    files = glob.glob(os.path.join(src, f'pmc_{domain}_{counter}', '**', '*counter_collection.csv'), recursive=True)
    if not files:
        return None
    rows = list(csv.DictReader(open(files[0])))
    total, steps = 0.0, 0
    first = STEP_KERNELS[domain][0]
    for r in rows:
        name = r['Kernel_Name']
        if r['Counter_Name'] != counter or not any(k in name for k in STEP_KERNELS[domain]):
            continue
        head = name.split('(')[0]
        if 'rs_env_kernel' in name and ', 1>' in head:
            continue  # rebuild-mode instantiation (reset), not a step
        if 'wf_roles_kernel' in name:
            arguments = [a.strip() for a in head[head.index('<') + 1:head.rindex('>')].split(',')]
            if arguments[4] != '0':
                continue  # reset / rebuild mode
            # <CMAX, AMAX, EXACT, RNG, MODE, PERSIST>: a multi-step launch carries the whole 50-step episode of tools/traffic_run.py
            total += float(r['Counter_Value'])
            steps += 50 if (len(arguments) > 5 and arguments[5] in ('true', '1')) else 1
            continue
        total += float(r['Counter_Value'])
        steps += first in name
    # rideshare: reset's rebuild also runs offsets + emit once: negligible against 50 steps, left in
    return total / max(steps, 1), steps


out = {'source_fingerprint': source_fingerprint(), 'round': tag,
       'how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/traffic_run.py (one 50-step episode at B = 65536); '
              'bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per step, all launches of a step summed',
       'corrections': 'FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md §HBM; calibration: profiles/r01_pmc_*_calibration.csv)'}
for domain in STEP_KERNELS:
    f, w = per_step(domain, 'FETCH_SIZE'), per_step(domain, 'WRITE_SIZE')
    if f and w:
        out[f'{domain}_bytes_per_step'] = (2 * f[0] + w[0]) * 1024
        out[f'{domain}_read_bytes_per_env_step'] = 2 * f[0] * 1024 / 65536
        out[f'{domain}_write_bytes_per_env_step'] = w[0] * 1024 / 65536
        out[f'{domain}_steps_counted'] = f[1]
if 'wildfire_bytes_per_step' in out:  # bench.py scales it by the steps one launch of its timed region performs
    out['wf_step_kernel_bytes_per_step'] = out['wildfire_bytes_per_step']
json.dump(out, open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
