"""profiles/hbm_traffic.json from the per-dispatch counter CSVs of the rocprofv3 --pmc passes (tools/collect_profiles.sh):
HBM bytes per step launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB; FETCH_SIZE reports half the bytes read on gfx950, MI355X_MICROARCH.md §HBM,
re-checked on this access pattern by tools/ubench/traffic_calib.hip), averaged over the step launches of one 50-step episode.
usage: python tools/traffic_summarise.py <dir with pmc_<domain>_<COUNTER>/...counter_collection.csv> <round tag>"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_fingerprint

src, tag = sys.argv[1], sys.argv[2]
STEP_KERNELS = {'wildfire': ('wf_roles_kernel', ), 'cybersecurity': ('cy_roles_kernel', ), 'rideshare': ('rs_env_kernel', 'rs_offsets_kernel', 'rs_emit_kernel'),
                'wildfire_grid_8x8': ('wg_env_kernel', 'wg_lists_kernel'), 'wildfire_grid_16x16': ('wg_env_kernel', 'wg_lists_kernel'),
                # A/B of round 4: the same cybersecurity episode with libfrz_hip_cy_direct.so — every lane stores its own task rows
                'cybersecurity_direct': ('cy_roles_kernel', )}


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
        if re.search(r'rs_env_kernel<[^>]*, 1>', name):
            continue  # rebuild-mode instantiation (reset), not a step
        grid = re.search(r'wg_env_kernel<([^>]*)>', name)  # <CPL, MODE, RNG>
        if grid and grid.group(1).split(',')[1].strip() != '0':
            continue  # reset / rebuild mode (its lists launch is counted: negligible against 50 steps)
        roles = re.search(r'wf_roles_kernel<([^>]*)>', name)
        if roles:
            arguments = [a.strip() for a in roles.group(1).split(',')]
            if arguments[4] != '0':
                continue  # reset / rebuild mode
            # <CMAX, AMAX, EXACT, RNG, MODE, PERSIST>: a multi-step launch carries the whole 50-step episode of tools/traffic_run.py
            total += float(r['Counter_Value'])
            steps += 50 if (len(arguments) > 5 and arguments[5] in ('true', '1')) else 1
            continue
        roles = re.search(r'cy_roles_kernel<([^>]*)>', name)  # <NMAX, AMAX, ATT, RNG, PERSIST, EXTRA>
        if roles and len(roles.group(1).split(',')) > 4 and roles.group(1).split(',')[4].strip() in ('true', '1'):
            total += float(r['Counter_Value'])
            steps += 50  # a multi-step launch carries the whole 50-step episode of tools/traffic_run.py
            continue
        total += float(r['Counter_Value'])
        steps += first in name
    # rideshare: reset's rebuild also runs offsets + emit once: negligible against 50 steps, left in
    return total / max(steps, 1), steps


def short_name(name):
    """`wf_roles_kernel<6, 3, true, 1, 0, true>` out of rocprofv3's full signature."""
    m = re.search(r'((?:wf|wg|cy|rs)_[a-z_]+kernel)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else None


def write_reduced(domain, counter):
    """profiles/<tag>_pmc_<COUNTER>_<domain>_step.csv: the per-dispatch counter values of this library's kernels (one row per dispatch)."""
    files = glob.glob(os.path.join(src, f'pmc_{domain}_{counter}', '**', '*counter_collection.csv'), recursive=True)
    if not files:
        return
    path = os.path.join(ROOT, 'profiles', f'{tag}_pmc_{counter}_{domain}_step.csv')
    with open(path, 'w', newline='') as handle:
        w = csv.writer(handle)
        w.writerow(['Dispatch_Id', 'Kernel', 'Grid_Size', 'Workgroup_Size', 'VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Counter_Name', 'Counter_Value_KiB',
                    'Duration_ns'])
        for r in csv.DictReader(open(files[0])):
            kernel = short_name(r['Kernel_Name'])
            if kernel and r['Counter_Name'] == counter:
                w.writerow([r['Dispatch_Id'], kernel, r['Grid_Size'], r['Workgroup_Size'], r['VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], counter,
                            f"{float(r['Counter_Value']):.6f}", int(r['End_Timestamp']) - int(r['Start_Timestamp'])])


out = {'source_fingerprint': source_fingerprint(), 'round': tag,
       'how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/traffic_run.py (one 50-step episode at B = 65536); '
              'bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per step, all launches of a step summed',
       'corrections': 'FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md §HBM; calibration: profiles/r01_pmc_*_calibration.csv)'}
for domain in STEP_KERNELS:
    f, w = per_step(domain, 'FETCH_SIZE'), per_step(domain, 'WRITE_SIZE')
    write_reduced(domain, 'FETCH_SIZE'), write_reduced(domain, 'WRITE_SIZE')
    if f and w:
        out[f'{domain}_bytes_per_step'] = (2 * f[0] + w[0]) * 1024
        out[f'{domain}_read_bytes_per_env_step'] = 2 * f[0] * 1024 / 65536
        out[f'{domain}_write_bytes_per_env_step'] = w[0] * 1024 / 65536
        out[f'{domain}_steps_counted'] = f[1]
if 'wildfire_bytes_per_step' in out:  # the 50-step launch of tools/traffic_run.py wildfire: opening reset + 50 steps + episode metrics
    out['wildfire_rollout50_bytes_per_launch'] = out['wildfire_bytes_per_step'] * 50


def rollout_launch(tagname, steps):
    """bytes of the ONE multi-step dispatch of `tools/traffic_run.py wildfire<steps>` (pass directories pmc_<tagname>_<COUNTER>)"""
    values = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        files = glob.glob(os.path.join(src, f'pmc_{tagname}_{counter}', '**', '*counter_collection.csv'), recursive=True)
        if not files:
            return None
        def multi_step(name):  # wf_roles_kernel<CMAX, AMAX, EXACT, RNG, MODE, PERSIST, EXTRA>: a step-mode multi-step instantiation
            m = re.search(r'wf_roles_kernel<([^>]*)>', name)
            if not m:
                return False
            arguments = [a.strip() for a in m.group(1).split(',')]
            return arguments[4] == '0' and len(arguments) > 5 and arguments[5] in ('true', '1')
        rows = [r for r in csv.DictReader(open(files[0])) if r['Counter_Name'] == counter and multi_step(r['Kernel_Name'])]
        if len(rows) != 1:
            return None
        values[counter] = float(rows[0]['Counter_Value'])
    return (2 * values['FETCH_SIZE'] + values['WRITE_SIZE']) * 1024


for steps in (20, ):
    b = rollout_launch(f'wildfire{steps}', steps)
    if b:
        out[f'wildfire_rollout{steps}_bytes_per_launch'] = b
        out[f'wildfire_rollout{steps}_bytes_per_env_step'] = b / 65536 / steps
json.dump(out, open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
