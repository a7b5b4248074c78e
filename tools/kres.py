#!/usr/bin/env python3
"""Per-kernel resource usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
    python tools/kres.py free-range-zoo_amd/csrc/rideshare.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '--cuda-device-only', '-c', src,
       '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark: (?:\s*)(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (.*?) \[-Rpass', line)
    if not m:
        continue
    key, val = m.group(1), m.group(2)
    if key == 'Function Name':
        cur = subprocess.run(['c++filt', val], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][key.split(' [')[0]] = val
for name, r in rows.items():
    if flt in name:
        short = re.sub(r'\(anonymous namespace\)::', '', name)
        short = re.sub(r'\(.*', '', short)
        print(f"{short:60s} sgpr {r.get('TotalSGPRs'):>4} vgpr {r.get('VGPRs'):>4} scratch {r.get('ScratchSize'):>4} occ {r.get('Occupancy'):>2} "
              f"sspill {r.get('SGPRs Spill'):>3} vspill {r.get('VGPRs Spill'):>3} lds {r.get('LDS Size'):>6}")
