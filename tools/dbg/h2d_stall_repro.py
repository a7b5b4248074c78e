"""Is the ~90 ms stall of a host-to-device copy between launches a property of the platform, not of this library?  Pure torch: runs of
4 x [one small H2D copy + 50 small kernels], bracketed by synchronize.  modes: pinned (async copy out of pinned memory), pageable, device (no
host copy).  Run with HSA_ENABLE_SDMA=1 (default) and =0 (copies by blit kernels on the compute queues instead of the SDMA engines).
    python tools/dbg/h2d_stall_repro.py [--runs 60]"""
import argparse, json, os, time
import numpy as np
import torch
ap = argparse.ArgumentParser(); ap.add_argument('--runs', type=int, default=60); args = ap.parse_args()
dev = torch.device('cuda', 0)
x = torch.zeros(65536, dtype=torch.int32, device=dev)
y = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
src = {'pinned': torch.arange(65536, dtype=torch.int32).pin_memory(), 'pageable': torch.arange(65536, dtype=torch.int32), 'device': torch.arange(65536, dtype=torch.int32, device=dev)}
out = {'HSA_ENABLE_SDMA': os.environ.get('HSA_ENABLE_SDMA', '(unset)')}
for mode, host in src.items():
    walls = []
    for r in range(args.runs + 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for ep in range(4):
            x.copy_(host, non_blocking=(mode == 'pinned'))
            for _ in range(50):
                y.add_(1.0)
        torch.cuda.synchronize(); walls.append(1e3 * (time.perf_counter() - t0))
    walls = np.array(walls[3:])
    out[mode] = {'median_ms': round(float(np.median(walls)), 3), 'max_ms': round(float(walls.max()), 3), 'runs_over_20ms': int((walls > 20).sum()), 'runs': len(walls)}
print(json.dumps(out))
