"""Where the ~100 ms stall of bench.py's per-step API leg comes from (python_api_runs_ms = [2.6, 2.5, 106.6, 2.5, 2.5]).

The leg of bench.py is replayed as it stands there (4 episodes x [reset(seed = host tensor) + 50 x step_random_policy], bracketed by synchronize),
RUNS times, with: a perf_counter stamp per call (reset / step), the garbage collector's own callbacks (generation, duration), a HIP event pair
around each run (device time of the run), the device error word after each run (a bounded look-back spin that gave up sets FRZ_ERR_SCAN_TIMEOUT).
The slowest call of every slow run is printed with what overlapped it.

    python tools/dbg/api_stall_probe.py [--runs 40] [--gc on|off|freeze] [--seeds host|device] [--pre-blocks 180]
HIP_FORCE_DEV_KERNARG is taken from the environment (bench.py sets 1)."""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--runs', type=int, default=40)
ap.add_argument('--gc', choices=['on', 'off', 'freeze'], default='on')
ap.add_argument('--seeds', choices=['host', 'device'], default='host')
ap.add_argument('--threads', type=int, default=0, help='torch.set_num_threads(n) before anything runs (0: leave the default)')
ap.add_argument('--pre-blocks', type=int, default=180, help='20-step rollout launches enqueued first, as bench.py\'s timed blocks are')
args = ap.parse_args()

if args.threads:
    torch.set_num_threads(args.threads)


def cgroup_cpu():
    """(quota, throttled periods, throttled microseconds) of this process's cgroup: CFS bandwidth throttling freezes every thread of the
    cgroup until the next 100 ms period once the quota is spent."""
    out = {}
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.stat'):
        try:
            out[path] = open(path).read().split()
        except OSError:
            pass
    return out


import configs  # noqa: E402
from free_range_zoo_amd import _capi  # noqa: E402
from free_range_zoo_amd.envs import wildfire_v0  # noqa: E402
from free_range_zoo_amd.utils.env import stream_ptr  # noqa: E402

EPISODE, B = 50, 65536
device = torch.device('cuda', 0)
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=EPISODE, device=device, rng='philox',
                               exact_shapes=False)
exclusive = env.set_exclusive_device(True)
base_seed = torch.arange(B, dtype=torch.int32)
if args.seeds == 'device':
    base_seed = base_seed.to(device)
env.reset(seed=base_seed)
metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device=device)

# what precedes the leg in bench.py: the timed blocks (one 20-step launch each, synchronised)
spec = _capi.frz_rollout_spec()
spec.n_steps, spec.rng_mode, spec.policy_seed, spec.first_step = 20, _capi.FRZ_RNG_PHILOX, 20260104, 0
spec.flags, spec.seed_increment = _capi.FRZ_ROLLOUT_RESET_FIRST, 1000003
spec.actions_out, spec.metrics = env._actions.data_ptr(), metrics.data_ptr()
for _ in range(args.pre_blocks):
    _capi.check(env._lib.frz_wildfire_rollout(env._handle, ctypes.byref(spec), stream_ptr(device)), 'rollout')
    torch.cuda.synchronize(device)

gc_events = []
_gc_t0 = [0.0]


def on_gc(phase, info):
    if phase == 'start':
        _gc_t0[0] = time.perf_counter()
    else:
        gc_events.append((_gc_t0[0], time.perf_counter(), info['generation'], info['collected']))


gc.callbacks.append(on_gc)
if args.gc == 'off':
    gc.disable()
elif args.gc == 'freeze':
    gc.collect()
    gc.freeze()

state = {'step': 0, 'episode': 0}
stamps = []


def one_step():
    if state['step'] % EPISODE == 0:
        t = time.perf_counter()
        env.reset(seed=base_seed + 1000003 * state['episode'])
        stamps.append(('reset', t, time.perf_counter()))
        state['episode'] += 1
    t = time.perf_counter()
    env.step_random_policy(policy_seed=20260104, policy_step=state['step'] % EPISODE)
    stamps.append(('step', t, time.perf_counter()))
    state['step'] += 1


for _ in range(20):
    one_step()
CGROUP_BEFORE = cgroup_cpu()
runs = []
for r in range(args.runs):
    state['step'] = 0
    del stamps[:]
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t1 = time.perf_counter()
    e0.record()
    for _ in range(4 * EPISODE):
        one_step()
    e1.record()
    t_enq = time.perf_counter()
    torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    flags = int(env._error_flags.item())
    worst = max(stamps, key=lambda s: s[2] - s[1])
    over = [(g[2], round(1e3 * (g[1] - g[0]), 3), g[3]) for g in gc_events if g[1] >= t1 and g[0] <= t2]
    runs.append({'run': r, 'wall_ms': round(1e3 * (t2 - t1), 3), 'enqueue_ms': round(1e3 * (t_enq - t1), 3), 'device_ms': round(e0.elapsed_time(e1), 3),
                 'worst_call': worst[0], 'worst_call_ms': round(1e3 * (worst[2] - worst[1]), 3), 'worst_call_index': stamps.index(worst),
                 'resets_ms': [round(1e3 * (s[2] - s[1]), 3) for s in stamps if s[0] == 'reset'],
                 'gc_during_run': over, 'error_flags': flags})
walls = np.array([r['wall_ms'] for r in runs])
print(json.dumps({'torch_threads': torch.get_num_threads(), 'cpus_visible': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0)),
                  'cgroup_before': CGROUP_BEFORE, 'cgroup_after': cgroup_cpu()}))
print(json.dumps({'gc': args.gc, 'seeds': args.seeds, 'dev_kernarg': os.environ.get('HIP_FORCE_DEV_KERNARG'), 'exclusive': bool(exclusive),
                  'wall_ms_median': float(np.median(walls)), 'wall_ms_max': float(walls.max()), 'slow_runs': int((walls > 2 * np.median(walls)).sum()),
                  'gc_counts': gc.get_count(), 'gc_threshold': gc.get_threshold()}))
for r in runs:
    if r['wall_ms'] > 1.5 * np.median(walls) or r['gc_during_run'] or r['error_flags']:
        print(json.dumps(r))
print('all walls:', [r['wall_ms'] for r in runs])
