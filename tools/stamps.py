"""Where does ONE workgroup of the wildfire step kernel spend its cycles?  Runs the diagnostic build
(free-range-zoo_amd/csrc/libfrz_hip_stamps.so: s_memtime stamps at phase boundaries, written by thread 0 of workgroup 0
into a scratch region of the arena) and prints per-phase shares.  Shares, not absolute run time (stamps perturb)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['FRZ_HIP_LIB'] = os.path.join(ROOT, 'free-range-zoo_amd', 'csrc', 'libfrz_hip_stamps.so')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.utils.env import stream_ptr
lane_names = ['loads+config/LDS init', 'epoch+totals+frozen test', 'loop top', 'randomness (philox)', 'decode', 'agent transitions+agent/obs stores',
              'fire inc/dec+spread+cell stores', 'rebuild masks+scan', 'publish+rewards+reward stores+prefetch', 'look-back', 'jagged stores']
role_points = ['kernel entry (config requested)', 'config staged', 'epoch/totals read', 'phase 1 done', 'past barrier 1', 'phase 2 done', 'past barrier 2', 'phase 3 done',
               'phase 4 done', 'past barrier 5', 'phase 6 done']
roles = os.environ.get('FRZ_WF_KERNEL', 'roles') != 'lane'
for B in [int(x) for x in sys.argv[1:]] or [256, 65536]:
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'),
                                   rng='philox', exact_shapes=False)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    bufs = env._bufs
    lib, h, s = env._lib, env._handle, stream_ptr(env.device)
    rows = []
    for t in range(30):
        lib.frz_wildfire_random_policy(h, 1, t, env._actions.data_ptr(), s)
        lib.frz_wildfire_step(h, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, None, None, s)
        torch.cuda.synchronize()
        # the stamps live at the start of the (unused in philox mode) agent-randomness staging region, which precedes mt_state
        off = bufs.mt_state - env._arena.data_ptr() - ((5 * B * 3 * 4 + 255) // 256) * 256
        st = env._arena[off:off + 32 * 8].view(torch.int64).cpu().numpy().astype(np.int64)
        rows.append(st.copy())
        if roles and t == 29 and B >= 65536:
            offf = off - ((3 * B * 6 * 4 + 255) // 256) * 256  # field-randomness staging region precedes the agent one
            n = (B + 255) // 256
            wall = env._arena[offf:offf + n * 32].view(torch.int64).cpu().numpy().astype(np.int64).reshape(n, 4)
            t00 = wall[:, [0, 2]].min()
            start = wall[:, [0, 2]].min(axis=1) - t00
            end = wall[:, [1, 3]].max(axis=1) - t00
            q = lambda v: ' '.join(f'{x * 10:7.0f}' for x in np.percentile(v, [0, 10, 50, 90, 100]))
            print(f'--- B={B}: per-workgroup wall clock, ns since the first workgroup started (min p10 p50 p90 max)')
            print(f'  start {q(start)}')
            print(f'  end   {q(end)}')
            print(f'  life  {q(end - start)}')
    st = np.array(rows[5:])
    if roles:
        t0 = st[:, 0:1]
        field = np.median(st[:, 0:11] - t0, axis=0)
        crew = np.median(st[:, 16:27] - t0, axis=0)
        print(f'--- B={B} field/crew kernel: shader cycles since the field role entered the kernel (workgroup 0)')
        print(f'  {"point":22s} {"field":>8s} {"crew":>8s}')
        for n, a, c in zip(role_points, field, crew):
            print(f'  {n:22s} {int(a):8d} {int(c):8d}')
        extra = ['state loads issued', 'config piece arrived (LDS written)', 'past staging barrier']  # stamps 11..13
        for k, n in enumerate(extra):
            print(f'  staging: {n:36s} {int(np.median(st[:, 11 + k] - t0[:, 0])):8d} {int(np.median(st[:, 27 + k] - t0[:, 0])):8d}')
    else:
        dd = np.median(np.diff(st[:, :len(lane_names)], axis=1), axis=0)
        total = dd.sum()
        print(f'--- B={B} lane kernel: {total} shader cycles between first and last stamp (~{total / 2.4e3:.1f} us at 2.4 GHz)')
        for n, v in zip(lane_names[1:], dd):
            print(f'  {n:40s} {int(v):7d}  {100 * v / total:5.1f} %')
    del env
