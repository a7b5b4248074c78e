"""Cybersecurity networks of 9-16 nodes (<= 8 agents): us per step of a 50-step random-policy rollout at B = 65536 — one launch per step with the lane-per-env kernel
(FRZ_CY_KERNEL=lane: what these shapes ran before round 4), one launch per step with the state / view kernel, and ONE multi-step launch.
usage: python tools/dbg/cy_nodes_probe.py [N Att D]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch, configs
    from free_range_zoo_amd import _capi
    from free_range_zoo_amd.envs import cybersecurity_v0
    N, Att, D, exclusive = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == '1'
    B, steps = 65536, 50
    env = cybersecurity_v0.parallel_env(configuration=configs.cyber_grid(N, Att, D), parallel_envs=B, max_steps=steps, device=torch.device('cuda'), rng='philox', exact_shapes=False)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    if exclusive:
        env.set_exclusive_device(True)
    launches = env._lib.frz_cybersecurity_rollout_launches(env._handle, steps, _capi.FRZ_RNG_PHILOX)
    ts = []
    for rep in range(4):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + rep); torch.cuda.synchronize()
        t0 = time.perf_counter(); env.rollout(steps, policy_seed=3); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6 / steps)
    env.check()
    print(f'{N} nodes, {Att}+{D} agents, FRZ_CY_KERNEL={os.environ.get("FRZ_CY_KERNEL", "default")}, launches {launches}: {min(ts[1:]):.1f} us per step', flush=True)
else:
    shape = sys.argv[1:4] if len(sys.argv) >= 4 else ['12', '3', '3']
    for family, exclusive in (('lane', '0'), ('', '0'), ('', '1')):
        env = dict(os.environ)
        if family:
            env['FRZ_CY_KERNEL'] = family
        subprocess.run([sys.executable, __file__, '--child', *shape, exclusive], env=env)
