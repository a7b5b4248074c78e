"""Re-run one case of tests/test_hip_fuzz.py::test_rideshare_on_random_configurations and print where the HIP path and the oracle part:
python tools/dbg/fuzz_case.py <index>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_hip_fuzz as F, test_hip_rideshare as T
from oracle import oracle
from free_range_zoo_amd.envs.rideshare.env.structures.configuration import to_cstruct
case = F.rideshare_case(int(sys.argv[1]))
print(case)
B, max_steps, steps, seed, contest = case['B'], case['steps'] + 2, case['steps'] + 4, case['seed'], case['contest']
build = lambda: F.build_rideshare(case)
cfg, schedule = to_cstruct(build(), B, max_steps)
o = oracle.RideshareOracle(cfg, schedule); o.reset()
env = T.make_env(build, B, max_steps); env.reset(seed=torch.arange(B, dtype=torch.int32))
gen = np.random.default_rng(seed); A = cfg.num_agents
for t in range(steps):
    actions = o.random_policy(5 + seed, t)
    for a in range(A):
        off = o.agent_offsets[a]
        for b in np.nonzero(gen.random(B) < contest)[0]:
            states = o.agent_task_states[a, off[b]:off[b + 1]]
            free = np.nonzero(states == 0)[0]
            if free.size:
                actions[a, b] = (free[0], 0)
    before = T.np_(env._agents).copy()
    env.step(torch.from_numpy(actions).cuda()); o.step(actions)
    got, want = T.np_(env._rewards), np.asarray(o.rewards)
    bad = np.argwhere(got != want)
    if bad.size:
        for a, b in bad[:4]:
            after = T.np_(env._agents)
            print(f'step {t} agent {a} env {b}: got {got[a, b]!r} want {want[a, b]!r} diff {got[a, b] - want[a, b]:.3e} action {actions[a, b]} pos {before[b, a]} -> {after[b, a]} '
                  f'move {after[b, a] - before[b, a]}')
        break
