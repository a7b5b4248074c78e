import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
import golden_util as G
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import rideshare_v0
name = 'busy_waiting_costs'
data = np.load(G.golden_path(f'traj_rideshare_{name}.npz'))
cfg = G.load_cfg(data, _capi.frz_rideshare_cfg)
B, A = cfg.parallel_envs, cfg.num_agents
env = rideshare_v0.parallel_env(configuration=configs.RIDESHARE_GOLDEN[name](), parallel_envs=B, max_steps=cfg.max_steps, device=torch.device('cuda'))
env.reset(seed=torch.arange(B, dtype=torch.int32))
print([k for k in data.files if k.startswith('s0_')][:40])
for t in range(2):
    p = f's{t}_'
    acts = data[p + 'actions']
    print('step', t, 'actions env1', acts[:, 1].tolist())
    print(' before: agents', env.state().agents[1].tolist())
    tab = env.state().passengers
    print(' table env1', tab[tab[:, 0] == 1].tolist())
    for a in range(A):
        m = env.agent_action_mapping[env.agents[a]]
        print('  map', a, m.values()[m.offsets()[1]:m.offsets()[2]].tolist())
    env.step({agent: torch.from_numpy(acts[a]).cuda() for a, agent in enumerate(env.agents)})
    print(' after: agents', env.state().agents[1].tolist(), 'want', data[p + 'agents'][1].tolist())
