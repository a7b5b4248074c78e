"""Grid-family rollouts with and without the lists launch overlapped with the next env launch (FRZ_WG_OVERLAP=0 switches it off): a 50-step
`frz_wildfire_rollout` per episode, by events around it, eager and through a captured graph.  usage: python tools/dbg/grid_overlap_ab.py [8x8x12 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd.envs import wildfire_v0
B = int(os.environ.get('FRZ_PROBE_B', 65536))
for shape in sys.argv[1:] or ['8x8x12', '16x16x6']:
    H, W, A = (int(v) for v in shape.split('x'))
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_grid(H, W, A), parallel_envs=B, max_steps=50, device=torch.device('cuda', 0),
                                   rng='philox', exact_shapes=False)
    seeds = torch.arange(B, dtype=torch.int32, device='cuda')
    env.reset(seed=seeds)
    eager, graphed = [], []
    graph = env.capture_random_rollout(50, policy_seed=1, include_reset=False)
    for rep in range(4):
        for out, run in ((eager, lambda: env.rollout(50, policy_seed=1)), (graphed, graph.replay)):
            env.reset(seed=seeds)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / 50)
    env.check()
    print(f'{shape}: overlap={os.environ.get("FRZ_WG_OVERLAP", "1")} us per step: eager rollout {np.median(eager[1:]):.1f}  graph replay {np.median(graphed[1:]):.1f}', flush=True)
    del env, graph
