"""World_size 2 over gloo: the env-batch sharding used by bench.py --gpus N (one process per GPU, no step-path collective, one
metrics reduction per episode).  On a GPU box the two ranks step the PRODUCT (HIP env objects, both on device 0) and the result is
compared with the unsharded product run; without a GPU the oracle stands in for the device kernels (test infrastructure)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rollout(cfg, seeds, steps):
    from oracle import oracle
    ref = oracle.WildfireOracle(cfg)
    ref.reset()
    for t in range(steps):
        actions = oracle.wildfire_random_policy(cfg, ref.agent_task_count, ref.env_task_count, seeds, 42, t)
        field, agent = oracle.wildfire_philox_randomness(cfg, seeds, ref.num_moves)
        ref.step(actions, field, agent)
    return ref


class _Shard:
    pass


def _rollout_product(seeds, steps):
    """The same rollout on the HIP env: device random policy (its stream depends on the env seed only), Philox randomness."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import configs
    from free_range_zoo_amd.envs import wildfire_v0
    B = len(seeds)
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda', 0), rng='philox')
    env.reset(seed=torch.from_numpy(seeds))
    for t in range(steps):
        env.step_random_policy(policy_seed=42, policy_step=t)
    env.check()
    out = _Shard()
    out.fires = env.state().fires.reshape(B, -1).cpu().numpy()
    out.cumulative_rewards = env._cumulative.cpu().numpy()
    out.terminations, out.truncations = env._terminations.cpu().numpy(), env._truncations.cpu().numpy()
    return out


def _worker(rank, world, port, per_rank, steps, out_queue, product=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import configs
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    from free_range_zoo_amd.utils import sharding
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    seeds = sharding.shard_seeds(rank, per_rank).numpy()
    ref = _rollout_product(seeds, steps) if product else _rollout(to_cstruct(configs.wildfire_openness(), per_rank, 50), seeds, steps)
    finished = torch.from_numpy((ref.terminations.all(axis=0) | ref.truncations.all(axis=0)))
    metrics = sharding.episode_metrics(torch.from_numpy(ref.cumulative_rewards.copy()), finished, per_rank * steps)
    local = metrics.clone()
    sharding.reduce_metrics(metrics)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    fires = [torch.zeros((per_rank, 6), dtype=torch.int32) for _ in range(world)]
    dist.all_gather(fires, torch.from_numpy(ref.fires.copy()))
    if rank == 0:
        out_queue.put((metrics.numpy(), torch.stack(gathered).numpy(), torch.cat(fires).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_rollout_equals_unsharded():
    sys.path.insert(0, ROOT)
    import configs
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    from free_range_zoo_amd.utils import sharding
    world, per_rank, steps = 2, 300, 12
    ctx = mp.get_context('spawn')
    queue = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, steps, queue)) for r in range(world)]
    for p in procs:
        p.start()
    reduced, per_rank_metrics, fires = queue.get()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the single collective: a sum of the per-rank metric vectors
    np.testing.assert_allclose(reduced, per_rank_metrics.sum(axis=0), rtol=1e-12)
    assert reduced[3] == world * per_rank * steps
    # sharded == unsharded, env for env (seeds are the global env index; nothing crosses ranks on the step path)
    whole = _rollout(to_cstruct(configs.wildfire_openness(), world * per_rank, 50), np.arange(world * per_rank, dtype=np.int32), steps)
    assert np.array_equal(fires, whole.fires)
    np.testing.assert_allclose(reduced[:3], whole.cumulative_rewards.sum(axis=1, dtype=np.float64), rtol=1e-9)
    assert sharding.shard_range(1, per_rank) == (per_rank, 2 * per_rank)


@pytest.mark.gpu
def test_two_rank_sharded_product_rollout_equals_unsharded():
    """The product itself under two ranks (gloo, both on the box's one GPU): rank r steps the envs [r * n, (r + 1) * n) with seeds = global
    index; gathered, they are the unsharded run env for env, and the reduced metrics are its sums."""
    from free_range_zoo_amd.utils import sharding
    world, per_rank, steps = 2, 3000, 12
    ctx = mp.get_context('spawn')
    queue = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, steps, queue, True)) for r in range(world)]
    for p in procs:
        p.start()
    reduced, per_rank_metrics, fires = queue.get()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    np.testing.assert_allclose(reduced, per_rank_metrics.sum(axis=0), rtol=1e-12)
    whole = _rollout_product(np.arange(world * per_rank, dtype=np.int32), steps)
    assert np.array_equal(fires, whole.fires)
    np.testing.assert_allclose(reduced[:3], whole.cumulative_rewards.sum(axis=1, dtype=np.float64), rtol=1e-9)
    assert reduced[3] == world * per_rank * steps
