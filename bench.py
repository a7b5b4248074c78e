#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of wildfire_v0 random-policy rollouts on MI355X (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch: the device-side random policy writes the actions, then the fused
HIP kernel performs one ParallelEnv.step() for all parallel_envs (action decode, 7 transitions, rewards/termination,
open action/observation space rebuild).  Episodes are max_steps = 50 long (BASELINE.md protocol); the reset between
episodes is inside the timed region.  Inputs are resident in HBM; nothing crosses PCIe inside the timed region.
Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

EPISODE = 50          # max_steps of the benchmark protocol
BATCH_PER_GPU = 65536  # BASELINE.json configs[1]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes_per_env_step(HW, A, k, mean_tasks, mean_agent_tasks_sum, injected_randomness=False):
    """reference-visible dtypes, every array read or written once per env-step."""
    state = 2 * (3 * HW * 4 + 3 * A * 4)
    io = 8 * A + 4 * A + 2 * A + 24
    obs = 16 * A + 4 * k * A * (A - 1)
    tasks = 32 * mean_tasks + 8
    obs_map = 8 * mean_tasks + 8
    act_maps = 8 * mean_agent_tasks_sum + 8 * A
    counts = 8 + 4 * A
    rnd = 4 * (3 * HW + 5 * A) if injected_randomness else 0
    return state + io + obs + tasks + obs_map + act_maps + counts + rnd


def cpu_baseline(budget_s=12.0, cores=None):
    """The CPU oracle (scalar C restatement) on a bounded sample of the same workload: same policy, same Philox randomness,
    same step — reported beside the GPU number, never the thing measured above.  The env-batch axis shards on the host
    exactly as it does across GPUs: `cores` threads each step their own shard of envs with their own oracle instance
    (ctypes releases the GIL inside the C calls; no process is forked or exec'd after the GPU was initialised)."""
    import configs
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    try:
        available = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        available = os.cpu_count() or 1
    cores = max(1, min(cores or 16, available))
    shard = 16384
    oracle.lib()

    def worker(index):
        cfg = to_cstruct(configs.wildfire_openness(), shard, EPISODE)
        ref = oracle.WildfireOracle(cfg)
        seeds = np.arange(shard, dtype=np.int32) + index * shard
        steps, t_busy = 0, 0.0
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < budget_s:
            ref.reset()
            t0 = time.perf_counter()
            for t in range(EPISODE):
                actions = oracle.wildfire_random_policy(cfg, ref.agent_task_count, ref.env_task_count, seeds, 7, steps + t)
                field, agent = oracle.wildfire_philox_randomness(cfg, seeds, ref.num_moves)
                ref.step(actions, field, agent)
            t_busy += time.perf_counter() - t0
            steps += EPISODE
        return steps, t_busy

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        results = list(pool.map(worker, range(cores)))
    wall = time.perf_counter() - t0
    total_steps = sum(r[0] for r in results)
    return {
        'value': shard * total_steps / wall,
        'unit': 'env-steps/s',
        'cores': cores,
        'kind': 'port',
        'sample': f'oracle (scalar C restatement): wildfire cfg2, {cores} threads x {shard} envs each, {total_steps // EPISODE} episodes x {EPISODE} '
                  f'steps in total incl. policy + Philox randomness, {wall:.1f} s wall on {cores} of {os.cpu_count()} host cores',
    }


def secondary_workloads(device, B):
    """BASELINE.json configs 2 and 3 (parity-test cases, not the bench line): random-policy rollouts of the other two domains at the same
    batch, whole episodes replayed as HIP graphs, reported beside the headline (never part of `value`)."""
    import configs
    from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0
    out = {}
    for name, module, configuration, reps in (('cybersecurity_v0 cfg4 (3 nodes, 2+2 agents, agent openness on)', cybersecurity_v0, configs.cyber_openness(), 20),
                                              ('rideshare_v0 cfg3 (10x10 grid, 8 agents, 2 passengers entering per step)', rideshare_v0,
                                               configs.rideshare_busy(), 5)):
        env = module.parallel_env(configuration=configuration, parallel_envs=B, max_steps=EPISODE, device=device, rng='philox', exact_shapes=False)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        graph = env.capture_random_rollout(EPISODE, policy_seed=20260104, include_reset=True)
        graph.replay()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            graph.replay()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        env.check()
        out[name] = {'env_steps_per_s': B * EPISODE * reps / elapsed, 'ms_per_step': 1e3 * elapsed / (EPISODE * reps), 'parallel_envs': B,
                     'steps': EPISODE * reps}
        del graph, env
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--rng', choices=['philox', 'mt19937'], default='philox')
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='parallel_envs per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the cybersecurity / rideshare rollouts reported beside the headline')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    # rehearsal on a one-GPU box only (never set by the driver): FRZ_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and swaps RCCL
    # (which refuses two ranks on one device) for gloo, so that the N > 1 code path — seeds per rank, barriers, the max over ranks, the
    # metrics reduction, rank 0's JSON line — can be run end to end without an 8-GPU node
    rehearsal = os.environ.get('FRZ_BENCH_SHARE_DEVICE') == '1'
    device_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    device = torch.device('cuda', device_index)
    dist = None
    if world > 1 or 'RANK' in os.environ:  # launched by torch.distributed.run (also with one rank: same code path as N > 1)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=device)  # RCCL over xGMI

    import configs
    from free_range_zoo_amd.envs import wildfire_v0
    B = args.batch
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=EPISODE, device=device,
                                   rng=args.rng, exact_shapes=False)
    A, HW = len(env.agents), env.max_y * env.max_x
    from free_range_zoo_amd.utils import sharding
    base_seed = sharding.shard_seeds(rank, B)  # seed = global env index = rank * B + i
    metrics = torch.zeros(A + 2, dtype=torch.float64, device=device)  # (sum reward per agent, env-steps, finished)
    actions = torch.zeros((A, B, 2), dtype=torch.int32, device=device)

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    state = {'step': 0, 'episode': 0}

    def episode_metrics():
        # episode end: the rank's own metrics are accumulated on the device; nothing crosses GPUs on the step path.  The
        # job's single collective (one RCCL all-reduce of A + 2 doubles over xGMI) runs once, at the end of the timed region.
        env.accumulate_episode_metrics(metrics)  # one launch: cumulative rewards per agent, env-steps, finished envs

    def one_step():
        """Eager path: the same launches the graphs replay, issued one by one through the Python boundary."""
        if state['step'] % EPISODE == 0:
            if state['step'] > 0:
                episode_metrics()
            env.reset(seed=base_seed + 1000003 * state['episode'])
            state['episode'] += 1
        env.step_random_policy(policy_seed=20260104 + rank, policy_step=state['step'] % EPISODE)
        state['step'] += 1

    # ---- timed region: K steps as HIP-graph replays of whole episodes (reset + 50 x (policy, step) per replay)
    env.reset(seed=base_seed)
    full = env.capture_random_rollout(EPISODE, policy_seed=20260104 + rank, include_reset=True)
    rem_steps = args.steps % EPISODE
    rem = env.capture_random_rollout(rem_steps, policy_seed=20260104 + rank, include_reset=True) if rem_steps else None
    seed_stride = torch.tensor(1000003, dtype=torch.int32, device=device)

    def run(steps):
        done = 0
        while done < steps:
            env.seeds.add_(seed_stride)  # fresh env seeds for every episode
            if steps - done >= EPISODE:
                full.replay()
                done += EPISODE
            else:
                rem.replay()
                done += steps - done
            episode_metrics()

    run(max(EPISODE, (args.warmup // EPISODE) * EPISODE))
    barrier()
    t0 = time.perf_counter()
    metrics.zero_()
    run(args.steps)
    sharding.reduce_metrics(metrics)  # inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * args.steps / elapsed

    # ---- the drop-in Python API path (env.step per call, host-bound), reported beside the headline
    api_steps = min(args.steps, 200)
    for _ in range(20):
        one_step()
    barrier()
    t1 = time.perf_counter()
    for _ in range(api_steps):
        one_step()
    barrier()
    api_value = world * B * api_steps / (time.perf_counter() - t1)

    # ---- kernel-level pass (not part of `value`): two HIP events on the launch stream take the step dispatch's own begin
    # and end timestamps (hipExtLaunchKernel start/stop events: what rocprofv3's kernel trace reports, profiles/) + mean
    # task counts for the algorithmic bytes
    from free_range_zoo_amd import _capi
    from free_range_zoo_amd.utils.env import stream_ptr
    lib, handle = env._lib, env._handle
    mode = _capi.FRZ_RNG_MT19937 if args.rng == 'mt19937' else _capi.FRZ_RNG_PHILOX
    n_probe = min(args.steps, 200)
    kernel_ms = []
    task_sum = torch.zeros(1 + A, dtype=torch.float64, device=device)
    env.reset(seed=base_seed + 17)
    stream = stream_ptr(device)
    torch.cuda.synchronize(device)
    done = 0
    while done < n_probe:  # one episode at a time: reset, then the episode's steps back to back, each with its own event pair
        n = min(EPISODE, n_probe - done)
        if done > 0:
            env.seeds.add_(seed_stride)
            if args.rng == 'mt19937':
                env.generator._seed_streams(None)
            lib.frz_wildfire_reset(handle, stream)
        out = (ctypes.c_float * n)()
        _capi.check(lib.frz_wildfire_timed_rollout(handle, 20260104 + rank, 0, n, env._actions.data_ptr(), mode, stream, out), 'frz_wildfire_timed_rollout')
        kernel_ms.extend(out[i] for i in range(n))
        done += n
    # mean task counts for the algorithmic bytes: the same episodes again, step by step (not timed)
    env.reset(seed=base_seed + 17)
    for i in range(n_probe):
        if i % EPISODE == 0 and i > 0:
            env.seeds.add_(seed_stride)
            if args.rng == 'mt19937':
                env.generator._seed_streams(None)
            lib.frz_wildfire_reset(handle, stream)
        lib.frz_wildfire_step_random_policy(handle, 20260104 + rank, i % EPISODE, env._actions.data_ptr(), mode, None, None, stream)
        task_sum[0] += env.environment_task_count.sum()
        task_sum[1:] += env.agent_task_count.sum(dim=1)
    torch.cuda.synchronize(device)
    kernel_ms_avg = float(np.mean(kernel_ms))
    kernel_ms_med = float(np.median(kernel_ms))
    mean_tasks = float(task_sum[0].item()) / (n_probe * B)
    mean_agent_tasks = float(task_sum[1:].sum().item()) / (n_probe * B)
    per_env = algorithmic_bytes_per_env_step(HW, A, env._k, mean_tasks, mean_agent_tasks, injected_randomness=(args.rng == 'mt19937'))
    achieved = per_env * B / (kernel_ms_avg * 1e-3) / 1e9
    traffic = None
    traffic_file = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if os.path.exists(traffic_file):
        try:
            traffic = json.load(open(traffic_file)).get('wf_step_kernel_bytes_per_launch')
        except Exception:  # noqa: BLE001
            traffic = None

    if rank == 0:
        env.check()
        line = {
            'metric': 'env steps/sec (batch x agents) wildfire random-policy',
            'value': value,
            'unit': 'env-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'i32/f32',
            'data': 'synthetic',
            'config': {'workload': f'wildfire_v0 cfg2 (2x3 grid, 3 agents, agent+task openness on), batch={B} per GPU, '
                                   f'max_steps={EPISODE}, uniform random policy sampled inside the step launch, rng={args.rng}, reset inside timed region, one HIP graph replay per episode',
                       'parallel_envs_per_gpu': B, 'agents': A, 'sharding': f'env-batch axis x{world}, no step-path collective, one metrics all-reduce per run'},
            'agent_steps_per_s': value * A,
            'python_api_env_steps_per_s': api_value,
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'kernel': 'wf_roles_kernel<6,3,exact,philox,step>' if args.rng == 'philox' else 'wf_roles_kernel<6,3,exact,injected,step>',
                         'kernel_ms_avg': kernel_ms_avg,
                         'kernel_ms_median': kernel_ms_med, 'algorithmic_bytes_per_env_step': per_env,
                         'mean_tasks_per_env': mean_tasks, 'mean_agent_tasks_per_env': mean_agent_tasks},
            'reference_cpu_env_steps_per_s': {'value': 21112, 'source': 'BASELINE.md §2: unmodified reference, 8 vCPU, B=65536 (survey container)'},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        if world == 1 and not args.no_secondary:
            line['secondary_workloads'] = secondary_workloads(device, B)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
