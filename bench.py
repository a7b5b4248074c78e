#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of wildfire_v0 random-policy rollouts on MI355X (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch: the fused HIP kernel samples the uniform random policy and performs one
ParallelEnv.step() for all parallel_envs (action decode, 7 transitions, rewards / termination, open action / observation space
rebuild, every per-step output written).  An episode's steps are ONE launch where the library has a multi-step launch for the shape
(the bench shape has: the workgroups keep their envs in registers from step to step), otherwise one launch per step.  Inputs are
resident in HBM; nothing crosses PCIe inside the timed region.

Protocol (timed rollouts, median reported).  The K steps are one BLOCK = a rollout loop captured as ONE HIP
graph: per episode of max_steps = 50 (the last one shorter when K is not a multiple) fresh env seeds, reset, the episode's
steps, the episode-metrics reduction.  W warm-up steps are run as whole blocks (graph uploaded and warm before any clock starts).
Then the block is timed R >= 10 times, each time bracketed by barrier + torch.cuda.synchronize() on both sides; a timed block
holds NO collective (N > 1: every rank accumulates its metrics on its own device and the job's one RCCL all-reduce runs after the
timed loop; what an all-reduce per block would cost is measured separately and reported under `distributed`); the per-block times are max-reduced over the ranks and the MEDIAN block is reported: value = N * B * K / median, ms_per_step =
median / K.  Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
# Kernel arguments of eagerly launched kernels in device memory instead of host memory (read by the HIP runtime when it loads, i.e.
# before `import torch`): the step kernel's first instructions wait for them, 10.8 -> 9.9 us per eager launch.  Graph launches — the
# timed region — keep their arguments on the device anyway (no difference measured); this makes the eagerly launched probe that
# times single dispatches for `roofline` see the kernel the graph runs.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

import numpy as np  # noqa: E402
import torch  # noqa: E402

EPISODE = 50          # max_steps of the benchmark protocol
BATCH_PER_GPU = 65536  # BASELINE.json configs[1]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PROBE_EPISODES = 4     # whole episodes behind the kernel-level figures (independent of --steps)
MIN_BLOCKS, TARGET_TIMED_S = 10, 0.05


def wildfire_bytes_per_env_step(HW, A, k, mean_tasks, mean_agent_tasks_sum, injected_randomness=False, state_passes=2.0):
    """reference-visible dtypes, every array read or written once per env-step.  state_passes = 2: the state read and
    written every step (the formula as the survey states it); 1 + 1/n: a launch of n steps reads the state once and writes it every step."""
    state = state_passes * (3 * HW * 4 + 3 * A * 4)
    io = 8 * A + 4 * A + 2 * A + 24
    obs = 16 * A + 4 * k * A * (A - 1)
    tasks = 32 * mean_tasks + 8
    obs_map = 8 * mean_tasks + 8
    act_maps = 8 * mean_agent_tasks_sum + 8 * A
    counts = 8 + 4 * A
    rnd = 4 * (3 * HW + 5 * A) if injected_randomness else 0
    return state + io + obs + tasks + obs_map + act_maps + counts + rnd


def cybersecurity_bytes_per_env_step(N, Att, D, mean_presence_sum):
    """2(4N + 4D + A) + 8A + 4A + ~10A + 16NA + 2 x sum_a (4 N presence_a + 8); the action maps hold N entries
    per PRESENT agent (measured mean presence), the observation map N per env."""
    A = Att + D
    maps = 4 * N * mean_presence_sum + 8 * A + 4 * N + 8
    return 2 * (4 * N + 4 * D + A) + 8 * A + 4 * A + 10 * A + 16 * N * A + maps


def rideshare_bytes_per_env_step(A, mean_passengers, mean_visible_sum):
    """2(8A + 44P) + 8A + 4A + 16A + 16A(A-1) + sum_a (32 P_a + 8) + 2 sum_a (8 P_a + 8), with the measured mean
    live passengers P and visible tasks P_a."""
    P = mean_passengers
    return 2 * (8 * A + 44 * P) + 8 * A + 4 * A + 16 * A + 16 * A * (A - 1) + 32 * mean_visible_sum + 8 * A + 2 * (8 * mean_visible_sum + 8 * A)


def source_fingerprint():
    """sha256 over the kernel sources: profiles/hbm_traffic.json records the fingerprint it was measured on."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'free-range-zoo_amd', 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.h', '.inl')):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), 'rb').read())
    return h.hexdigest()[:16]


def recorded_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/hbm_traffic.json): PMC counters cannot be collected from
    inside this process, so the figure is a RECORDED one and is only reported for the kernel build it was measured on."""
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    try:
        rec = json.load(open(path))
    except Exception:  # noqa: BLE001
        return None, None
    if rec.get('source_fingerprint') != source_fingerprint():
        return None, f'profiles/hbm_traffic.json was recorded on another kernel build ({rec.get("source_fingerprint")}); not reported'
    return rec.get(key), f'recorded: {rec.get("how", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes")} (profiles/hbm_traffic.json)'


def granted_cores():
    """CPU cores this process may really use: the scheduler affinity, cut down to the cgroup's CFS quota where one is set (the one-GPU box
    shows all 256 host threads and grants 16 cores' worth of time per 100 ms period)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        cores = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', ):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                cores = max(1, min(cores, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(cores, budget_s=8.0):
    """The CPU oracle (scalar C restatement) on a bounded sample of the same workload: same policy, same Philox randomness, same
    step — reported beside the GPU number, never the thing measured above.  The env-batch axis shards on the host exactly as it
    does across GPUs: `cores` threads each step their own shard of envs with their own oracle instance, one C call per episode
    (ctypes releases the interpreter lock for its duration; no process is forked or exec'd after the GPU was initialised)."""
    import configs
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    shard = 4096
    oracle.lib()

    def worker(index):
        cfg = to_cstruct(configs.wildfire_openness(), shard, EPISODE)
        ref = oracle.WildfireOracle(cfg)
        seeds = np.arange(shard, dtype=np.int32) + index * shard
        episodes = 0
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < budget_s:
            ref.reset()
            ref.rollout(seeds + 1000003 * episodes, 20260104, 0, EPISODE)
            episodes += 1
        return episodes

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        episodes = sum(pool.map(worker, range(cores)))
    wall = time.perf_counter() - t0
    return {
        'value': shard * episodes * EPISODE / wall,
        'unit': 'env-steps/s',
        'cores': cores,
        'kind': 'port',
        'sample': f'oracle (scalar C restatement): wildfire cfg2, {cores} threads x {shard} envs each, {episodes} episodes x {EPISODE} steps in '
                  f'total incl. reset, policy and Philox randomness, {wall:.1f} s wall on {cores} of {os.cpu_count()} host cores',
    }


def episode_ms(env, steps, policy_seed, reps, device):
    """Device time of `steps` random-policy steps (no reset inside), by HIP events on the launch stream around a graph replay."""
    graph = env.capture_random_rollout(steps, policy_seed=policy_seed, include_reset=False)
    times = []
    for _ in range(reps):
        env.reset(seed=torch.arange(env.parallel_envs, dtype=torch.int32))
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize(device)
        times.append(e0.elapsed_time(e1))
    del graph
    return float(np.median(times))


def secondary_workloads(device, B):
    """BASELINE.json configs 2 and 3 (parity-test cases, not the bench line): random-policy rollouts of the other two domains at the same
    batch, reported beside the headline (never part of `value`), each with its own roofline record: whole episodes replayed as HIP
    graphs for the rate, HIP events around an episode's steps for the step time, task counts averaged over the same episode."""
    import configs
    from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0, wildfire_v0
    out = {}
    specs = (('cybersecurity_v0 cfg4 (3 nodes, 2+2 agents, agent openness on)', cybersecurity_v0, configs.cyber_openness(), 20),
             ('rideshare_v0 cfg3 (10x10 grid, 8 agents, 2 passengers entering per step)', rideshare_v0, configs.rideshare_busy(), 5),
             # the wildfire scaling variants of SURVEY 8(d): grids beyond the bench shape run one env per wavefront (csrc/wildfire_grid.hip)
             ('wildfire_v0 8x8 grid, 12 agents (every stochastic switch on)', wildfire_v0, configs.wildfire_grid(8, 8, 12), 5),
             ('wildfire_v0 16x16 grid, 6 agents (every stochastic switch on)', wildfire_v0, configs.wildfire_grid(16, 16, 6), 5))
    for name, module, configuration, reps in specs:
        env = module.parallel_env(configuration=configuration, parallel_envs=B, max_steps=EPISODE, device=device, rng='philox', exact_shapes=False)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        if hasattr(env, 'set_exclusive_device'):
            env.set_exclusive_device(True)  # N = 1, one process, one stream: multi-step launches where the library has them
        graph = env.capture_random_rollout(EPISODE, policy_seed=20260104, include_reset=True)
        graph.replay()
        torch.cuda.synchronize(device)
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize(device)
            times.append(time.perf_counter() - t0)
        elapsed = float(np.median(times))
        env.check()
        del graph
        step_ms = episode_ms(env, EPISODE, 20260104, 3, device) / EPISODE
        # mean task counts over one episode (not timed)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        A = len(env.agents)
        sums = torch.zeros(2, dtype=torch.float64, device=device)
        for t in range(EPISODE):
            env.step_random_policy(policy_seed=20260104, policy_step=t)
            sums[0] += env.environment_task_count.sum()
            sums[1] += env.agent_task_count.sum()
        mean_env, mean_agents = (sums / (EPISODE * B)).tolist()
        if module is wildfire_v0:
            HW = env.max_y * env.max_x
            per_env = wildfire_bytes_per_env_step(HW, A, env._k, mean_env, mean_agents)
            kernels = 'wg_env_kernel (a wavefront per env for the cells, one crew wavefront per four envs for the agents; policy sampled in it) + wg_lists_kernel (offsets and lists, an output entry per lane)'
            counts = {'mean_tasks_per_env': mean_env, 'mean_agent_tasks_per_env': mean_agents, 'cells': HW}
        elif module is cybersecurity_v0:
            N = env.network_config.num_nodes
            Att, D = env.attacker_config.num_attackers, env.defender_config.num_defenders
            per_env = cybersecurity_bytes_per_env_step(N, Att, D, mean_agents / N)
            kernels = 'cy_roles_kernel (state + view roles, policy sampled in the launch; an episode is ONE multi-step launch)'
            counts = {'mean_present_agents_per_env': mean_agents / N}
        else:
            per_env = rideshare_bytes_per_env_step(A, mean_env, mean_agents)
            kernels = getattr(env, 'step_kernels', 'rs_step_kernel + rs_policy_kernel')
            counts = {'mean_passengers_per_env': mean_env, 'mean_visible_tasks_per_env_summed_over_agents': mean_agents}
        wall_step_ms = 1e3 * elapsed / EPISODE  # the clock of `ms_per_step`: whole episodes at the wall, per-episode reset included
        achieved = per_env * B / (wall_step_ms * 1e-3) / 1e9
        achieved_events = per_env * B / (step_ms * 1e-3) / 1e9
        traffic_key = {cybersecurity_v0: 'cybersecurity', rideshare_v0: 'rideshare'}.get(module, 'wildfire_grid_%dx%d' % (getattr(env, 'max_y', 0), getattr(env, 'max_x', 0)))
        traffic, traffic_source = recorded_traffic(traffic_key + '_bytes_per_step')
        del env
        # the loop a user of the reference writes (per agent action_space(agent).sample_nested(), step(dict), finished read once per episode), through
        # the drop-in API with its defaults; `exclusive`: env.set_exclusive_device() declared (counted steps where the library has a multi-step launch)
        loop_rates = {}
        for label, declare in (('default', False), ('exclusive_device', True)):
            loop_env = module.parallel_env(configuration=configuration, parallel_envs=B, max_steps=EPISODE, device=device, rng='philox')
            if declare and not (hasattr(loop_env, 'set_exclusive_device') and loop_env.set_exclusive_device(True) and loop_env._defer_chunk > 0):
                del loop_env
                continue
            seeds = torch.arange(B, dtype=torch.int32, device=device)

            def loop(episodes):
                for _ in range(episodes):
                    loop_env.reset(seed=seeds)
                    for _ in range(EPISODE):
                        loop_env.step({agent: loop_env.action_space(agent).sample_nested() for agent in loop_env.agents})
                    if not torch.all(loop_env.finished):
                        raise RuntimeError('the episode did not end at its horizon')

            loop(1)
            runs = []
            for _ in range(3):
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                loop(2)
                torch.cuda.synchronize(device)
                runs.append(time.perf_counter() - t0)
            loop_env.check()
            loop_rates[label] = {'env_steps_per_s': B * 2 * EPISODE / float(np.median(runs)), 'us_per_step': 1e6 * float(np.median(runs)) / (2 * EPISODE)}
            del loop_env
        out[name] = {'env_steps_per_s': B * EPISODE / elapsed, 'ms_per_step': 1e3 * elapsed / EPISODE, 'parallel_envs': B, 'steps': EPISODE,
                     'episodes_timed': reps, 'reference_loop': loop_rates,
                     'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                                  'traffic': traffic, 'traffic_source': traffic_source, 'kernel': kernels, 'algorithmic_bytes_per_env_step': per_env,
                                  'clock': 'frac / achieved use ms_per_step of this record (episodes at the wall, reset included: VERDICT r3 weak #5, #7)',
                                  'step_ms_avg': step_ms, 'frac_inside_the_episode': achieved_events / HBM_PEAK_GBS,
                                  'how_inside_the_episode': 'HIP events on the launch stream around one episode of step launches (graph replay, reset '
                                                            'outside), median of 3, divided by 50', **counts}}
    return out


def mt19937_workload(device, B, K, base_seed, policy_seed, seed_stride, repeats, barrier):
    """The headline block in the DEFAULT rng mode of the drop-in env (rng='mt19937': per-env MT19937 streams in HBM, bit-identical to the
    reference's CPU RandomGenerator for the same seeds, /root/reference utils/random_generator.py:49-146 — north_star's identical-seeds parity
    mode), N = 1 only, never part of `value`.  Same protocol as the headline: K-step blocks of <= 50-step episodes, reseed +
    reset + steps + metrics per episode, each block bracketed by barrier + synchronize, median block.  The block is ONE HIP graph here (the
    streams are re-seeded by a launch of their own, so an episode is not a single launch)."""
    import configs
    from free_range_zoo_amd import _capi
    from free_range_zoo_amd.envs import wildfire_v0
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=EPISODE, device=device, rng='mt19937',
                                   exact_shapes=False)
    A, HW = len(env.agents), env.max_y * env.max_x
    exclusive = env.set_exclusive_device(True)
    env.reset(seed=base_seed)
    metrics = torch.zeros(A + 2, dtype=torch.float64, device=device)
    block = env.capture_random_rollout(K, policy_seed=policy_seed, include_reset=True, episode_length=EPISODE, seed_stride=seed_stride, metrics=metrics)
    done = torch.cuda.Event()
    for _ in range(3):
        block.replay()
    torch.cuda.synchronize(device)
    block_s, event_ms = [], []
    for _ in range(max(MIN_BLOCKS, min(repeats, 200))):
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        block.replay()
        e1.record()
        done.record()
        while not done.query():
            pass
        torch.cuda.synchronize(device)
        block_s.append(time.perf_counter() - t0)
        event_ms.append(e0.elapsed_time(e1))
    del block
    median_s = float(np.median(block_s))
    # mean list sizes over the first min(K, 50) steps of an episode in this mode (not timed)
    n = min(K, EPISODE)
    sums = torch.zeros(2, dtype=torch.float64, device=device)
    env.reset(seed=base_seed + 17)
    for t in range(n):
        env.step_random_policy(policy_seed=policy_seed, policy_step=t)
        sums[0] += env.environment_task_count.sum()
        sums[1] += env.agent_task_count.sum()
    mean_tasks, mean_agent_tasks = (sums / (n * B)).tolist()
    env.check()
    draws = 3 * HW + 5 * A
    per_env = wildfire_bytes_per_env_step(HW, A, env._k, mean_tasks, mean_agent_tasks, injected_randomness=True)
    stream_bytes = 2 * 4 * draws + 4  # the env's generator words the step consumes, read and written back, + the stream position
    ms_step_events = float(np.median(event_ms)) / K
    out = {'value': B * K / median_s, 'unit': 'env-steps/s', 'ms_per_step': 1e3 * median_s / K, 'steps': K, 'rng': 'mt19937', 'multi_step_launches': bool(exclusive),
           'blocks': len(block_s), 'block_ms_median': 1e3 * median_s,
           'semantics': 'identical seeds => trajectories identical to the reference CPU RandomGenerator (tests/golden/mt19937_torch.npz, rng_modes.npz)',
           'roofline': {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'achieved': per_env * B / (median_s / K) / 1e9, 'frac': per_env * B / (median_s / K) / 1e9 / HBM_PEAK_GBS,
                        'clock': 'ms_per_step of this record (blocks at the wall)',
                        'frac_by_events': per_env * B / (ms_step_events * 1e-3) / 1e9 / HBM_PEAK_GBS, 'ms_per_step_by_events': ms_step_events,
                        'algorithmic_bytes_per_env_step': per_env,
                        'of_which_randomness': 4 * draws,  # the generator's output as the reference materialises it (field [3,B,H,W] + agent [5,B,A] float32)
                        'stream_words_bytes_per_env_step': stream_bytes,
                        'frac_with_stream_words': (per_env + stream_bytes) * B / (median_s / K) / 1e9 / HBM_PEAK_GBS,
                        'traffic': None, 'kernel': 'wf_roles_kernel<6,3,exact,mt19937,step,multi-step> + mt19937 re-seed launch per episode',
                        'mean_tasks_per_env': mean_tasks, 'mean_agent_tasks_per_env': mean_agent_tasks}}
    del env
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--rng', choices=['philox', 'mt19937'], default='philox')
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='parallel_envs per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the cybersecurity / rideshare rollouts reported beside the headline')
    ap.add_argument('--no-episode-probe', action='store_true', help='skip the whole-episode (50-step) launches timed beside the block\'s own launch and the '
                    'auto-reset workload: a kernel trace of the run then shows the block\'s launch in a row of its own')
    args = ap.parse_args()
    if args.steps <= 0:
        raise SystemExit('--steps must be positive')

    # torch's intra-op pool defaults to half the VISIBLE host threads (128 on the GPU box), the cgroup grants 16 cores per 100 ms period: one
    # CPU tensor op of B elements wakes 128 spinning workers, the period's quota is gone in ~12 ms and the kernel freezes the whole process
    # until the next period — round 3's unexplained 85-106 ms stalls in the per-step legs (tools/dbg/api_stall_probe.py)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), granted_cores())))
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
    from free_range_zoo_amd.utils import sharding
    B, K = args.batch, args.steps
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=EPISODE, device=device,
                                   rng=args.rng, exact_shapes=False)
    A, HW = len(env.agents), env.max_y * env.max_x
    # one process per GPU and one stream: the device is this env's alone, which is what a multi-step launch needs (its workgroups wait
    # for each other inside the kernel).  Not so in the shared-device rehearsal, where two ranks run on one GPU.
    exclusive = env.set_exclusive_device(not rehearsal)  # False: refused by the library's own residency check, or the rehearsal
    base_seed = sharding.shard_seeds(rank, B)  # seed = global env index = rank * B + i
    metrics = torch.zeros(A + 2, dtype=torch.float64, device=device)  # (sum reward per agent, env-steps, finished)
    job_metrics = torch.zeros_like(metrics)                           # what the collective reduces (copied at the end of the graph)
    policy_seed = 20260104 + rank

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---- the block: K steps as ONE graph = per episode (fresh seeds, reset, <= 50 x fused policy+step, episode metrics), metrics copy
    env.reset(seed=base_seed)
    seed_stride = 1000003
    from free_range_zoo_amd import _capi
    from free_range_zoo_amd.utils.env import stream_ptr
    fused_mode = _capi.FRZ_RNG_MT19937 if args.rng == 'mt19937' else _capi.FRZ_RNG_PHILOX
    one_launch_per_episode = env._lib.frz_wildfire_rollout_launches(env._handle, min(K, EPISODE), fused_mode) == 1 and args.rng == 'philox'
    if one_launch_per_episode:
        # an episode (opening reset with fresh seeds, its steps, its metrics) is ONE launch: the block's few launches are enqueued
        # directly through the C-ABI — a one-node graph costs more to launch than the launch it holds (169 vs 161 us per 20-step block)
        specs = []
        for first in range(0, K, EPISODE):
            spec = _capi.frz_rollout_spec()
            spec.n_steps, spec.rng_mode, spec.policy_seed, spec.first_step = min(EPISODE, K - first), fused_mode, policy_seed, 0
            spec.flags, spec.seed_increment = _capi.FRZ_ROLLOUT_RESET_FIRST, seed_stride
            spec.actions_out, spec.metrics = env._actions.data_ptr(), metrics.data_ptr()
            specs.append(ctypes.byref(spec))
            specs.append(spec)  # (keeps the struct alive)
        spec_refs = specs[0::2]
        rollout_entry, env_handle, launch_stream = env._lib.frz_wildfire_rollout, env._handle, stream_ptr(device)

        def enqueue_block():
            for ref in spec_refs:
                rollout_entry(env_handle, ref, launch_stream)
    else:
        block = env.capture_random_rollout(K, policy_seed=policy_seed, include_reset=True, episode_length=EPISODE, seed_stride=seed_stride,
                                           metrics=metrics)
        enqueue_block = block.replay
    done_event = torch.cuda.Event()

    def run_block(with_collective=False):
        # The timed region holds NO collective: every rank accumulates its episode metrics on its own device across
        # the blocks (the launches add to `metrics` in place) and the job's single collective — one RCCL all-reduce of A + 2 doubles —
        # runs once, after the timed loop.  `with_collective` is the untimed-for-`value` probe that prices an all-reduce per block.
        enqueue_block()
        if with_collective:
            job_metrics.copy_(metrics)
            sharding.reduce_metrics(job_metrics)
        done_event.record()
        while not done_event.query():  # poll instead of sleeping in the driver: the wake-up of a blocking wait costs more than a step
            pass
        torch.cuda.synchronize(device)

    for _ in range(max(1, math.ceil(args.warmup / K))):  # W warm-up steps, in whole blocks (graph uploaded, caches and clocks warm)
        run_block()
    repeats = max(MIN_BLOCKS, min(2000, math.ceil(TARGET_TIMED_S / (K * 12e-6 + 40e-6))))
    block_s = []
    metrics.zero_()
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        run_block()
        block_s.append(time.perf_counter() - t0)
    barrier()
    if dist is not None:  # the job's ONE collective: the metrics of all timed blocks, summed over the ranks (outside every timed block)
        job_metrics.copy_(metrics)
        sharding.reduce_metrics(job_metrics)
    finished_metrics = (job_metrics if dist is not None else metrics).clone()
    # what a per-block reduction would have cost (never part of `value`): the same block with the all-reduce inside, same bracketing
    collective_probe_s = []
    if dist is not None:
        for _ in range(min(repeats, 60)):
            barrier()
            t0 = time.perf_counter()
            run_block(with_collective=True)
            collective_probe_s.append(time.perf_counter() - t0)
        barrier()
    block_t = torch.tensor(block_s, dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(block_t, op=dist.ReduceOp.MAX)  # MAX over ranks, block by block
    # what the N > 1 record is checked with: the ranks RCCL saw, the devices behind them, every rank's own median block
    distributed = None
    if dist is not None:
        try:
            uuid = str(torch.cuda.get_device_properties(device).uuid)
        except Exception:  # noqa: BLE001
            uuid = f'cuda:{device_index}'
        seen = [None] * world
        dist.all_gather_object(seen, {'rank': rank, 'local_rank': local_rank, 'device': device_index, 'uuid': uuid, 'host': os.uname().nodename,
                                      'block_ms_median': 1e3 * float(np.median(block_s))})
        probe = torch.ones(1, dtype=torch.float64, device=device)
        dist.all_reduce(probe)  # one more collective on the job's backend: the number of ranks it really reduces over
        probe_t = torch.tensor(collective_probe_s, dtype=torch.float64, device=device)
        dist.all_reduce(probe_t, op=dist.ReduceOp.MAX)
        distributed = {'world_size': dist.get_world_size(), 'backend': dist.get_backend(), 'ranks_in_all_reduce': int(probe.item()),
                       'distinct_devices': len({(r['host'], r['uuid']) for r in seen}), 'ranks': seen,
                       'collectives_in_timed_region': 0,
                       'collective': 'one all-reduce of A + 2 float64 metrics after the timed loop (the launches accumulate them on each device)',
                       'block_ms_median_without_collective': 1e3 * float(np.median(block_t.cpu().numpy())),
                       'block_ms_median_with_a_collective_per_block': 1e3 * float(np.median(probe_t.cpu().numpy())),
                       'collective_probe_blocks': len(collective_probe_s)}
    block_s = block_t.cpu().numpy()
    median_s = float(np.median(block_s))
    value = world * B * K / median_s

    # ---- sustained rate (the timed blocks are ~30 ms of work in all): the same block enqueued back to back for
    # SOAK_S seconds, synchronised only every 64 blocks, in windows of ~0.5 s — what the part does once clocks and temperature have settled
    soak = None
    if world == 1 and not args.no_episode_probe:
        SOAK_S, batch_blocks = 3.0, 64
        windows, t_window, blocks_in_window = [], time.perf_counter(), 0
        t_soak = t_window
        while time.perf_counter() - t_soak < SOAK_S:
            for _ in range(batch_blocks):
                enqueue_block()
            torch.cuda.synchronize(device)
            blocks_in_window += batch_blocks
            now = time.perf_counter()
            if now - t_window >= 0.5:
                windows.append(B * K * blocks_in_window / (now - t_window))
                t_window, blocks_in_window = now, 0
        if windows:
            soak = {'seconds': round(time.perf_counter() - t_soak, 2), 'window_s': 0.5, 'env_steps_per_s_per_window': [round(w) for w in windows],
                    'env_steps_per_s_last_window': windows[-1], 'env_steps_per_s_mean': float(np.mean(windows)),
                    'how': f'{K}-step blocks enqueued back to back, one synchronize per {batch_blocks} blocks (no per-block barrier): sustained throughput'}

    # ---- the drop-in Python API path (env.step_random_policy per call, host-bound), reported beside the headline
    state = {'step': 0, 'episode': 0}

    api_seed = base_seed.to(device)  # (device arithmetic: no CPU tensor op inside a timed leg, see torch.set_num_threads above)

    def one_step():
        if state['step'] % EPISODE == 0:
            env.reset(seed=api_seed + 1000003 * state['episode'])
            state['episode'] += 1
        env.step_random_policy(policy_seed=policy_seed, policy_step=state['step'] % EPISODE)
        state['step'] += 1

    api_steps = 4 * EPISODE
    for _ in range(20):
        one_step()
    api_times = []
    for _ in range(5):  # four episodes are 2-3 ms of work: the median of five such runs (one hiccup of the host halves a single one)
        state['step'] = 0
        barrier()
        t1 = time.perf_counter()
        for _ in range(api_steps):
            one_step()
        barrier()
        api_times.append(time.perf_counter() - t1)
    api_value = world * B * api_steps / float(np.median(api_times))

    # ---- kernel-level pass (not part of `value`; always PROBE_EPISODES whole episodes, whatever --steps is): HIP events on the
    # launch stream take each dispatch's own begin and end timestamps (hipExtLaunchKernel start/stop events: what rocprofv3's kernel
    # trace reports, profiles/) + mean task counts per step index over the same episodes for the algorithmic bytes
    # ---- the reference-shaped rollout loop (docs/source/events/moasei-2026/evaluation.md `test()`, baselines/random.py:20): per agent
    # `env.action_space(agent).sample_nested()`, `env.step(actions)`, `torch.all(env.finished)` once per episode, reset per episode — through
    # the drop-in API with its defaults (exact_shapes=True).  The samples are handed to step() untouched, so they are drawn inside the step
    # launch (utils/env.py LazySample): one launch per step; the per-episode reset and the finished test (one host read) are in the time.
    loop_seeds = [base_seed.to(device) + seed_stride * episode for episode in range(8)]  # (on the device: a reset does not cross PCIe)

    def reference_loop_rate(declare_exclusive):
        """env-steps/s of the reference-shaped loop; five runs of eight episodes, the median (and every run, for the record).
        declare_exclusive: `env.set_exclusive_device()` after construction — the one call a drop-in user adds on a dedicated GPU; the env
        then only counts such steps and launches them in chunks (one multi-step launch per chunk: utils/env.py, deferred steps)."""
        loop_env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=EPISODE, device=device, rng=args.rng)
        deferred = bool(declare_exclusive and not rehearsal and loop_env.set_exclusive_device(True) and loop_env._defer_chunk > 0)
        chunks = loop_env._deferred_log = []

        def loop(episodes):
            steps = 0
            for episode in range(episodes):
                loop_env.reset(seed=loop_seeds[episode])
                while True:
                    for _ in range(EPISODE):
                        loop_env.step({agent: loop_env.action_space(agent).sample_nested() for agent in loop_env.agents})
                    steps += EPISODE
                    if torch.all(loop_env.finished):
                        break
            return steps

        loop(2)
        runs = []
        for _ in range(5):
            del chunks[:]
            barrier()
            t2 = time.perf_counter()
            loop_steps = loop(8)
            barrier()
            runs.append(time.perf_counter() - t2)
        loop_env.check()
        launches = len(chunks) if deferred else loop_steps
        del loop_env
        return {'env_steps_per_s': world * B * loop_steps / float(np.median(runs)), 'us_per_step': 1e6 * float(np.median(runs)) / loop_steps,
                'runs_ms': [round(1e3 * t, 3) for t in runs], 'steps_per_run': loop_steps, 'step_launches_per_run': launches, 'deferred_steps': deferred}

    reference_loop_default = reference_loop_rate(False)
    reference_loop_exclusive = reference_loop_rate(True)
    reference_loop_value = reference_loop_exclusive['env_steps_per_s']

    lib, handle = env._lib, env._handle
    mode = _capi.FRZ_RNG_MT19937 if args.rng == 'mt19937' else _capi.FRZ_RNG_PHILOX
    kernel_ms = []
    task_sum = torch.zeros((EPISODE, 1 + A), dtype=torch.float64, device=device)  # per step index: lit fires, attackable fires per agent
    stream = stream_ptr(device)
    # the launches of the timed region: an episode's steps (and its opening reset, and its metrics) are ONE multi-step launch where the
    # library has one for the shape (frz_wildfire_rollout_launches), otherwise a reset launch and one launch per step
    multi_step = lib.frz_wildfire_rollout_launches(handle, EPISODE, mode) == 1
    reset_in_launch = multi_step and args.rng == 'philox'  # (the MT19937 streams are re-seeded by a launch of their own)

    def timed_rollout(n, first_step=0, reset_first=True, with_metrics=True, auto_reset=False):
        """ONE multi-step launch as the timed graph holds it, by events around the dispatch (ms)."""
        spec = _capi.frz_rollout_spec()
        spec.n_steps, spec.rng_mode, spec.policy_seed, spec.first_step = n, mode, policy_seed, first_step
        spec.flags = (_capi.FRZ_ROLLOUT_RESET_FIRST if reset_first else 0) | (_capi.FRZ_ROLLOUT_AUTO_RESET if auto_reset else 0)
        spec.seed_increment = spec.seed_stride = seed_stride
        spec.actions_out = env._actions.data_ptr()
        spec.metrics = metrics.data_ptr() if with_metrics else None
        one = ctypes.c_float()
        _capi.check(lib.frz_wildfire_timed_rollout_spec(handle, ctypes.byref(spec), stream, ctypes.byref(one)), 'frz_wildfire_timed_rollout_spec')
        return one.value

    for episode in range(PROBE_EPISODES):
        env.reset(seed=base_seed + 17 + seed_stride * episode)
        torch.cuda.synchronize(device)
        out = (ctypes.c_float * EPISODE)()
        _capi.check(lib.frz_wildfire_timed_rollout(handle, policy_seed, 0, EPISODE, env._actions.data_ptr(), mode, stream, out), 'frz_wildfire_timed_rollout')
        kernel_ms.extend(out[i] for i in range(EPISODE))
    for episode in range(PROBE_EPISODES):  # the same episodes again, step by step, for the task counts (not timed)
        env.reset(seed=base_seed + 17 + seed_stride * episode)
        for t in range(EPISODE):
            lib.frz_wildfire_step_random_policy(handle, policy_seed, t, env._actions.data_ptr(), mode, None, None, stream)
            task_sum[t, 0] += env.environment_task_count.sum()
            task_sum[t, 1:] += env.agent_task_count.sum(dim=1)
    torch.cuda.synchronize(device)
    task_mean = (task_sum / (PROBE_EPISODES * B)).cpu().numpy()  # [step index][lit fires, attackable per agent]

    def bytes_per_env_step(n_first, state_passes=2.0):
        """algorithmic bytes per env-step averaged over the first n_first steps of an episode (F measured over exactly those steps)"""
        window = task_mean[:n_first]
        return wildfire_bytes_per_env_step(HW, A, env._k, float(window[:, 0].mean()), float(window[:, 1:].sum(axis=1).mean()),
                                           injected_randomness=(args.rng == 'mt19937'), state_passes=state_passes)

    # the dominant kernel of the timed region = the launch a block is made of: the first min(K, 50) steps of an episode with the
    # opening reset and the episode metrics inside (a block of K > 50 steps is ceil(K / 50) such launches; the whole-episode launch is
    # reported beside it).  Both by events around the dispatch itself.
    n_block = min(K, EPISODE)
    block_launch_ms, episode_launch_ms = [], []
    if multi_step and n_block > 1:
        for rep_i in range(12):
            env.reset(seed=base_seed + 17)
            block_launch_ms.append(timed_rollout(n_block, reset_first=reset_in_launch))
            if n_block != EPISODE and not args.no_episode_probe:
                env.reset(seed=base_seed + 17)
                episode_launch_ms.append(timed_rollout(EPISODE, reset_first=reset_in_launch))
        block_launch_ms, episode_launch_ms = block_launch_ms[2:], episode_launch_ms[2:]  # (the first two: clocks and caches)
        if n_block == EPISODE:
            episode_launch_ms = block_launch_ms
    single_step_ms_avg = float(np.mean(kernel_ms))
    steps_per_launch = n_block if block_launch_ms else 1
    kernel_ms_avg = float(np.mean(block_launch_ms)) if block_launch_ms else single_step_ms_avg
    per_env = bytes_per_env_step(n_block)
    per_env_state_once = bytes_per_env_step(n_block, state_passes=1.0 + 1.0 / steps_per_launch)
    per_env_episode = bytes_per_env_step(EPISODE)
    achieved = per_env * B * steps_per_launch / (kernel_ms_avg * 1e-3) / 1e9  # algorithmic bytes of one launch / its duration
    traffic, traffic_source = recorded_traffic(f'wildfire_rollout{steps_per_launch}_bytes_per_launch' if block_launch_ms else 'wildfire_single_step_bytes_per_launch')
    if args.rng != 'philox':  # the counter passes were made in the Philox mode (the MT19937 mode also streams 67 generator words per env-step)
        traffic, traffic_source = None, 'recorded for rng=philox only'

    # ---- continuous rollouts at fixed B: the same K-step block with device-side auto-reset instead of the episode structure
    # (an env that finishes is reset inside the step that finished it: no env idles extinguished until the episode's horizon)
    dense = None
    if multi_step and args.rng == 'philox' and K > 1 and not args.no_episode_probe:
        env.reset(seed=base_seed)
        dense_metrics = torch.zeros_like(metrics)
        dense_graph = env.capture_random_rollout(K, policy_seed=policy_seed, include_reset=False, seed_stride=seed_stride, metrics=dense_metrics,
                                                 auto_reset=True)
        dense_warm = max(3, math.ceil(150 / K))
        for _ in range(dense_warm):  # into the steady state of the continuous rollout (episode phases spread out)
            dense_graph.replay()
        dense_s = []
        for _ in range(repeats):
            barrier()
            t0 = time.perf_counter()
            dense_graph.replay()
            done_event.record()
            while not done_event.query():
                pass
            torch.cuda.synchronize(device)
            dense_s.append(time.perf_counter() - t0)
        dense_t = torch.tensor(dense_s, dtype=torch.float64, device=device)
        if dist is not None:
            dist.all_reduce(dense_t, op=dist.ReduceOp.MAX)
        dense_median = float(np.median(dense_t.cpu().numpy()))
        del dense_graph
        counts = torch.zeros(1 + A, dtype=torch.float64, device=device)
        for t in range(60):  # mean task counts of the steady state (not timed): one-step rollouts, counts read between them
            env.rollout(1, policy_seed=policy_seed, first_step=1000 + t, auto_reset=True, seed_stride=seed_stride)
            counts[0] += env.environment_task_count.sum()
            counts[1:] += env.agent_task_count.sum(dim=1)
        counts = (counts / (60 * B)).tolist()
        dense_launch = [timed_rollout(n_block, first_step=2000 + i * n_block, reset_first=False, with_metrics=True, auto_reset=True) for i in range(10)][2:]
        dense_per_env = wildfire_bytes_per_env_step(HW, A, env._k, counts[0], sum(counts[1:]))
        dense_ms = float(np.mean(dense_launch))
        dense_frac = dense_per_env * B * n_block / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        dense = {'value': world * B * K / dense_median, 'unit': 'env-steps/s', 'ms_per_step': 1e3 * dense_median / K, 'auto_reset': True,
                 'semantics': 'step(); reset_batches(finished, seed + stride) inside the launch (frz_rollout_spec FRZ_ROLLOUT_AUTO_RESET): every env is '
                              'always mid-episode; NOT the headline (the reference loop lets finished envs idle until all are done)',
                 'mean_tasks_per_env': counts[0], 'mean_agent_tasks_per_env': sum(counts[1:]),
                 'episodes_ended_per_block': float(dense_metrics[A + 1].item()) / (repeats + dense_warm),
                 'roofline': {'bound': 'hbm', 'achieved': dense_frac * HBM_PEAK_GBS, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': dense_frac, 'traffic': None,
                              'kernel': f'wf_roles_kernel<6,3,exact,{args.rng},step,multi-step> with auto-reset', 'steps_per_launch': n_block,
                              'kernel_ms_avg': dense_ms, 'kernel_ms_per_step': dense_ms / n_block, 'algorithmic_bytes_per_env_step': dense_per_env}}

    # ---- a rollout a learner can consume (missing #5): the same K-step block with EVERYTHING kept per step — reward / done /
    # sampled-action tapes, the list record (every step's packed lists: the `tasks` of the observations, the action mappings) and the
    # observation tape in its compact form (the suppressant column, the only part of the self / others records a step changes) — the EXTRA
    # instantiation of the multi-step kernel.  N = 1 only, never part of `value`.
    recorded = None
    if multi_step and args.rng == 'philox' and K > 1 and world == 1 and not args.no_episode_probe:
        block, nbytes = ctypes.c_void_p(), ctypes.c_int64()
        _capi.check(lib.frz_wildfire_list_block(handle, ctypes.byref(block), ctypes.byref(nbytes)), 'frz_wildfire_list_block')
        rec_specs, rec_keep = [], []
        for first in range(0, K, EPISODE):
            n = min(EPISODE, K - first)
            tapes = {'rewards': torch.zeros((n, A, B), dtype=torch.float32, device=device), 'dones': torch.zeros((n, 2, B), dtype=torch.uint8, device=device),
                     'actions': torch.zeros((n, A, B, 2), dtype=torch.int32, device=device), 'lists': torch.zeros((max(n - 1, 1), nbytes.value), dtype=torch.uint8, device=device),
                     'observations': torch.zeros((n, A, B), dtype=torch.float32, device=device)}
            spec = _capi.frz_rollout_spec()
            spec.n_steps, spec.rng_mode, spec.policy_seed, spec.first_step = n, mode, policy_seed, 0
            spec.flags, spec.seed_increment = _capi.FRZ_ROLLOUT_RESET_FIRST | _capi.FRZ_ROLLOUT_OBS_COMPACT, seed_stride
            spec.actions_out, spec.record_actions, spec.metrics = tapes['actions'].data_ptr(), 1, metrics.data_ptr()
            spec.reward_tape, spec.done_tape, spec.obs_tape = tapes['rewards'].data_ptr(), tapes['dones'].data_ptr(), tapes['observations'].data_ptr()
            spec.list_record = tapes['lists'].data_ptr() if n > 1 else None
            rec_specs.append(spec)
            rec_keep.append(tapes)
        rec_refs = [ctypes.byref(spec) for spec in rec_specs]
        tape_bytes_per_block = sum(sum(t.numel() * t.element_size() for t in tapes.values()) for tapes in rec_keep)

        def recorded_block():
            for ref in rec_refs:
                lib.frz_wildfire_rollout(handle, ref, stream)
            done_event.record()
            while not done_event.query():
                pass
            torch.cuda.synchronize(device)

        env.reset(seed=base_seed)
        for _ in range(3):
            recorded_block()
        rec_s = []
        for _ in range(min(repeats, 200)):
            barrier()
            t0 = time.perf_counter()
            recorded_block()
            rec_s.append(time.perf_counter() - t0)
        barrier()
        rec_median = float(np.median(rec_s))
        rec_launch = []
        for i in range(10):
            env.reset(seed=base_seed + 17)
            one = ctypes.c_float()
            _capi.check(lib.frz_wildfire_timed_rollout_spec(handle, rec_refs[0], stream, ctypes.byref(one)), 'frz_wildfire_timed_rollout_spec')
            rec_launch.append(one.value)
        rec_launch = rec_launch[2:]
        tape_bytes = 4 * A + 2 + 8 * A + 4 * A  # rewards + dones + sampled actions + suppressants, per env-step (the lists go to the record instead of the scratch copy)
        rec_per_env = bytes_per_env_step(n_block) + tape_bytes
        rec_ms = float(np.mean(rec_launch))
        recorded = {'value': B * K / rec_median, 'unit': 'env-steps/s', 'ms_per_step': 1e3 * rec_median / K, 'blocks': len(rec_s),
                    'keeps_per_step': 'rewards f32[A][B], (terminated, truncated) u8[2][B], sampled actions i32[A][B][2], the packed-list block (task rows = the '
                                      'observations\' `tasks`, action / observation mappings, offsets), suppressants f32[A][B] (the compact observation tape: '
                                      'env.recorded_observations(rec, t) rebuilds {agent: self, others, tasks})',
                    'tape_bytes_per_block': tape_bytes_per_block,
                    'roofline': {'bound': 'hbm', 'achieved': rec_per_env * B * n_block / (rec_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                 'frac': rec_per_env * B * n_block / (rec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 'frac_at_ms_per_step': rec_per_env * B / (rec_median / K) / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                                 'kernel': f'wf_roles_kernel<6,3,exact,{args.rng},step,multi-step,EXTRA> (every option of frz_rollout_spec)',
                                 'steps_per_launch': n_block, 'kernel_ms_avg': rec_ms, 'kernel_ms_per_step': rec_ms / n_block,
                                 'algorithmic_bytes_per_env_step': rec_per_env, 'of_which_tapes': tape_bytes}}
        del rec_keep, rec_specs

    if rank == 0:
        env.check()
        episodes_per_block = math.ceil(K / EPISODE)
        line = {
            'metric': 'env steps/sec (batch x agents) wildfire random-policy',
            'value': value,
            'unit': 'env-steps/s',
            'n_gpus': world,
            'steps': K,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * median_s / K,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'i32/f32',
            'data': 'synthetic',
            'multi_step_launches': bool(exclusive),
            'config': {'workload': f'wildfire_v0 cfg2 (2x3 grid, 3 agents, agent+task openness on), batch={B} per GPU, max_steps={EPISODE}, uniform random '
                                   f'policy sampled inside the step launch, rng={args.rng}',
                       'parallel_envs_per_gpu': B, 'agents': A,
                       'sharding': f'env-batch axis x{world}, no step-path collective, no collective inside a timed block; one metrics all-reduce per job'},
            'timing': {'protocol': (f'{K}-step block = {episodes_per_block} x [reseed, reset, <= {EPISODE} steps, episode metrics] = {episodes_per_block} '
                                    'launch(es) enqueued through the C-ABI' if one_launch_per_episode else
                                    f'{K}-step block = ONE HIP graph ({episodes_per_block} x [reseed, reset, <= {EPISODE} steps, episode metrics])') + '; timed '
                                   f'{repeats} times, each bracketed by barrier + synchronize; median block over the max-over-ranks times',
                       'blocks': repeats, 'block_ms_median': 1e3 * median_s, 'block_ms_min': 1e3 * float(block_s.min()),
                       'block_ms_max': 1e3 * float(block_s.max()), 'block_ms_mean': 1e3 * float(block_s.mean()),
                       'steps_timed_in_total': repeats * K,
                       'launches_per_block': ((1 if reset_in_launch else 2) * episodes_per_block) if multi_step else None,
                       'job_metrics': {'mean_episode_return_per_agent': (finished_metrics[:A] / max(world * B * episodes_per_block * repeats, 1)).tolist(),
                                       'env_steps_counted': float(finished_metrics[A].item())}},
            'distributed': distributed,
            'agent_steps_per_s': value * A,
            'python_api_env_steps_per_s': api_value,
            'python_api_runs_ms': [round(1e3 * t, 3) for t in api_times],  # (five runs of four episodes; the figure above is their median)
            'reference_loop_env_steps_per_s': reference_loop_value,
            'reference_loop': 'per agent env.action_space(agent).sample_nested(), env.step(dict), torch.all(env.finished) once per episode, reset per '
                              'episode; exact_shapes=True (the default); env.set_exclusive_device() declared (this process owns the GPU): step() counts '
                              'such steps and launches them in chunks of 4..32 — one multi-step launch per chunk, run as soon as anything is looked at',
            'reference_loop_exclusive_device': reference_loop_exclusive,
            'reference_loop_without_the_declaration': dict(reference_loop_default, note='no set_exclusive_device(): one step launch per step() call'),
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'frac_algorithmic': achieved / HBM_PEAK_GBS,
                         'frac_algorithmic_state_read_once': per_env_state_once * B * steps_per_launch / (kernel_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         'frac_measured': (traffic / (kernel_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         'fractions': 'frac = frac_algorithmic: SURVEY 8(d) bytes (state read AND written every env-step, reference dtypes) / launch '
                                      'duration / 8 TB/s; frac_algorithmic_state_read_once: the same with the state read once per launch (a multi-step '
                                      'launch keeps it in registers); frac_measured: HBM bytes the counters saw (traffic) / launch duration / 8 TB/s',
                         'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': f'wf_roles_kernel<6,3,exact,{args.rng},step' + (',multi-step>' if steps_per_launch > 1 else '>'),
                         'launch': ((f'the launch a timed block is made of: the first {steps_per_launch} steps of an episode, '
                                     + ('opening reset and ' if reset_in_launch else '') + 'episode metrics inside') if steps_per_launch > 1 else 'one step'),
                         'steps_per_launch': steps_per_launch,
                         'kernel_ms_avg': kernel_ms_avg, 'kernel_ms_median': float(np.median(block_launch_ms if block_launch_ms else kernel_ms)),
                         'launches_timed': len(block_launch_ms) if block_launch_ms else len(kernel_ms),
                         'kernel_ms_per_step': kernel_ms_avg / steps_per_launch,
                         'single_step_launch_ms_avg': single_step_ms_avg,
                         'algorithmic_bytes_per_launch': per_env * B * steps_per_launch,
                         'algorithmic_bytes_per_env_step': per_env,
                         'mean_tasks_per_env': float(task_mean[:n_block, 0].mean()), 'mean_agent_tasks_per_env': float(task_mean[:n_block, 1:].sum(axis=1).mean()),
                         'whole_episode_launch': ({'steps_per_launch': EPISODE, 'kernel_ms_avg': float(np.mean(episode_launch_ms)),
                                                   'kernel_ms_per_step': float(np.mean(episode_launch_ms)) / EPISODE,
                                                   'algorithmic_bytes_per_env_step': per_env_episode,
                                                   'frac': per_env_episode * B * EPISODE / (float(np.mean(episode_launch_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                                  if episode_launch_ms else None),
                         'frac_at_driver_ms_per_step': per_env * B / (median_s / K) / 1e9 / HBM_PEAK_GBS},
            'sustained': soak,
            'auto_reset_workload': dense,
            'recorded_rollout_workload': recorded,
            'reference_cpu_env_steps_per_s': {'value': 21112, 'source': 'BASELINE.md §2: unmodified reference, 8 vCPU, B=65536 (survey container)'},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                available = len(os.sched_getaffinity(0))
            except AttributeError:  # pragma: no cover
                available = os.cpu_count() or 1
            # a one-GPU box grants this process a 16-core share of the host (cgroup quota; more threads than that gain nothing: both runs
            # are reported)
            granted = granted_cores()
            line['cpu_baseline'] = cpu_baseline(granted)
            if available > granted:
                line['cpu_baseline_all_visible_cores'] = cpu_baseline(available, budget_s=5.0)
        if world == 1 and args.rng == 'philox' and not args.no_episode_probe:
            line['mt19937_workload'] = mt19937_workload(device, B, K, base_seed.to(device), policy_seed, seed_stride, repeats, barrier)
        if world == 1 and not args.no_secondary:
            line['secondary_workloads'] = secondary_workloads(device, B)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
