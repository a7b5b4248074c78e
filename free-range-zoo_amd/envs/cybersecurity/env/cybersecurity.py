"""Cybersecurity environment: Python boundary over the fused HIP step kernel.

Mirrors free_range_zoo/envs/cybersecurity/env/cybersecurity.py (``parallel_env`` :112-128, ``raw_env`` :149-584): same
constructor keywords and defaults, agent names (``attacker_i`` / ``defender_i``), observation / mapping attributes,
dtypes and shapes.  The arithmetic runs in ``csrc/cybersecurity.hip`` through ``frz_cybersecurity_*`` (include/frz.h).
"""
import ctypes
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch

from free_range_zoo_amd.utils.spaces import bounds as _space_bounds

from free_range_zoo_amd import _capi
from free_range_zoo_amd.utils.env import BatchedParallelEnv, jagged, stream_ptr
from free_range_zoo_amd.utils.spaces import BatchedOneOfSpace
from free_range_zoo_amd.utils.tensordict import TensorDict
from free_range_zoo_amd.envs.cybersecurity.env.structures.configuration import to_cstruct
from free_range_zoo_amd.envs.cybersecurity.env.structures.state import CybersecurityState
from free_range_zoo_amd.envs.cybersecurity.env.utils import masking


def parallel_env(wrappers: List[Callable] = [], **kwargs) -> 'raw_env':
    env = raw_env(**kwargs)
    for wrapper in wrappers:
        env = wrapper(env)
    return env


def env(wrappers: List[Callable] = [], **kwargs) -> 'BatchedAECView':
    """AEC constructor of the reference (cybersecurity.py): agents act one after the other (``agent_selection`` / ``step`` / ``last``), the
    simulation steps — one launch — when the last one has acted (utils/aec.py)."""
    from free_range_zoo_amd.utils.aec import BatchedAECView
    return BatchedAECView(parallel_env(wrappers, **kwargs))


class raw_env(BatchedParallelEnv):
    """Environment definition for the cybersecurity environment."""
    _rebuild_symbol = 'frz_cybersecurity_rebuild'
    _domain = 'cybersecurity'
    _hands_out_lazy = True  # step() may count steps instead of launching them (utils/env.py: deferred steps): public tensors are EnvTensors

    metadata = {'render.modes': ['human', 'rgb_array'], 'name': 'cybersecurity_v0', 'is_parallelizable': True, 'render_fps': 2,
                'null_value': -100}

    @torch.no_grad()
    def __init__(self, *args, observe_other_location: bool = False, observe_other_presence: bool = False, observe_other_power: bool = True,
                 partially_observable: bool = True, show_bad_actions: bool = True, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.observe_other_power = observe_other_power
        self.observe_other_location = observe_other_location
        self.observe_other_presence = observe_other_presence
        self.partially_obserable = partially_observable  # attribute name as spelled by the reference (cybersecurity.py:185)
        self.show_bad_actions = show_bad_actions
        Att, D = self.attacker_config.num_attackers, self.defender_config.num_defenders
        attackers = tuple(f'attacker_{i}' for i in range(1, Att + 1))
        defenders = tuple(f'defender_{i}' for i in range(1, D + 1))
        self.attacker_name_mapping = dict(zip(attackers, torch.arange(0, Att, device=self.device)))
        self.defender_name_mapping = dict(zip(defenders, torch.arange(0, D, device=self.device)))
        self.possible_agents = attackers + defenders
        self.agents = self.possible_agents
        self.agent_name_mapping = {**self.attacker_name_mapping, **self.defender_name_mapping}
        self.offset_agent_name_mapping = dict(zip(self.possible_agents, torch.arange(0, Att + D, device=self.device)))
        self.agent_observation_mask = lambda agent_name: masking.mask_observation(
            agent_name=agent_name, observe_other_power=observe_other_power, observe_other_presence=observe_other_presence,
            observe_other_location=observe_other_location)
        self._allocate()
        self._exclusive_if_forced()

    def _view(self, ptr: int, shape, dtype) -> torch.Tensor:
        offset = ptr - self._arena.data_ptr()
        numel = 1
        for s in shape:
            numel *= int(s)
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        return self._arena[offset:offset + nbytes].view(dtype).view(*shape)

    def _allocate(self) -> None:
        B = self.parallel_envs
        N, Att, D = self.network_config.num_nodes, self.attacker_config.num_attackers, self.defender_config.num_defenders
        A = Att + D
        self._N, self._Att, self._D = N, Att, D
        self._ka = int(self.observe_other_power) + int(self.observe_other_presence)
        self._kd = self._ka + int(self.observe_other_location)
        self._cfg = to_cstruct(self.config, B, self.max_steps, observe_other_location=self.observe_other_location,
                               observe_other_presence=self.observe_other_presence, observe_other_power=self.observe_other_power,
                               partially_observable=self.partially_obserable, show_bad_actions=self.show_bad_actions)
        self._bind_handle(allocate=True)
        bufs = self._bufs
        f32, i32, i64, u8 = torch.float32, torch.int32, torch.int64, torch.uint8
        v = self._view
        self._network_state, self._location = v(bufs.network_state, (N, B), i32), v(bufs.location, (D, B), i32)
        self._presence = v(bufs.presence, (A, B), torch.bool)
        self._last_action = v(bufs.last_action, (D, B), i32)
        lazy = self._lazy  # (underscore names: plain views for the env's own code, which runs pending steps itself; public names: what callers get)
        self._num_moves = v(bufs.num_moves, (B, ), i32)
        self.num_moves = lazy(self._num_moves)
        self._rewards, self._cumulative = v(bufs.rewards, (A, B), f32), v(bufs.cumulative_rewards, (A, B), f32)
        self._terminations, self._truncations = v(bufs.terminations, (A, B), torch.bool), v(bufs.truncations, (A, B), torch.bool)
        self._self_att, self._self_def = v(bufs.obs_self_attackers, (Att, B, 2), f32), v(bufs.obs_self_defenders, (D, B, 3), f32)
        self._others_att = v(bufs.obs_others_attackers, (Att, B, max(Att - 1, 0), self._ka), f32)
        self._others_def = v(bufs.obs_others_defenders, (D, B, max(D - 1, 0), self._kd), f32)
        self._tasks = v(bufs.obs_tasks, (A, B, N, 2), i64)
        self._act_map_values, self._act_map_offsets = v(bufs.act_map_values, (A, B * N), i32), v(bufs.act_map_offsets, (A, B + 1), i64)
        self._obs_map_values, self._obs_map_offsets = v(bufs.obs_map_values, (B, N), i32), v(bufs.obs_map_offsets, (B + 1, ), i64)
        self._env_task_count, self._agent_task_count = v(bufs.env_task_count, (B, ), i32), v(bufs.agent_task_count, (A, B), i32)
        self.environment_task_count, self.agent_task_count = lazy(self._env_task_count), lazy(self._agent_task_count)
        self._frozen_scaled = v(bufs.frozen_scaled, (B, ), u8)
        self._error_flags = v(bufs.error_flags, (1, ), i32)
        self._actions = v(bufs.actions, (A, B, 2), i32)
        self.generator.attach(seeds=lazy(v(bufs.seeds, (B, ), i32)), states=lazy(v(bufs.mt_state, (624, B), i32)), index=lazy(v(bufs.mt_index, (B, ), i32)))
        self.seeds = self.generator.seeds
        self._state = CybersecurityState(network_state=lazy(self._network_state.t()), location=lazy(self._location.t()), presence=lazy(self._presence.t()))

    def _bind_handle(self, allocate: bool) -> None:
        handle = ctypes.c_void_p()
        _capi.check(self._lib.frz_cybersecurity_create(ctypes.byref(self._cfg), ctypes.byref(handle)), 'frz_cybersecurity_create')
        self._handle = handle
        if allocate:
            nbytes = self._lib.frz_cybersecurity_arena_bytes(self._handle)
            self._arena = self._alloc((nbytes, ), torch.uint8)
        _capi.check(self._lib.frz_cybersecurity_bind(self._handle, self._arena.data_ptr(), stream_ptr(self.device)), 'frz_cybersecurity_bind')
        bufs = _capi.frz_cybersecurity_bufs()
        _capi.check(self._lib.frz_cybersecurity_get_bufs(self._handle, ctypes.byref(bufs)), 'frz_cybersecurity_get_bufs')
        self._bufs = bufs

    def _set_max_steps(self, max_steps) -> None:
        if max_steps != self.max_steps:
            self.max_steps = max_steps
            self._cfg.max_steps = -1 if max_steps is None else int(max_steps)
            self._lib.frz_cybersecurity_destroy(self._handle)
            self._bind_handle(allocate=False)

    def __del__(self):
        try:
            if self._handle is not None:
                self._lib.frz_cybersecurity_destroy(self._handle)
                self._handle = None
        except Exception:  # noqa: BLE001
            pass

    # ---------------------------------------------------------------------------------------- output plumbing
    def _materialize(self) -> None:
        B, A, N, Att = self.parallel_envs, len(self.agents), self._N, self._Att
        if self.exact_shapes:
            totals = self._host_read(self._act_map_offsets[:, -1])  # one small device->host read per step
            act_maps = [jagged(self._act_map_values[a, :totals[a]], self._act_map_offsets[a], max_seqlen=N if totals[a] else 0)
                        for a in range(A)]
        elif getattr(self, '_static_views', None) is None:
            act_maps = [jagged(self._act_map_values[a], self._act_map_offsets[a], max_seqlen=N, lengths=self._agent_task_count[a])
                        for a in range(A)]
            self._static_views = True
        else:
            return
        obs_map = jagged(self._obs_map_values, self._obs_map_offsets, max_seqlen=1)  # [B, j=1, N] (cybersecurity.py:434-439)
        self.task_store = self._tasks[0]
        self.agent_action_mapping, self.agent_observation_mapping, self.agent_bad_actions, self.observations = {}, {}, {}, {}
        for a, agent in enumerate(self.agents):
            self.agent_action_mapping[agent] = act_maps[a]
            self.agent_observation_mapping[agent] = obs_map
            self.agent_bad_actions[agent] = None
            if a < Att:
                own, others = self._self_att[a], self._others_att[a]
            else:
                own, others = self._self_def[a - Att], self._others_def[a - Att]
            self.observations[agent] = TensorDict({'self': own, 'others': others, 'tasks': self._tasks[a]}, batch_size=[B],
                                                  device=self.device)

    # ------------------------------------------------------------------------------- what a recorded rollout keeps of every step
    def _block_bytes(self, which: str) -> int:
        lib, P, I = self._lib, ctypes.c_void_p, ctypes.c_int64
        if which == 'obs':
            block, nbytes = P(), I()
            _capi.check(lib.frz_cybersecurity_obs_block(self._handle, ctypes.byref(block), ctypes.byref(nbytes)), 'frz_cybersecurity_obs_block')
            self._obs_block_base = block.value
            return nbytes.value
        rows, rows_bytes, presence, presence_bytes = P(), I(), P(), I()
        _capi.check(lib.frz_cybersecurity_state_block(self._handle, ctypes.byref(rows), ctypes.byref(rows_bytes), ctypes.byref(presence),
                                                      ctypes.byref(presence_bytes)), 'frz_cybersecurity_state_block')
        self._state_rows_bytes = rows_bytes.value
        return (rows_bytes.value + presence_bytes.value + 255) // 256 * 256  # (a step of the tape starts aligned)

    def recorded_state(self, rec: Dict[str, Any], t: int) -> CybersecurityState:
        """Step ``t`` of ``rollout(..., record_state=True)`` as a CybersecurityState over the tape (views, batch-major like ``env.state()``)."""
        B, N, D, A = self.parallel_envs, self._N, self._D, len(self.possible_agents)
        step = rec['state'][t]
        rows = step[:self._state_rows_bytes].view(torch.int32).view(N + 2 * D, B)
        presence = step[self._state_rows_bytes:self._state_rows_bytes + A * B].view(torch.bool).view(A, B)
        return CybersecurityState(network_state=rows[0:N].t(), location=rows[N:N + D].t(), presence=presence.t())

    def recorded_observations(self, rec: Dict[str, Any], t: int) -> Dict[str, TensorDict]:
        """Step ``t`` of ``rollout(..., record_observations='full')`` as the ``{agent: TensorDict(self, others, tasks)}`` the reference's
        loop gets back from its t-th ``step()`` (utils/conversions.py:92-99; cybersecurity.py:459-511): views of the tape."""
        B, A, N, Att, D = self.parallel_envs, len(self.agents), self._N, self._Att, self._D
        block, base, bufs = rec['observations'][t], self._obs_block_base, self._bufs

        def piece(ptr, shape, dtype):
            n = 1
            for v in shape:
                n *= int(v)
            nbytes = n * torch.empty((), dtype=dtype).element_size()
            return block[ptr - base:ptr - base + nbytes].view(dtype).view(*shape)

        f32 = torch.float32
        self_att, self_def = piece(bufs.obs_self_attackers, (Att, B, 2), f32), piece(bufs.obs_self_defenders, (D, B, 3), f32)
        others_att = piece(bufs.obs_others_attackers, (Att, B, max(Att - 1, 0), self._ka), f32)
        others_def = piece(bufs.obs_others_defenders, (D, B, max(D - 1, 0), self._kd), f32)
        tasks = piece(bufs.obs_tasks, (A, B, N, 2), torch.int64)
        out = {}
        for a, agent in enumerate(self.agents):
            own, others = (self_att[a], others_att[a]) if a < Att else (self_def[a - Att], others_def[a - Att])
            out[agent] = TensorDict({'self': own, 'others': others, 'tasks': tasks[a]}, batch_size=[B], device=self.device)
        return out

    def _publish_dense(self) -> None:
        views = self.__dict__.get('_dense_views')
        if views is None:  # the per-agent rows of the dense blocks: fixed views, made once (a reset builds new dicts over them)
            lazy, A = self._lazy, len(self.possible_agents)
            views = self._dense_views = tuple([lazy(block[a]) for a in range(A)]
                                              for block in (self._rewards, self._cumulative, self._terminations, self._truncations, self._actions))
        agents = self.agents
        self.rewards, self._cumulative_rewards = dict(zip(agents, views[0])), dict(zip(agents, views[1]))
        self.terminations, self.truncations, self.actions = dict(zip(agents, views[2])), dict(zip(agents, views[3])), dict(zip(agents, views[4]))

    # ------------------------------------------------------------------------------------------------- reset
    @torch.no_grad()
    def reset(self, seed=None, options: Optional[Dict[str, Any]] = None):
        self._flush()
        self._defer_chunk = self._DEFER_MIN if self._defer_chunk else 0  # (an episode starts with short chunks: the device gets work at once)
        self._reset_options(options)
        if options and options.get('skip_seeding'):
            if not self.generator.has_been_seeded:
                raise ValueError('Seed must be set before skipping seeding is possible')
        else:
            self.generator.seed(seed, partial_seeding=None)
        self.agents = self.possible_agents
        stream = stream_ptr(self.device)
        self._call('reset')
        self._actions.fill_(-2)  # cybersecurity.py:233-236
        if options is not None and options.get('initial_state') is not None:
            initial_state = options['initial_state']
            if len(initial_state) != self.parallel_envs:
                raise ValueError('Initial state must have the same number of environments as the parallel environments')
            self._state.load_state(initial_state.to(self.device))
            self._call('rebuild')
        self._state.save_initial()
        if options is not None and options.get('initial_state') is not None:
            # the device-side partial reset (reset_finished) restores what was SAVED — the caller's state — like reset_batches (cybersecurity.py:284)
            saved, desc = self._state.initial_state, _capi.frz_cybersecurity_saved_state()
            for name in ('network_state', 'location', 'presence'):
                t = getattr(saved, name)
                setattr(desc, name, t.data_ptr())
                setattr(desc, f'{name}_stride_env', t.stride(0))
                setattr(desc, f'{name}_stride_item', t.stride(1))
            _capi.check(self._lib.frz_cybersecurity_set_saved_initial(self._handle, ctypes.byref(desc)), 'frz_cybersecurity_set_saved_initial')
        self.infos = {agent: {} for agent in self.agents}
        self._has_reset = True
        self._publish()
        self._publish_dense()
        if self.logger is not None:  # _post_reset_hook (utils/env.py:191-195)
            self._log_environment(reset=True)
        return self._observations_out(), self.infos

    @torch.no_grad()
    def reset_batches(self, batch_indices: torch.Tensor, seed: Optional[List[int]] = None, options: Optional[Dict[str, Any]] = None) -> None:
        """Partial reset (cybersecurity.py:268-292 + utils/env.py:162-189)."""
        self._flush()
        batch_indices = torch.as_tensor(batch_indices, device=self.device).long()
        self.generator.seed(seed, partial_seeding=batch_indices)
        self._rewards[:, batch_indices] = 0
        self._cumulative[:, batch_indices] = 0
        self._terminations[:, batch_indices] = False
        self._truncations[:, batch_indices] = False
        self._num_moves[batch_indices] = 0
        self._frozen_scaled[batch_indices] = 0
        self._last_action[:, batch_indices] = -2
        self._actions[:, batch_indices] = -2
        self._state.restore_initial(batch_indices)
        self._call('rebuild')
        self._publish()

    # -------------------------------------------------------------------------------------------------- step
    def step(self, actions, randomness: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """
        One simultaneous step.  ``actions``: ``{agent: IntTensor[B, 2]}`` (attackers ``(node, 0)`` attack / ``(_, -1)`` noop;
        defenders ``(node, 0)`` move / ``-1`` noop / ``-2`` patch / ``-3`` monitor) or a stacked int32 ``[A, B, 2]`` tensor.
        ``randomness``: optional ``(network [1,B,N], agent [1,B,A])`` float32 tensors replacing the env's generator.
        Invalid targets / actions of absent agents raise ``ValueError`` lazily (``env.check()``), not per step.
        """
        if not self._has_reset:
            raise RuntimeError('reset() must be called before step()')
        if randomness is None:  # the reference's random-rollout loop hands over untouched samples of the action spaces (utils/env.py)
            out = self._try_fast_step(actions)
            if out is not None:
                return out
        with torch.no_grad():
            return self._step_given_actions(actions, randomness)

    def _step_given_actions(self, actions, randomness):
        self._flush()
        logged = self._logs_this_step()
        if isinstance(actions, dict):
            self._stage_actions(actions)
            actions = self._actions
        else:
            if actions.dtype != torch.int32 or not actions.is_contiguous() or tuple(actions.shape) != tuple(self._actions.shape):
                raise ValueError('stacked actions must be a contiguous int32 [A, B, 2] tensor')
            self._action_keepalive = actions
            if self.logger is not None:
                self._actions.copy_(actions)
        B, N, A = self.parallel_envs, self._N, len(self.agents)

        def launch(mode, network=None, agent=None):
            self._call('step', (actions.data_ptr(), mode, None if network is None else network.data_ptr(), None if agent is None else agent.data_ptr()),
                       lambda: (actions, mode, network, agent, A, B, N))

        fused_mt = (randomness is None and self.rng == 'mt19937' and not self.single_seeding and self.generator.buffer_size == 0)
        if fused_mt:
            # unbuffered per-env streams: the step launch advances the env's own MT19937 stream (same draws, same order as
            # generator.generate(B, 1, (N,)) followed by generate(B, 1, (A,)), cybersecurity.py:304-315)
            self.generator._ensure_streams()
            launch(_capi.FRZ_RNG_MT19937)
        elif randomness is not None or self.rng == 'mt19937':
            if randomness is None:  # cybersecurity.py:304-315
                network = self.generator.generate(B, 1, (N, ), key='network')
                agent = self.generator.generate(B, 1, (A, ), key='agent')
            else:
                network, agent = randomness
            network = network.to(device=self.device, dtype=torch.float32).contiguous()
            agent = agent.to(device=self.device, dtype=torch.float32).contiguous()
            if network.numel() != B * N or agent.numel() != B * A:
                raise ValueError('randomness tensors have the wrong size')
            self._randomness_keepalive = (network, agent)
            launch(_capi.FRZ_RNG_INJECTED, network, agent)
        else:
            launch(_capi.FRZ_RNG_PHILOX)
        self._publish()
        self.infos = {agent: {} for agent in self.agents}
        if logged:
            self._log_environment()
        return (self._observations_out(), self.rewards, self.terminations, self.truncations, self.infos)

    def _log_extra(self, reset: bool):
        """cybersecurity.py:580-584: the adjacency matrix in every row."""
        return {'adj_matrix': [str(self.network_config.adj_matrix.int().tolist())] * self.parallel_envs}

    @torch.no_grad()
    def random_policy_actions(self, policy_seed: int, policy_step: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._flush()
        out = self._actions if out is None else out
        self._call('random_policy', (policy_seed, policy_step, out.data_ptr()),
                   lambda: (policy_seed, policy_step, out, len(self.agents), self.parallel_envs))
        return out

    @torch.no_grad()
    def step_random_policy(self, policy_seed: int, policy_step: int):
        """``random_policy_actions`` + ``step`` as one launch (same results as the two calls); the sampled actions are left in
        ``self.actions`` (not on a step taken after every env has finished, which is a no-op)."""
        if not self._has_reset:
            raise RuntimeError('reset() must be called before step_random_policy()')
        self._flush()
        logged = self._logs_this_step()
        stream = stream_ptr(self.device)
        if self.rng == 'mt19937':
            if self.single_seeding or self.generator.buffer_size:
                raise NotImplementedError('fused rollouts need the per-env device streams (no single_seeding / buffer_size)')
            self.generator._ensure_streams()
            mode = _capi.FRZ_RNG_MT19937
        else:
            mode = _capi.FRZ_RNG_PHILOX
        self._call('step_random_policy', (policy_seed, policy_step, self._actions.data_ptr(), mode, None, None),
                   lambda: (policy_seed, policy_step, self._actions, mode, len(self.agents), self.parallel_envs))
        self._publish()
        self.infos = {agent: {} for agent in self.agents}
        if logged:
            self._log_environment()
        return (self._observations_out(), self.rewards, self.terminations, self.truncations, self.infos)

    @torch.no_grad()
    def capture_random_rollout(self, steps: int, policy_seed: int = 0, include_reset: bool = True) -> 'torch.cuda.CUDAGraph':
        """``[reset] + steps x (step with the random policy sampled in the launch)`` as one HIP graph (see the wildfire env)."""
        if not self._has_reset:
            raise RuntimeError('reset() must be called once before capturing a rollout')
        self._flush()
        lib, handle, actions = self._lib, self._handle, self._actions.data_ptr()
        mt = self.rng == 'mt19937'
        if mt:
            self.generator._ensure_streams()
        mode = _capi.FRZ_RNG_MT19937 if mt else _capi.FRZ_RNG_PHILOX
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            stream = stream_ptr(self.device)
            if include_reset:
                if mt:
                    _capi.check(lib.frz_mt19937_seed(self._bufs.mt_state, self._bufs.mt_index, self._bufs.seeds, None, 0, self.parallel_envs,
                                                     stream), 'frz_mt19937_seed')
                _capi.check(lib.frz_cybersecurity_reset(handle, stream), 'frz_cybersecurity_reset')
            # (one multi-step launch where the library has one and the device is this env's alone: set_exclusive_device)
            _capi.check(lib.frz_cybersecurity_rollout_random_policy(handle, policy_seed, 0, steps, actions, mode, stream),
                        'frz_cybersecurity_rollout_random_policy')
        return graph

    def _fused_rng_mode(self) -> int:
        if self.rng != 'mt19937':
            return _capi.FRZ_RNG_PHILOX
        if self.single_seeding or self.generator.buffer_size:
            raise NotImplementedError('fused rollouts need the per-env device streams (no single_seeding / buffer_size)')
        self.generator._ensure_streams()
        return _capi.FRZ_RNG_MT19937

    def _check_randomness_tapes(self, steps: int, a: torch.Tensor, b: torch.Tensor) -> None:
        if a.numel() != steps * self.parallel_envs * self._N or b.numel() != steps * self.parallel_envs * len(self.agents):
            raise ValueError('randomness tapes must hold [steps, B, N] and [steps, B, A] float32 values')

    def set_exclusive_device(self, exclusive: bool = True, defer_steps: bool = True) -> bool:
        """State that nothing else uses this GPU while the env's rollouts run: allows ``rollout`` / ``rollout_random_policy`` /
        ``capture_random_rollout`` to run a rollout as ONE launch (include/frz.h: frz_cybersecurity_set_exclusive_device; see the wildfire
        env).  Off by default.  False: the library's own residency check refused (the rollouts keep taking one launch per step)."""
        code = self._lib.frz_cybersecurity_set_exclusive_device(self._handle, 1 if exclusive else 0)
        if code not in (0, _capi.DEFINES['FRZ_E_INVALID']):  # (no device, a dead handle: errors, not a refusal)
            _capi.check(code, 'frz_cybersecurity_set_exclusive_device')
        self._enable_deferral(code == 0 and exclusive, defer_steps)
        return code == 0

    @torch.no_grad()
    def rollout_random_policy(self, steps: int, policy_seed: int = 0, first_step: int = 0):
        """``steps`` x ``step_random_policy`` (same results), enqueued by one call through the C boundary (no CSV / SQL rows: with a logger
        the steps are taken one by one)."""
        if not self._has_reset:
            raise RuntimeError('reset() must be called before rollout_random_policy()')
        self._flush()
        if self.logger is not None or self.rng == 'mt19937' and (self.single_seeding or self.generator.buffer_size):
            out = None
            for t in range(steps):
                out = self.step_random_policy(policy_seed, first_step + t)
            return out
        if self.rng == 'mt19937':
            self.generator._ensure_streams()
        mode = _capi.FRZ_RNG_MT19937 if self.rng == 'mt19937' else _capi.FRZ_RNG_PHILOX
        _capi.check(self._lib.frz_cybersecurity_rollout_random_policy(self._handle, policy_seed, first_step, steps, self._actions.data_ptr(), mode,
                                                                     stream_ptr(self.device)), 'frz_cybersecurity_rollout_random_policy')
        self._publish()
        self.infos = {agent: {} for agent in self.agents}
        return (self._observations_out(), self.rewards, self.terminations, self.truncations, self.infos)

    # ------------------------------------------------------------------------------------------------ spaces
    def action_space(self, agent: str) -> BatchedOneOfSpace:
        """Per-env OneOf (cybersecurity.py:528-551, spaces/actions.py:11-99)."""
        try:  # (the object is count-based over views of the env's buffers — it always describes the current step — so one per agent is built)
            return self._action_spaces[agent]
        except (AttributeError, KeyError):
            pass
        from free_range_zoo_amd.envs.cybersecurity.env.spaces import actions
        index = self.possible_agents.index(agent)
        counts = self.environment_task_count if self.show_bad_actions else self.agent_task_count[index]
        location = self._lazy(self._location[index - self._Att]) if index >= self._Att else None  # (looked at only by code that inspects the members)
        space = actions.build_action_space(agent.split('_')[0], self.show_bad_actions, counts, location, sampler=self._space_sampler(index),
                                           epoch=lambda: self._epoch_counter)
        self.__dict__.setdefault('_action_spaces', {})[agent] = space
        return space

    def observation_space(self, agent: str):
        """The same ``Dict{self, others, tasks}`` for every env (it never changes size; cached like the reference's, cybersecurity.py:553-578)."""
        from free_range_zoo_amd.envs.cybersecurity.env.spaces import observations
        return observations.build_observation_space(
            agent_type=agent.split('_')[0], num_nodes=self._N, parallel_envs=self.parallel_envs, num_attackers=self._Att, num_defenders=self._D,
            attacker_high=_space_bounds(self.config.attacker_observation_bounds),
            defender_high=_space_bounds(self.config.defender_observation_bounds),
            network_high=_space_bounds(self.config.network_observation_bounds), include_power=self.observe_other_power,
            include_presence=self.observe_other_presence, include_location=self.observe_other_location)
