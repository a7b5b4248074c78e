"""Wildfire environment: Python boundary over the fused HIP step kernel.

Mirrors free_range_zoo/envs/wildfire/env/wildfire.py (``parallel_env`` :126-142, ``raw_env`` :163-762): same
constructor keywords, agent names (``firefighter_i``), observation / action-mapping attributes, dtypes and shapes.
All arithmetic of ``step_environment`` / ``update_actions`` / ``update_observations`` runs in
``csrc/wildfire.hip`` through the C-ABI ``frz_wildfire_*`` (include/frz.h).
"""
import ctypes
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch

from free_range_zoo_amd.utils.spaces import bounds as _space_bounds

from free_range_zoo_amd import _capi
from free_range_zoo_amd.utils.env import BatchedParallelEnv, LazyAgentDict, jagged, stream_ptr
from free_range_zoo_amd.utils.spaces import BatchedOneOfSpace
from free_range_zoo_amd.utils.tensordict import TensorDict
from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
from free_range_zoo_amd.envs.wildfire.env.structures.state import WildfireState


def parallel_env(wrappers: List[Callable] = [], **kwargs) -> 'raw_env':
    """Batched parallel wildfire environment (reference wildfire.py:126-142)."""
    env = raw_env(**kwargs)
    for wrapper in wrappers:
        env = wrapper(env)
    return env


def env(wrappers: List[Callable] = [], **kwargs) -> 'BatchedAECView':
    """AEC constructor of the reference (wildfire.py): agents act one after the other (``agent_selection`` / ``step`` / ``last``), the
    simulation steps — one launch — when the last one has acted (utils/aec.py)."""
    from free_range_zoo_amd.utils.aec import BatchedAECView
    return BatchedAECView(parallel_env(wrappers, **kwargs))


class raw_env(BatchedParallelEnv):
    """Environment definition for the wildfire environment."""
    _rebuild_symbol = 'frz_wildfire_rebuild'
    _domain = 'wildfire'
    _hands_out_lazy = True  # step() may defer (utils/env.py: deferred steps): public tensors are EnvTensors

    metadata = {'render.modes': ['human', 'rgb_array'], 'name': 'wildfire_v0', 'is_parallelizable': True, 'render_fps': 2}

    @torch.no_grad()
    def __init__(self, *args, observe_other_suppressant: bool = False, observe_other_power: bool = False,
                 show_bad_actions: bool = False, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.observe_other_suppressant = observe_other_suppressant
        self.observe_other_power = observe_other_power
        self.show_bad_actions = show_bad_actions

        A = self.agent_config.num_agents
        self.possible_agents = tuple(f'firefighter_{i}' for i in range(1, A + 1))
        self.agents = self.possible_agents
        self.agent_name_mapping = dict(zip(self.possible_agents, torch.arange(0, A, device=self.device)))
        self.agent_position_mapping = dict(zip(self.possible_agents, self.agent_config.agents))
        self.ignition_temp = self.fire_config.ignition_temp
        self.max_x, self.max_y = self.config.grid_width, self.config.grid_height
        self.fire_spread_weights = self.config.fire_spread_weights.to(self.device)
        self.fire_reduction_power = self.agent_config.fire_reduction_power
        self.suppressant_states = self.agent_config.suppressant_states
        self.agent_observation_bounds = (self.max_y, self.max_x, self.agent_config.max_fire_reduction_power,
                                         self.agent_config.suppressant_states)
        self.fire_observation_bounds = (self.max_y, self.max_x, self.fire_config.max_fire_type, self.fire_config.num_fire_states)
        agent_ids = torch.arange(0, A, device=self.device)
        self.observation_ordering = {agent: agent_ids[agent_ids != i] for i, agent in enumerate(self.possible_agents)}
        self._allocate()
        self._create_handle()
        self._exclusive_if_forced()

    # ------------------------------------------------------------------------------------------------ buffers
    def _view(self, ptr: int, shape, dtype) -> torch.Tensor:
        """Typed view of a region of the arena (``ptr`` is an absolute device address reported by the library)."""
        offset = ptr - self._arena.data_ptr()
        numel = 1
        for s in shape:
            numel *= int(s)
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        return self._arena[offset:offset + nbytes].view(dtype).view(*shape)

    def _allocate(self) -> None:
        """Create the C handle, allocate its single device arena and wrap every array as a view."""
        B, H, W, A = self.parallel_envs, self.max_y, self.max_x, len(self.possible_agents)
        HW, cap = H * W, self.parallel_envs * H * W
        self._k = 2 + int(self.observe_other_power) + int(self.observe_other_suppressant)
        self._cfg = to_cstruct(self.config, B, self.max_steps, show_bad_actions=self.show_bad_actions,
                               observe_other_power=self.observe_other_power, observe_other_suppressant=self.observe_other_suppressant)
        handle = ctypes.c_void_p()
        _capi.check(self._lib.frz_wildfire_create(ctypes.byref(self._cfg), ctypes.byref(handle)), 'frz_wildfire_create')
        self._handle = handle
        nbytes = self._lib.frz_wildfire_arena_bytes(self._handle)
        if nbytes <= 0:
            raise _capi.FrzError('frz_wildfire_arena_bytes rejected the configuration')
        self._arena = self._alloc((nbytes, ), torch.uint8)  # zero-filled, 256-byte aligned by the caching allocator
        _capi.check(self._lib.frz_wildfire_bind(self._handle, self._arena.data_ptr(), stream_ptr(self.device)), 'frz_wildfire_bind')
        bufs = _capi.frz_wildfire_bufs()
        _capi.check(self._lib.frz_wildfire_get_bufs(self._handle, ctypes.byref(bufs)), 'frz_wildfire_get_bufs')
        self._bufs = bufs
        f32, i32, i64, u8 = torch.float32, torch.int32, torch.int64, torch.uint8
        v = self._view
        # struct-of-arrays HBM state.  Cell arrays: env index innermost for grids of up to 16 cells (one env per lane), env-major — the
        # reference's own layout — above (one env per wavefront, cells across its lanes); `_cells(name)` is [B, H*W] either way
        self._cells_env_major = bool(bufs.cells_env_major)
        cell_shape = (B, HW) if self._cells_env_major else (HW, B)
        self._fires, self._intensity, self._fuel = v(bufs.fires, cell_shape, i32), v(bufs.intensity, cell_shape, i32), v(bufs.fuel, cell_shape, i32)
        self._suppressants, self._capacity = v(bufs.suppressants, (A, B), f32), v(bufs.capacity, (A, B), f32)
        self._equipment = v(bufs.equipment, (A, B), i32)
        # (underscore names: plain views for the env's own code, which flushes pending steps itself; public names: what callers get)
        lazy = self._lazy
        self._num_moves, self._num_burnouts = v(bufs.num_moves, (B, ), i32), v(bufs.num_burnouts, (B, ), i32)
        self.num_moves, self.num_burnouts = lazy(self._num_moves), lazy(self._num_burnouts)
        self._rewards, self._cumulative = v(bufs.rewards, (A, B), f32), v(bufs.cumulative_rewards, (A, B), f32)
        self._terminations, self._truncations = v(bufs.terminations, (A, B), torch.bool), v(bufs.truncations, (A, B), torch.bool)
        self._burnouts, self._putouts = v(bufs.burnouts, (B, ), i64), v(bufs.putouts, (B, ), i64)
        self._obs_self, self._obs_others = v(bufs.obs_self, (A, B, 4), f32), v(bufs.obs_others, (A, B, max(A - 1, 0), self._k), f32)
        self._task_values, self._task_offsets = v(bufs.task_values, (cap, 4), i64), v(bufs.task_offsets, (B + 1, ), i64)
        self._obs_map_values = v(bufs.obs_map_values, (cap, ), i64)
        self._act_map_values, self._act_map_offsets = v(bufs.act_map_values, (A, cap), i64), v(bufs.act_map_offsets, (A, B + 1), i64)
        self._bad_map_values = v(bufs.bad_map_values, (A, cap), i64) if self.show_bad_actions else None
        self._bad_map_offsets = v(bufs.bad_map_offsets, (A, B + 1), i64)
        self._env_task_count, self._agent_task_count = v(bufs.env_task_count, (B, ), i64), v(bufs.agent_task_count, (A, B), i32)
        self.environment_task_count, self.agent_task_count = lazy(self._env_task_count), lazy(self._agent_task_count)
        self._burnouts_out, self._putouts_out = lazy(self._burnouts), lazy(self._putouts)
        self._frozen_scaled = v(bufs.frozen_scaled, (B, ), u8)
        self._error_flags = v(bufs.error_flags, (1, ), i32)
        self._actions = v(bufs.actions, (A, B, 2), i32)
        self.generator.attach(seeds=lazy(v(bufs.seeds, (B, ), i32)), states=lazy(v(bufs.mt_state, (624, B), i32)), index=lazy(v(bufs.mt_index, (B, ), i32)))
        self.seeds = self.generator.seeds
        grid = (lambda t: lazy(t.view(B, H, W))) if self._cells_env_major else (lambda t: lazy(t.view(H, W, B).permute(2, 0, 1)))
        self._state = WildfireState(
            fires=grid(self._fires), intensity=grid(self._intensity), fuel=grid(self._fuel), agents=self.agent_config.agents,
            suppressants=lazy(self._suppressants.t()), capacity=lazy(self._capacity.t()), equipment=lazy(self._equipment.t()))

    def _create_handle(self) -> None:
        pass  # handle, arena and views are created together in _allocate()

    def _save_initial(self) -> None:
        """``State.save_initial()`` (utils/state.py:36-128) as ONE device copy where the state arrays are rows of one block of the arena
        (the env-per-lane kernel families): the saved state is a set of views over the copy, laid out like the live one."""
        if self._cells_env_major:
            self._state.save_initial()
            return
        B, H, W, A = self.parallel_envs, self.max_y, self.max_x, len(self.possible_agents)
        HW = H * W
        first = self._fires.data_ptr() - self._arena.data_ptr()
        rows = 3 * HW + 3 * A  # fires, intensity, fuel, suppressants, capacity, equipment: consecutive rows of the [rows][B] block
        if self._equipment.data_ptr() + A * B * 4 - self._fires.data_ptr() != rows * B * 4:
            self._state.save_initial()
            return
        saved = self.__dict__.get('_initial_block')
        if saved is not None:  # the buffer and the views over it are kept from reset to reset: one copy launch
            saved[0].copy_(self._arena[first:first + rows * B * 4])
            self._state.initial_state = saved[1]
            return
        block = self._arena[first:first + rows * B * 4].clone()
        i32 = block.view(torch.int32).view(rows, B)
        f32 = block.view(torch.float32).view(rows, B)
        grid = lambda t: t.view(H, W, B).permute(2, 0, 1)  # noqa: E731
        saved = WildfireState(fires=grid(i32[0:HW]), intensity=grid(i32[HW:2 * HW]), fuel=grid(i32[2 * HW:3 * HW]), agents=self.agent_config.agents,
                              suppressants=f32[3 * HW:3 * HW + A].t(), capacity=f32[3 * HW + A:3 * HW + 2 * A].t(),
                              equipment=i32[3 * HW + 2 * A:3 * HW + 3 * A].t())
        self._initial_block = (block, saved)
        self._state.initial_state = saved

    def _set_max_steps(self, max_steps) -> None:
        """A new horizon changes the device configuration block: re-create the handle over the same arena."""
        if max_steps != self.max_steps:
            self.max_steps = max_steps
            self._cfg.max_steps = -1 if max_steps is None else int(max_steps)
            self._lib.frz_wildfire_destroy(self._handle)
            handle = ctypes.c_void_p()
            _capi.check(self._lib.frz_wildfire_create(ctypes.byref(self._cfg), ctypes.byref(handle)), 'frz_wildfire_create')
            self._handle = handle
            _capi.check(self._lib.frz_wildfire_bind(self._handle, self._arena.data_ptr(), stream_ptr(self.device)), 'frz_wildfire_bind')

    def __del__(self):
        try:
            if self._handle is not None:
                self._lib.frz_wildfire_destroy(self._handle)
                self._handle = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    # ---------------------------------------------------------------------------------------- output plumbing
    def _materialize(self) -> None:
        """Wrap the persistent output buffers in the reference's dict / TensorDict / nested-tensor structure."""
        B, A = self.parallel_envs, len(self.agents)
        HW = self.max_y * self.max_x
        if self.exact_shapes:
            # one small device->host read: total and maximum length of each jagged output
            stats = torch.cat([self._task_offsets[-1:], self._act_map_offsets[:, -1],
                               self._env_task_count.max().reshape(1), self._agent_task_count.max(dim=1).values.to(torch.int64)])
            if self.show_bad_actions:
                bad = self._env_task_count.unsqueeze(0) - self._agent_task_count
                stats = torch.cat([stats, self._bad_map_offsets[:, -1], bad.max(dim=1).values])
            stats = self._host_read(stats)
            total_f, total_a, max_f, max_a = stats[0], stats[1:1 + A], stats[1 + A], stats[2 + A:2 + 2 * A]
            tasks = jagged(self._task_values[:total_f], self._task_offsets, max_seqlen=max_f)
            obs_map = jagged(self._obs_map_values[:total_f], self._task_offsets, max_seqlen=max_f)
            act_maps = [jagged(self._act_map_values[a, :total_a[a]], self._act_map_offsets[a], max_seqlen=max_a[a]) for a in range(A)]
            if self.show_bad_actions:
                total_b, max_b = stats[2 + 2 * A:2 + 3 * A], stats[2 + 3 * A:]
                bad_maps = [jagged(self._bad_map_values[a, :total_b[a]], self._bad_map_offsets[a], max_seqlen=max_b[a]) for a in range(A)]
        elif getattr(self, '_static_views', None) is None:
            counts = self._env_task_count
            tasks = jagged(self._task_values, self._task_offsets, max_seqlen=HW, lengths=counts)
            obs_map = jagged(self._obs_map_values, self._task_offsets, max_seqlen=HW, lengths=counts)
            act_maps = [jagged(self._act_map_values[a], self._act_map_offsets[a], max_seqlen=HW, lengths=self._agent_task_count[a])
                        for a in range(A)]
            if self.show_bad_actions:
                self._bad_counts = self._env_task_count.unsqueeze(0) - self._agent_task_count
                bad_maps = [jagged(self._bad_map_values[a], self._bad_map_offsets[a], max_seqlen=HW, lengths=self._bad_counts[a])
                            for a in range(A)]
            self._static_views = True
        else:
            if self.show_bad_actions:
                torch.sub(self._env_task_count.unsqueeze(0), self._agent_task_count, out=self._bad_counts)
            return  # persistent views already published

        self.task_store = tasks
        self.agent_action_mapping, self.agent_observation_mapping, self.agent_bad_actions, self.observations = {}, {}, {}, {}
        for a, agent in enumerate(self.agents):
            if self.show_bad_actions:  # wildfire.py:656-660
                self.agent_action_mapping[agent] = obs_map
                self.agent_bad_actions[agent] = bad_maps[a]
            else:
                self.agent_action_mapping[agent] = act_maps[a]
                self.agent_bad_actions[agent] = None
            self.agent_observation_mapping[agent] = obs_map
            self.observations[agent] = TensorDict({'self': self._obs_self[a], 'others': self._obs_others[a], 'tasks': tasks},
                                                  batch_size=[B], device=self.device)
        if not hasattr(self, 'rewards') or self.rewards is None:
            self._publish_dense()

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
        """Reset every env; returns ``(observations, infos)`` like the parallel adapter (conversions.py:39-57)."""
        self._flush()
        self._defer_chunk = self._DEFER_MIN if self._defer_chunk else 0  # (an episode starts with short chunks: the device gets work at once)
        self._reset_options(options)
        if options and options.get('skip_seeding'):
            if not self.generator.has_been_seeded:
                raise ValueError('Seed must be set before skipping seeding is possible')
        else:
            self.generator.seed(seed, partial_seeding=None)
        self.agents = self.possible_agents
        self.rewards = None
        stream = stream_ptr(self.device)
        self._call('reset')
        if options is not None and options.get('initial_state') is not None:
            initial_state = options['initial_state']
            if len(initial_state) != self.parallel_envs:
                raise ValueError('Initial state must have the same number of environments as the parallel environments')
            self._state.load_state(initial_state.to(self.device))
            self._call('rebuild')
        self._save_initial()
        if options is not None and options.get('initial_state') is not None:
            # the device-side partial resets (reset_finished, rollout(auto_reset=True)) restore what was SAVED — the caller's state — like
            # the reference's reset_batches (wildfire.py:391), not the configured initial state
            saved, desc = self._state.initial_state, _capi.frz_wildfire_saved_state()
            for name in ('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment'):
                t = getattr(saved, name)
                flat = t.reshape(t.shape[0], -1) if t.dim() == 3 and t.stride(1) == t.shape[2] * t.stride(2) else t
                if flat.dim() != 2:  # (cells not expressible with one stride: keep a contiguous copy beside the saved state)
                    flat = t.reshape(t.shape[0], -1).contiguous()
                    self.__dict__.setdefault('_saved_keepalive', {})[name] = flat
                setattr(desc, name, flat.data_ptr())
                setattr(desc, f'{name}_stride_env', flat.stride(0))
                setattr(desc, f'{name}_stride_item', flat.stride(1))
            _capi.check(self._lib.frz_wildfire_set_saved_initial(self._handle, ctypes.byref(desc)), 'frz_wildfire_set_saved_initial')
        if self.__dict__.get('fire_rewards') is None:
            self.fire_rewards = self.reward_config.fire_rewards.unsqueeze(0).expand(self.parallel_envs, -1, -1)
        self.infos = {agent: {} for agent in self.agents}
        self._has_reset = True
        self._publish()  # (rewards is None: the dense dicts are rebuilt in there)
        if self.logger is not None:  # _post_reset_hook (utils/env.py:191-195)
            self._log_environment(reset=True)
        return self._observations_out(), self.infos

    @torch.no_grad()
    def reset_batches(self, batch_indices: torch.Tensor, seed: Optional[List[int]] = None, options: Optional[Dict[str, Any]] = None) -> None:
        """Partial reset (wildfire.py:376-397 + utils/env.py:162-189): reseed, zero bookkeeping, restore initial state."""
        self._flush()
        batch_indices = torch.as_tensor(batch_indices, device=self.device).long()
        self.generator.seed(seed, partial_seeding=batch_indices)
        self._rewards[:, batch_indices] = 0
        self._cumulative[:, batch_indices] = 0
        self._terminations[:, batch_indices] = False
        self._truncations[:, batch_indices] = False
        self._num_moves[batch_indices] = 0
        self._num_burnouts[batch_indices] = 0
        self._frozen_scaled[batch_indices] = 0
        self._state.restore_initial(batch_indices)
        self._call('rebuild')
        self._publish()

    # -------------------------------------------------------------------------------------------------- step
    def step(self, actions, randomness: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """
        One simultaneous step of every env.

        Args:
            actions: ``{agent: IntTensor[B, 2]}`` (column 0 = index into the agent's action mapping, column 1 = 0 to
                fight / -1 to noop-refill), or an already stacked int32 ``[A, B, 2]`` device tensor.
            randomness: optional ``(field [3,B,H,W], agent [5,B,A])`` float32 tensors to use instead of the env's
                generator — the reference's own cross-device test convention (randomness as an input).
        Returns:
            ``(observations, rewards, terminations, truncations, infos)`` dicts keyed by agent name.
        """
        if not self._has_reset:
            raise RuntimeError('reset() must be called before step()')
        if randomness is None:  # the reference's random-rollout loop hands over untouched samples of the action spaces (utils/env.py)
            out = self._try_fast_step(actions)
            if out is not None:
                return out
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
        B, H, W, A = self.parallel_envs, self.max_y, self.max_x, len(self.agents)

        def launch(mode, field=None, agent=None):
            self._call('step', (actions.data_ptr(), mode, None if field is None else field.data_ptr(), None if agent is None else agent.data_ptr()),
                       lambda: (actions, mode, field, agent, A, B, H * W))

        fused_mt = (randomness is None and self.rng == 'mt19937' and not self.single_seeding and self.generator.buffer_size == 0)
        if fused_mt:
            # unbuffered per-env streams: the step kernel advances the env's own MT19937 stream (same draws, same order as
            # generator.generate(B, 3, (H, W)) followed by generate(B, 5, (A,)), wildfire.py:409-410)
            self.generator._ensure_streams()
            launch(_capi.FRZ_RNG_MT19937)
        elif randomness is not None or self.rng == 'mt19937':
            if randomness is None:  # wildfire.py:409-410
                field = self.generator.generate(B, 3, (H, W), key='field')
                agent = self.generator.generate(B, 5, (A, ), key='agent')
            else:
                field, agent = randomness
            field = field.to(device=self.device, dtype=torch.float32).contiguous()
            agent = agent.to(device=self.device, dtype=torch.float32).contiguous()
            if field.numel() != 3 * B * H * W or agent.numel() != 5 * B * A:
                raise ValueError('randomness tensors have the wrong size')
            self._randomness_keepalive = (field, agent)
            launch(_capi.FRZ_RNG_INJECTED, field, agent)
        else:
            launch(_capi.FRZ_RNG_PHILOX)
        self._publish()
        self.infos = {agent: {} for agent in self.agents}
        self.infos['burnouts'] = self._burnouts_out
        self.infos['putouts'] = self._putouts_out
        if logged:
            self._log_environment()
        return (self._observations_out(), self.rewards, self.terminations, self.truncations, self.infos)

    def _log_extra(self, reset: bool):
        """wildfire.py:755-762: the step's burnouts / putouts (NULL in the reset row)."""
        if reset:
            return {'burnouts': [None] * self.parallel_envs, 'putouts': [None] * self.parallel_envs}
        return {'burnouts': self._burnouts, 'putouts': self._putouts}

    @torch.no_grad()
    def random_policy_actions(self, policy_seed: int, policy_step: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Device-side uniform random policy over each agent's current action space -> int32 ``[A, B, 2]``."""
        self._flush()
        out = self._actions if out is None else out
        self._call('random_policy', (policy_seed, policy_step, out.data_ptr()),
                   lambda: (policy_seed, policy_step, out, len(self.agents), self.parallel_envs))
        return out

    @torch.no_grad()
    def capture_random_rollout(self, steps: int, policy_seed: int = 0, include_reset: bool = True, episode_length: Optional[int] = None,
                               seed_stride: int = 0, metrics: Optional[torch.Tensor] = None,
                               metrics_copy: Optional[torch.Tensor] = None, auto_reset: bool = False) -> 'torch.cuda.CUDAGraph':
        """
        Capture ``[reset] + steps x (random policy + step)`` into a HIP graph and return it: one ``frz_wildfire_rollout`` per episode —
        ONE launch where the library has a multi-step kernel for the shape (the opening reset, the steps and the episode metrics all
        inside it), otherwise a reset launch and one launch per step.

        Launch-bound rollouts are replayed with ``graph.replay()`` without per-step host work; results land in the same persistent
        buffers ``step()`` fills.  The env seeds are read at replay time (``env.seeds`` may be changed between replays); the reset
        inside the graph restores the configured initial state (it does not re-run the Python-side ``save_initial``).

        A whole rollout loop as ONE graph: with ``episode_length`` the ``steps`` are cut into episodes, each starting with a
        reset (policy steps restart at 0); ``seed_stride`` (an int) is added to ``env.seeds`` by every reset (fresh seeds per
        episode); ``metrics`` (float64 ``[A + 2]``) receives ``accumulate_episode_metrics`` after every episode and is copied to
        ``metrics_copy`` at the end of the graph (the buffer a collective then reduces).

        ``auto_reset=True``: no episode structure — the ``steps`` are one continuous rollout in which an env that finishes is reset
        inside the step that finished it (seed += ``seed_stride``); ``metrics`` then receives the returns of the episodes that ended.
        """
        if not self._has_reset:
            raise RuntimeError('reset() must be called once before capturing a rollout')
        self._flush()
        self._no_multi_step_while_consistent('capture_random_rollout()')
        if metrics is not None and (metrics.dtype != torch.float64 or metrics.numel() != len(self.agents) + 2 or not metrics.is_contiguous()):
            raise ValueError('metrics must be a contiguous float64 [A + 2] tensor on the env device')
        lib, handle = self._lib, self._handle
        mode = self._fused_rng_mode()
        episode = steps if (not episode_length or auto_reset) else int(episode_length)
        spec = _capi.frz_rollout_spec()
        spec.rng_mode, spec.policy_seed, spec.actions_out = mode, int(policy_seed), self._actions.data_ptr()
        spec.metrics = metrics.data_ptr() if metrics is not None else None
        spec.flags = (_capi.FRZ_ROLLOUT_RESET_FIRST if include_reset else 0) | (_capi.FRZ_ROLLOUT_AUTO_RESET if auto_reset else 0)
        spec.seed_increment, spec.seed_stride = int(seed_stride), int(seed_stride) & 0xFFFFFFFF
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        # thread-local capture: other threads of the process (e.g. the RCCL watchdog of torch.distributed) may touch the HIP
        # runtime while this thread records the launches
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            stream = stream_ptr(self.device)
            done = 0
            while True:
                n = min(episode, steps - done)
                spec.n_steps, spec.first_step = n, 0 if include_reset else done
                _capi.check(lib.frz_wildfire_rollout(handle, ctypes.byref(spec), stream), 'frz_wildfire_rollout')
                done += n
                if done >= steps:
                    break
            if metrics is not None and metrics_copy is not None:
                metrics_copy.copy_(metrics)
        return graph

    def _fused_rng_mode(self) -> int:
        """RNG mode of the fused policy + step entry points: they advance the raw per-env device streams, which is what step() draws
        from only without single_seeding and without a draw buffer (utils/random_generator.py:116-146 semantics need step())."""
        if self.rng != 'mt19937':
            return _capi.FRZ_RNG_PHILOX
        if self.single_seeding or self.generator.buffer_size:
            raise NotImplementedError('fused rollouts need the per-env device streams (no single_seeding / buffer_size)')
        self.generator._ensure_streams()
        return _capi.FRZ_RNG_MT19937

    def _after_fused(self, logged: bool):
        self._publish()
        self.infos = {agent: {} for agent in self.agents}
        self.infos['burnouts'] = self._burnouts_out
        self.infos['putouts'] = self._putouts_out
        if logged:
            self._log_environment()
        return (self._observations_out(), self.rewards, self.terminations, self.truncations, self.infos)

    def _step_infos(self) -> dict:
        infos = {agent: {} for agent in self.agents}
        infos['burnouts'] = self._burnouts_out
        infos['putouts'] = self._putouts_out
        return infos

    @torch.no_grad()
    def step_random_policy(self, policy_seed: int, policy_step: int):
        """``random_policy_actions`` + ``step`` as one launch (same results as the two calls, CSV log rows included); the sampled actions
        are left in ``self.actions``."""
        if not self._has_reset:
            raise RuntimeError('reset() must be called before step_random_policy()')
        self._flush()
        logged = self._logs_this_step()
        mode = self._fused_rng_mode()
        self._call('step_random_policy', (policy_seed, policy_step, self._actions.data_ptr(), mode, None, None),
                   lambda: (policy_seed, policy_step, self._actions, mode, len(self.agents), self.parallel_envs))
        return self._after_fused(logged)

    def _check_randomness_tapes(self, steps: int, a: torch.Tensor, b: torch.Tensor) -> None:
        B, HW, A = self.parallel_envs, self.max_y * self.max_x, len(self.agents)
        if a.numel() != steps * 3 * B * HW or b.numel() != steps * 5 * B * A:
            raise ValueError('randomness tapes must hold [steps, 3, B, H, W] and [steps, 5, B, A] float32 values')

    def _after_rollout(self) -> None:
        self._publish()
        self.infos = {agent: {} for agent in self.agents}
        self.infos['burnouts'] = self._burnouts_out
        self.infos['putouts'] = self._putouts_out

    # ------------------------------------------------------------------------------- what a recorded rollout keeps of every step
    def _block_bytes(self, which: str) -> int:
        lib, P, I = self._lib, ctypes.c_void_p, ctypes.c_int64
        if which == 'obs':
            block, nbytes, others = P(), I(), I()
            _capi.check(lib.frz_wildfire_obs_block(self._handle, ctypes.byref(block), ctypes.byref(nbytes), ctypes.byref(others)), 'frz_wildfire_obs_block')
            self._obs_others_offset = others.value
            return nbytes.value
        cells, cells_bytes, agents, agents_bytes = P(), I(), P(), I()
        _capi.check(lib.frz_wildfire_state_block(self._handle, ctypes.byref(cells), ctypes.byref(cells_bytes), ctypes.byref(agents), ctypes.byref(agents_bytes)),
                    'frz_wildfire_state_block')
        self._state_cells_bytes = cells_bytes.value
        return cells_bytes.value + agents_bytes.value

    def recorded_state(self, rec: Dict[str, Any], t: int) -> WildfireState:
        """Step ``t`` of ``rollout(..., record_state=True)`` as a WildfireState over the tape (views, batch-major like ``env.state()``)."""
        B, H, W, A = self.parallel_envs, self.max_y, self.max_x, len(self.possible_agents)
        HW = H * W
        step = rec['state'][t]
        cells = step[:self._state_cells_bytes].view(torch.int32)
        agents_i = step[self._state_cells_bytes:].view(torch.int32).view(3 * A, B)
        agents_f = step[self._state_cells_bytes:].view(torch.float32).view(3 * A, B)
        if self._cells_env_major:
            fires, intensity, fuel = cells.view(3, B, H, W)
        else:
            fires, intensity, fuel = (cells.view(3, H, W, B)[k].permute(2, 0, 1) for k in range(3))
        return WildfireState(fires=fires, intensity=intensity, fuel=fuel, agents=self.agent_config.agents, suppressants=agents_f[0:A].t(),
                             capacity=agents_f[A:2 * A].t(), equipment=agents_i[2 * A:3 * A].t())

    def recorded_observations(self, rec: Dict[str, Any], t: int) -> Dict[str, TensorDict]:
        """Step ``t`` of a recorded rollout as the ``{agent: TensorDict(self, others, tasks)}`` the reference's loop gets back from its t-th
        ``step()`` (utils/conversions.py:92-99; wildfire.py:662-717): ``self`` / ``others`` from the observation tape (the compact form is
        expanded: positions and base power are configuration, the suppressant column is what the tape holds), ``tasks`` from the list record
        (``record=True``; the last step's lists are the env's own).  Reads the step's list total from the device."""
        B, A, k = self.parallel_envs, len(self.agents), self._k
        steps = rec['observations'].shape[0]
        self._flush()
        if rec.get('observations_form') == 'compact':
            supp = rec['observations'][t]  # [A, B]
            obs_self = self._obs_self.clone()
            obs_self[:, :, 3] = supp
            obs_others = self._obs_others.clone()  # [A, B, A - 1, k]: (y, x[, power][, suppressant])
            if self.observe_other_suppressant and A > 1:
                for a, agent in enumerate(self.agents):
                    obs_others[a, :, :, k - 1] = supp[self.observation_ordering[agent]].t()
        else:
            block = rec['observations'][t]
            obs_self = block[:A * B * 16].view(torch.float32).view(A, B, 4)
            obs_others = block[self._obs_others_offset:self._obs_others_offset + A * B * max(A - 1, 0) * k * 4].view(torch.float32).view(A, B, max(A - 1, 0), k)
        if t < steps - 1:
            if 'lists' not in rec:
                raise ValueError('the tasks of an intermediate step are in the list record: rollout(..., record=True)')
            base = self._arena.data_ptr() + rec['list_block_offset']
            block = rec['lists'][t]
            offsets = block[self._bufs.task_offsets - base:self._bufs.task_offsets - base + (B + 1) * 8].view(torch.int64)
            total = int(offsets[-1])
            values = block[self._bufs.task_values - base:self._bufs.task_values - base + total * 32].view(torch.int64).view(total, 4)
        else:
            offsets = self._task_offsets
            total = int(offsets[-1])
            values = self._task_values[:total]
        tasks = jagged(values, offsets)
        return {agent: TensorDict({'self': obs_self[a], 'others': obs_others[a], 'tasks': tasks}, batch_size=[B], device=self.device)
                for a, agent in enumerate(self.agents)}

    def set_exclusive_device(self, exclusive: bool = True, defer_steps: bool = True) -> bool:
        """State that nothing else uses this GPU while the env's rollouts run (no other process, no concurrent stream).  It allows
        ``rollout`` / ``rollout_random_policy`` / ``capture_random_rollout`` to run a whole rollout as ONE launch where the library has a
        multi-step kernel for the shape (include/frz.h: frz_wildfire_set_exclusive_device): its workgroups wait inside the kernel for each
        other between steps, which is only safe when all of them are resident.  Off by default.  The library checks its own part — the
        launch's workgroups fit on the device that owns the arena (occupancy query x compute units, no CU mask): if they do not, the request
        is refused, rollouts keep taking one launch per step, and False is returned."""
        self._flush()
        self._defer_chunk = 0
        if exclusive and self.__dict__.get('_global_group', False) is not False:
            return False  # globally consistent batch semantics exchange the totals between any two steps
        code = self._lib.frz_wildfire_set_exclusive_device(self._handle, 1 if exclusive else 0)
        if code not in (0, _capi.DEFINES['FRZ_E_INVALID']):  # (no device, a dead handle: errors, not a refusal)
            _capi.check(code, 'frz_wildfire_set_exclusive_device')
        self._enable_deferral(code == 0 and exclusive, defer_steps)
        return code == 0

    # ------------------------------------------------------------------------ sharded jobs: globally consistent batch semantics (optional)
    def set_global_consistency(self, enabled: bool = True, group=None) -> None:
        """Evaluate the two batch-global semantics of a step over the WHOLE sharded job instead of this shard (off by default):
        the all-done early-out (utils/env.py:211-213) and the skip of an agent without a task in any env (wildfire.py:434-435).  After
        every ``reset`` / ``step`` / ``step_random_policy`` / ``update_observations`` the batch totals the kernels keep (lit fires, fires
        each agent can attack, envs not terminated / not truncated: A + 3 integers) are summed over the ranks of ``group`` — one tiny
        ``all_reduce`` per step, stream-ordered with RCCL — and the next step reads the sums: a sharded run is then the unsharded one env
        for env also when a whole shard finishes early or with ``show_bad_actions``.  Multi-step launches cannot stop for the exchange:
        ``rollout`` / ``rollout_random_policy`` / ``capture_random_rollout`` raise while this is on."""
        self._flush()
        self._global_group = group if enabled else False
        if enabled:
            self._defer_chunk = 0
            self._lib.frz_wildfire_set_exclusive_device(self._handle, 0)
            if self.__dict__.get('_totals_staging') is None:
                self._totals_staging = torch.zeros(len(self.possible_agents) + 3, dtype=torch.int32, device=self.device)

    def _publish(self) -> None:
        if self.__dict__.get('_global_group', False) is not False:
            self._exchange_batch_totals()
        super()._publish()

    def _exchange_batch_totals(self) -> None:
        import torch.distributed as dist
        from free_range_zoo_amd.utils import sharding
        staging, stream = self._totals_staging, stream_ptr(self.device)
        _capi.check(self._lib.frz_wildfire_export_totals(self._handle, staging.data_ptr(), stream), 'frz_wildfire_export_totals')
        if dist.is_available() and dist.is_initialized() and dist.get_backend(self._global_group) == 'gloo':
            host = staging.cpu()  # (gloo reduces host tensors: the CPU rehearsal of the exchange; RCCL keeps it on the device)
            sharding.globalize_totals(host, self._global_group)
            staging.copy_(host)
        else:
            sharding.globalize_totals(staging, self._global_group)
        _capi.check(self._lib.frz_wildfire_import_totals(self._handle, staging.data_ptr(), stream), 'frz_wildfire_import_totals')

    def _no_multi_step_while_consistent(self, what: str) -> None:
        if self.__dict__.get('_global_group', False) is not False:
            raise NotImplementedError(f'{what} enqueues several steps without the per-step exchange of set_global_consistency(); use step() / '
                                      f'step_random_policy()')

    @torch.no_grad()
    def rollout_random_policy(self, steps: int, policy_seed: int = 0, first_step: int = 0):
        """``steps`` x ``step_random_policy`` (same results), enqueued by one call through the C boundary.  With ``log_directory`` set the
        steps are taken one by one so that every one of them reaches the CSV files."""
        if not self._has_reset:
            raise RuntimeError('reset() must be called before rollout_random_policy()')
        self._flush()
        self._no_multi_step_while_consistent('rollout_random_policy()')
        if self.logger is not None:
            out = None
            for t in range(steps):
                out = self.step_random_policy(policy_seed, first_step + t)
            return out if out is not None else self._after_fused(False)
        mode = self._fused_rng_mode()
        _capi.check(self._lib.frz_wildfire_rollout_random_policy(self._handle, policy_seed, first_step, steps, self._actions.data_ptr(), mode,
                                                                 stream_ptr(self.device)), 'frz_wildfire_rollout_random_policy')
        return self._after_fused(False)

    @torch.no_grad()
    def accumulate_episode_metrics(self, metrics: torch.Tensor) -> torch.Tensor:
        """``metrics`` (float64 ``[A + 2]`` on the device) += (cumulative reward per agent ..., env-steps taken, finished envs): one launch."""
        if metrics.dtype != torch.float64 or metrics.numel() != len(self.agents) + 2 or not metrics.is_contiguous() or not metrics.is_cuda:
            raise ValueError('metrics must be a contiguous float64 [A + 2] tensor on the env device')
        self._flush()
        _capi.check(self._lib.frz_wildfire_episode_metrics(self._handle, metrics.data_ptr(), stream_ptr(self.device)), 'frz_wildfire_episode_metrics')
        return metrics

    # ------------------------------------------------------------------------------------------------ spaces
    def action_space(self, agent: str) -> BatchedOneOfSpace:
        """Per-env ``OneOf([fight task]*n + [noop])`` (wildfire.py:719-734, spaces/actions.py:23-41).  The object is count-based over views
        of the env's buffers — it always describes the current step — so one per agent is built and handed out again."""
        try:
            return self._action_spaces[agent]
        except (AttributeError, KeyError):
            pass
        cache = self.__dict__.setdefault('_action_spaces', {})
        space = cache.get(agent)
        if space is None:
            index = self.possible_agents.index(agent)
            counts = self.environment_task_count if self.show_bad_actions else self.agent_task_count[index]  # (EnvTensors: a look at them runs pending steps)
            from free_range_zoo_amd.envs.wildfire.env.spaces import actions
            space = cache[agent] = actions.build_action_space(counts, sampler=self._space_sampler(index))
        return space

    def observation_space(self, agent: str):
        """Per-env ``Dict{self, others, tasks}`` sized by the env's lit fires (wildfire.py:736-753), count-based."""
        from free_range_zoo_amd.envs.wildfire.env.spaces import observations
        return observations.build_observation_space(
            environment_task_counts=self.environment_task_count, num_agents=len(self.possible_agents),
            agent_high=_space_bounds(self.agent_observation_bounds), fire_high=_space_bounds(self.fire_observation_bounds),
            include_suppressant=self.observe_other_suppressant, include_power=self.observe_other_power)
