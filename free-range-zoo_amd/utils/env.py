"""Batched parallel environment base (mirrors the reference's L3 + L4 layers in one object).

Reference: ``BatchedAECEnv`` (free_range_zoo/utils/env.py:18-359) wrapped by
``batched_aec_to_batched_parallel_wrapper`` (free_range_zoo/utils/conversions.py:32-118).  The reference steps the
AEC env once per agent and only the last call does work; here ``step(actions)`` is ONE fused HIP kernel launch over
persistent struct-of-arrays HBM buffers, and the AEC facade (``env.aec_env``) is the same object.

Behavioural notes kept from the reference:
  * rewards / terminations / truncations / infos are dicts keyed by agent name; values are tensors over the env batch;
  * ``finished = terminated | truncated`` with ``terminated`` / ``truncated`` = all agents' flags (env.py:331-359);
  * once ALL envs are terminated or ALL are truncated, ``step`` changes nothing (env.py:211-213) and the adapter returns
    the stale rewards summed once per agent (conversions.py:87-90) — reproduced on the device without a host sync.
Returned tensors are views of persistent buffers: valid until the next ``step()`` / ``reset()``.
"""
import ctypes
from typing import Any, Dict, List, Optional

import torch

from free_range_zoo_amd import _capi
from free_range_zoo_amd.utils.configuration import Configuration
from free_range_zoo_amd.utils.random_generator import RandomGenerator


def stream_ptr(device) -> int:
    """Raw handle of the current stream of `device` (a torch.device with an index)."""
    return torch._C._cuda_getCurrentRawStream(device.index)


def jagged(values: torch.Tensor, offsets: torch.Tensor, max_seqlen: Optional[int] = None, lengths: Optional[torch.Tensor] = None):
    """torch.nested jagged tensor over (values, offsets) — O(1) Python objects instead of the reference's O(B) split."""
    return torch.nested.nested_tensor_from_jagged(values, offsets, lengths=lengths, max_seqlen=max_seqlen)


class LazyAgentDict(dict):
    """``{agent: observation}`` as ``step()`` / ``reset()`` return it, filled on first use.  The jagged parts of an observation need the
    step's list lengths on the host to take their exact shape (``exact_shapes=True``, the default): that one small device read happens when
    an observation is actually looked at, not on every ``step()`` — rollouts whose policy runs on the device (``step_random_policy``,
    ``action_space(agent).sample_nested()``, the scripted baselines) never pay it.  Like every tensor the env hands out, the contents are
    views of the env's buffers, valid until its next ``step()`` / ``reset()``."""

    def __init__(self, env, agents):
        super().__init__()
        self._env, self._agents, self._filled = env, tuple(agents), False
        self._epoch = env._epoch_counter  # the step whose observations this dict stands for

    def _fill(self):
        if not self._filled:
            if self._env._epoch_counter != self._epoch:
                # the buffers behind this dict now hold a later step: filling it would silently hand out that step's observations (a replay
                # buffer that stores `obs` and reads it after the next step would get `next_obs` twice)
                raise RuntimeError('these observations belong to an earlier step(): like every tensor the env hands out they were valid until the '
                                   'next step() / reset() — look at (or .copy()) them before stepping again')
            self._filled = True
            observations = self._env.observations
            dict.update(self, {agent: observations[agent] for agent in self._agents})
        return self

    def __getitem__(self, key):
        return dict.__getitem__(self._fill(), key)

    def get(self, key, default=None):
        return dict.get(self._fill(), key, default)

    def __iter__(self):
        return iter(self._agents)

    def __len__(self):
        return len(self._agents)

    def __contains__(self, key):
        return key in self._agents

    def keys(self):
        return dict.fromkeys(self._agents).keys()

    def values(self):
        return dict.values(self._fill())

    def items(self):
        return dict.items(self._fill())

    def __eq__(self, other):
        return dict.__eq__(self._fill(), other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        return dict.__repr__(self._fill())

    def copy(self):
        return dict(self._fill())


class LazySample(torch.Tensor):
    """What ``env.action_space(agent).sample_nested()`` returns: the agent's int32 ``[B, 2]`` slice of the env's sample buffer, whose draw
    is made on first use.  The reference's rollout loop — ``{agent: env.action_space(agent).sample_nested() for agent in env.agents}`` handed
    straight to ``env.step`` (docs/source/events/moasei-2026/evaluation.md ``test()``, baselines/random.py:20) — never looks at the samples
    itself: ``step`` then draws them INSIDE its own launch (``frz_<domain>_step_random_policy``: same stream, same values as the policy
    launch would have produced) and leaves them in this buffer.  Any other use — a torch function, ``.cpu()``, indexing, printing —
    launches the policy kernel first.  Like every tensor the env hands out: valid until the next ``step()`` / ``reset()``."""

    @staticmethod
    def __new__(cls, view, env):
        t = torch.Tensor._make_subclass(cls, view)
        t._env = env
        return t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def plain(x):
            if isinstance(x, LazySample):
                x._env._materialize_samples()
                return x.as_subclass(torch.Tensor)
            if isinstance(x, (list, tuple)):
                return type(x)(plain(v) for v in x)
            return x
        return func(*plain(args), **{k: plain(v) for k, v in (kwargs or {}).items()})


_NO_SUBCLASS_DISPATCH = torch._C.DisableTorchFunctionSubclass
# attribute reads that say nothing about the contents: they do not make pending steps run
_METADATA_GETTERS = frozenset(getattr(torch.Tensor, name).__get__ for name in ('shape', 'dtype', 'device', 'ndim', 'is_cuda', 'layout', 'requires_grad'))


class EnvTensor(torch.Tensor):
    """A view of the env's device buffers as the env hands it out (``rewards[agent]``, ``num_moves``, ``state().fires``, ...) when steps may
    be DEFERRED (``set_exclusive_device``: the reference-shaped random loop enqueues its steps in chunks — one multi-step launch per chunk
    instead of one launch per step).  It is an ordinary tensor — same storage, same values — whose every use as a tensor (a torch function,
    a method, indexing, ``.cpu()``, printing, ``data_ptr()``) first runs the steps that are still pending, so that what is read is what a
    step-by-step execution would have left.  Views derived from it stay ``EnvTensor``s; everything else it produces is a plain tensor.
    Like every tensor the env hands out: valid until the next ``step()`` / ``reset()``."""

    @staticmethod
    def __new__(cls, view, env):
        t = torch.Tensor._make_subclass(cls, view)
        t._env = env
        return t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        env = None
        for a in args:
            if type(a) is EnvTensor:
                env = a._env
                break
        else:
            for a in (kwargs or {}).values():
                if type(a) is EnvTensor:
                    env = a._env
                    break
            else:  # (nested in a list: torch.stack([...]), torch.cat([...]))
                for a in args:
                    if isinstance(a, (list, tuple)):
                        for v in a:
                            if type(v) is EnvTensor:
                                env = v._env
                                break
        if env is not None and env._deferred and func not in _METADATA_GETTERS:
            env._flush()
        with _NO_SUBCLASS_DISPATCH():
            out = func(*args, **(kwargs or {}))
        if env is not None and type(out) is torch.Tensor and out._is_view():
            return EnvTensor(out, env)
        return out

    # leaving the process or being copied, an EnvTensor is the plain tensor it shows (never the env behind it)
    def __reduce_ex__(self, protocol):
        self._env._flush()
        return self.as_subclass(torch.Tensor).__reduce_ex__(protocol)

    def __deepcopy__(self, memo):
        self._env._flush()
        return self.as_subclass(torch.Tensor).clone()


class BatchedParallelEnv:
    """Common constructor / bookkeeping of the three domains."""

    metadata: Dict[str, Any] = {}

    def __init__(self,
                 *args,
                 configuration: Configuration = None,
                 max_steps: int = 1,
                 parallel_envs: int = 1,
                 device: torch.device = torch.device('cuda'),
                 render_mode: Optional[str] = None,
                 log_directory: Optional[str] = None,
                 single_seeding: bool = False,
                 buffer_size: int = 0,
                 override_initialization_check: bool = False,
                 rng: str = 'mt19937',
                 exact_shapes: bool = True,
                 dispatch: Optional[str] = None,
                 **kwargs):
        """
        Same keyword arguments as the reference (utils/env.py:21-34), plus:
            rng: 'mt19937' — per-env streams identical to the reference's CPU generator for the same seeds (default);
                 'philox'  — counter-based in-kernel Philox4x32-10 keyed by the env seed (no RNG state traffic).
            exact_shapes: True — jagged outputs are sliced to their exact total length (one small host read per step,
                 objects identical in shape to the reference's); False — sync-free: jagged outputs are persistent
                 views over capacity buffers with explicit lengths.
            dispatch: how the stream-ordered launches reach libfrz_hip.so: 'ctypes' (default; torch-free C-ABI calls) or 'torch'
                 (the same entry points as PyTorch custom ops, ``torch.ops.frz.<domain>_<entry>``: csrc/torch_ops).  Same kernels, same
                 results; the environment variable FRZ_DISPATCH sets the default.
        """
        device = torch.device(device)
        if device.type != 'cuda':
            raise ValueError('free_range_zoo_amd environments run on a GPU device only (device="cuda"); there is no CPU path')
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if rng not in ('mt19937', 'philox'):
            raise ValueError("rng must be 'mt19937' or 'philox'")
        self.parallel_envs = parallel_envs
        self.max_steps = max_steps
        self.device = device
        self.render_mode = render_mode
        self.log_directory = log_directory
        self.single_seeding = single_seeding
        self.buffer_size = buffer_size
        self.rng = rng
        self.exact_shapes = exact_shapes
        self.log_description = None
        self.logger = None
        if log_directory is not None:  # env.py:65-85; rows are written by a background thread, off the step path
            from free_range_zoo_amd.utils.logging_handlers import CSVLogger, SQLLogger
            if log_directory.startswith(('sqlite://', 'postgresql://')):
                self.logger = SQLLogger(connection_string=log_directory, domain=f'{self._domain}_v0', parallel_envs=parallel_envs)
            else:
                self.logger = CSVLogger(log_directory=log_directory, parallel_envs=parallel_envs,
                                        override_initialization_check=override_initialization_check)
        if configuration is not None:
            self.config = configuration.to(device)
            for key, value in vars(configuration).items():  # nested configurations become attributes (env.py:58-63)
                if isinstance(value, Configuration):
                    setattr(self, key, value)
        self.generator = RandomGenerator(parallel_envs=parallel_envs, buffer_size=buffer_size, single_seeding=single_seeding,
                                         device=device)
        self._lib = _capi.lib()
        self._handle = None
        self._has_reset = False
        import os
        self._dispatch = dispatch or os.environ.get('FRZ_DISPATCH', 'ctypes')
        if self._dispatch not in ('ctypes', 'torch'):
            raise ValueError("dispatch must be 'ctypes' or 'torch'")
        self._ops = None
        if self._dispatch == 'torch':
            from free_range_zoo_amd import _torch_ops
            self._ops = _torch_ops.load()

    # the reference exposes the AEC env under the parallel wrapper; one object plays both roles here
    @property
    def aec_env(self):
        return self

    @property
    def unwrapped(self):
        return self

    @property
    def num_agents(self) -> int:
        return len(self.agents)

    @property
    def max_num_agents(self) -> int:
        return len(self.possible_agents)

    def _alloc(self, shape, dtype) -> torch.Tensor:
        return torch.zeros(shape, dtype=dtype, device=self.device)

    _domain: str = ''  # 'wildfire' | 'cybersecurity' | 'rideshare': the C entry points are frz_<domain>_<entry>

    # -- published outputs ----------------------------------------------------------------------------------------------------------
    # What a step leaves for the caller besides the dense per-agent dicts: the observations (TensorDicts with jagged `tasks`), the task
    # store and the jagged action / observation / bad-action mappings.  With exact_shapes (the default) their construction needs the list
    # lengths on the host, so `_publish()` only marks them stale and the first access builds them (`_materialize()` of the domain, one
    # small device read); with exact_shapes=False they are persistent views over capacity buffers, built once.
    _LAZY_OUTPUTS = frozenset(('observations', 'task_store', 'agent_action_mapping', 'agent_observation_mapping', 'agent_bad_actions'))

    def _publish(self) -> None:
        self._bump_space_epoch()
        if self.exact_shapes:
            for name in self._LAZY_OUTPUTS:
                self.__dict__.pop(name, None)
        else:
            self._materialize()
        if getattr(self, 'rewards', None) is None:
            self._publish_dense()

    def __getattr__(self, name):
        # only reached when normal lookup fails: a stale published output is rebuilt on demand
        if name in BatchedParallelEnv._LAZY_OUTPUTS and self.__dict__.get('_has_reset'):
            self._flush()
            self._materialize()
            return self.__dict__[name]
        raise AttributeError(f'{type(self).__name__!r} object has no attribute {name!r}')

    def _observations_out(self):
        """What step() / reset() return as observations: filled on first use in the exact-shapes mode, the persistent views otherwise."""
        if self.exact_shapes:
            return LazyAgentDict(self, self.agents)
        return {agent: self.observations[agent] for agent in self.agents}

    # -- deferred steps ------------------------------------------------------------------------------------------------------------
    # With the device declared exclusive (set_exclusive_device) the library can run n steps as ONE launch that keeps the envs in registers
    # (6.7 us per step inside it against ~10 us per single-step launch).  The reference-shaped random loop —
    # `env.step({agent: env.action_space(agent).sample_nested() ...})` with the samples untouched — gives the env everything a step needs
    # without looking at anything the step produces, so `step()` only COUNTS such a step; the pending steps run as one launch when the
    # count reaches the current chunk size, or as soon as anything is looked at: every tensor the env hands out in this mode is an
    # `EnvTensor` / `LazySample` / `LazyAgentDict`, and every method that reads or writes device state starts with `_flush()`.
    _deferred: int = 0          # steps counted and not launched yet
    _deferred_first: int = 0    # policy step (draw index) of the first of them
    _deferred_seed: int = 0     # policy seed they were counted under
    _defer_chunk: int = 0       # 0: deferral off; else the number of pending steps that triggers a launch (doubles from _DEFER_MIN to _DEFER_MAX while nobody looks)
    _DEFER_MIN, _DEFER_MAX = 4, 32

    def _flush(self) -> None:
        """Launch the pending steps (one multi-step launch; a single pending step: the ordinary step launch)."""
        n = self._deferred
        if n:
            self._deferred = 0
            self._launch_deferred(n, self._deferred_first, self._deferred_seed)

    # what the fast path of the reference-shaped random loop needs from a domain: how its fused policy + step entry is called
    # (`_single_fused_args`: what follows (handle, policy seed, policy step, sample buffer)), whether `frz_<domain>_rollout` can run several
    # such steps as ONE launch (`_deferral_possible`), and what `step()` returns besides the dense dicts (`_step_infos`)
    def _single_fused_args(self, mode: int) -> tuple:
        return (mode, None, None)

    def _fused_mode_or_none(self) -> Optional[int]:
        """RNG mode of the fused policy + step entry, or None where the draws must go through the generator API (single_seeding, buffered
        draws: the samples are then made by the policy launch and the step takes them as given actions)."""
        if self.rng == 'mt19937' and (self.single_seeding or self.generator.buffer_size):
            return None
        return self._fused_rng_mode()

    def _step_infos(self) -> dict:
        return {agent: {} for agent in self.agents}

    def _try_fast_step(self, actions):
        """`step(actions)` of the reference-shaped random loop — `actions` is exactly this step's untouched
        `{agent: env.action_space(agent).sample_nested()}` — with the samples drawn INSIDE the step launch (one launch per step), or, once the
        device was declared exclusive, only counted (deferred steps, above).  None: not that case, `step()` goes on as usual."""
        if type(actions) is not dict or self.__dict__.get('_pending_samples') is None:
            return None
        draw = self._untouched_samples(actions)
        if draw is None:
            return None
        if self._fused_mode_or_none() is None:
            self._pending_samples[3] = False  # (drawn by the policy launch when step() stages them)
            return None
        if self._ops is not None:
            mode = self._fused_rng_mode()
            self._call('step_random_policy', (), lambda: (self.policy_seed, draw, self._sampled_actions, *self._fused_ops_tail(mode)))
            return self._after_fused(False)
        chunk = self._defer_chunk
        if chunk:  # the device is this env's alone: the step is counted, not launched
            n = self._deferred
            if n and (self._deferred_first + n != draw or self._deferred_seed != self.policy_seed):
                self._flush()
                n = 0
            if n == 0:
                self._deferred_first, self._deferred_seed = draw, self.policy_seed
            self._deferred = n + 1
            if n + 1 >= chunk:  # nobody looked for a whole chunk: launch it, make the next one longer
                self._flush()
                self._defer_chunk = min(2 * chunk, self._DEFER_MAX)
            return self._after_fast_step()
        self._launch_deferred(1, draw, self.policy_seed)
        return self._after_fast_step()

    def _fused_ops_tail(self, mode: int) -> tuple:
        """What follows (arena, handle, policy seed, policy step, sample buffer) in `torch.ops.frz.<domain>_step_random_policy`."""
        return (mode, len(self.agents), self.parallel_envs)

    def _launch_deferred(self, n: int, first: int, seed: int) -> None:
        """`n` counted steps of the reference-shaped random loop (policy steps first .. first + n - 1, drawn inside the launch into the
        sample buffer, which afterwards holds the last step's draw — what n single-step launches leave): ONE multi-step launch through
        `frz_<domain>_rollout`; a single step: the fused policy + step entry."""
        launcher = self.__dict__.get('_deferred_launcher')
        if launcher is None:
            spec = _capi.frz_rollout_spec()
            mode = self._fused_rng_mode()
            spec.rng_mode, spec.actions_out = mode, self._sampled_actions.data_ptr()
            launcher = self._deferred_launcher = (spec, ctypes.byref(spec), getattr(self._lib, f'frz_{self._domain}_rollout'),
                                                  getattr(self._lib, f'frz_{self._domain}_step_random_policy'), self._sampled_actions.data_ptr(),
                                                  self.device.index, self._single_fused_args(mode), mode)
        spec, ref, rollout, single, samples, index, tail, mode = launcher
        if mode == _capi.FRZ_RNG_MT19937:
            self.generator._ensure_streams()  # (a reset with new seeds since the last step: the streams are expanded again)
        stream = torch._C._cuda_getCurrentRawStream(index)
        if n == 1:
            code = single(self._handle, seed, first, samples, *tail, stream)
        else:
            spec.n_steps, spec.policy_seed, spec.first_step = n, seed, first
            code = rollout(self._handle, ref, stream)
        log = self.__dict__.get('_deferred_log')
        if log is not None:  # (tests, bench: the chunk sizes that were launched)
            log.append(n)
        if code:
            _capi.check(code, f'frz_{self._domain}_rollout (counted steps)')

    def _after_fast_step(self):
        """What `step()` returns on the fast path: the publication of the exact-shapes default inlined (nothing is read from the device)."""
        d = self.__dict__
        if d.get('_global_group', False) is not False:
            self._exchange_batch_totals()
        self._epoch_counter += 1
        if self.exact_shapes:
            if 'observations' in d or 'task_store' in d:  # (somebody looked at the last step's outputs: they are stale now)
                for name in self._LAZY_OUTPUTS:
                    d.pop(name, None)
            observations = LazyAgentDict(self, self.agents)
        else:
            self._materialize()
            observations = {agent: self.observations[agent] for agent in self.agents}
        self.infos = infos = self._step_infos()
        return (observations, self.rewards, self.terminations, self.truncations, infos)

    def _exclusive_if_forced(self) -> None:
        """Test hook: FRZ_FORCE_EXCLUSIVE=1 declares every env's device exclusive at construction, so that a whole test suite runs with counted
        steps and multi-step launches wherever the library has them (whatever a test then looks at must be what the eager path leaves)."""
        import os
        if os.environ.get('FRZ_FORCE_EXCLUSIVE') == '1':
            self.set_exclusive_device(True)

    def _after_fused(self, logged: bool):
        """What a fused policy + step call returns (domains with extra infos override `_step_infos`)."""
        self._publish()
        self.infos = self._step_infos()
        if logged:
            self._log_environment()
        return (self._observations_out(), self.rewards, self.terminations, self.truncations, self.infos)

    def _enable_deferral(self, accepted: bool, defer_steps: bool) -> None:
        """After `frz_<domain>_set_exclusive_device`: counted steps need a multi-step launch for the shape, the exact-shapes publication
        (observations are built when looked at), the ctypes dispatch, no logging tap (it reads every step), tensors handed out as
        EnvTensors, and the per-env device streams in the MT19937 mode."""
        self._defer_chunk = 0
        if not (accepted and defer_steps and self._hands_out_lazy and self.exact_shapes and self._ops is None and self.logger is None):
            return
        if self.rng == 'mt19937' and (self.single_seeding or self.generator.buffer_size):
            return
        mode = _capi.FRZ_RNG_MT19937 if self.rng == 'mt19937' else _capi.FRZ_RNG_PHILOX
        if getattr(self._lib, f'frz_{self._domain}_rollout_launches')(self._handle, 2, mode) == 1:
            self._defer_chunk = self._DEFER_MIN

    def _lazy(self, view: torch.Tensor) -> torch.Tensor:
        """`view` as the env hands it out: an EnvTensor where steps may be deferred."""
        return EnvTensor(view, self) if self._hands_out_lazy else view

    _hands_out_lazy: bool = False  # set by the domains whose step() can defer (before their views are made)

    def _call(self, entry: str, c_args=(), op_args=None) -> None:
        """One stream-ordered launch of ``frz_<domain>_<entry>`` on the current stream of the env's device: a ctypes call into
        libfrz_hip.so (``c_args`` = what follows the handle, the stream is appended), or — ``dispatch='torch'`` — the PyTorch custom op
        ``torch.ops.frz.<domain>_<entry>(arena, handle, *op_args())``, which takes its stream from the dispatcher's current-stream state."""
        if self._ops is not None:
            getattr(self._ops, f'{self._domain}_{entry}')(self._arena, self._handle.value, *(op_args() if op_args else ()))
        else:
            entries = self.__dict__.get('_entries')
            if entries is None:
                entries = self._entries = {}
            fn = entries.get(entry)
            if fn is None:
                fn = entries[entry] = getattr(self._lib, f'frz_{self._domain}_{entry}')
            code = fn(self._handle, *c_args, torch._C._cuda_getCurrentRawStream(self.device.index))
            if code:
                _capi.check(code, f'frz_{self._domain}_{entry}')

    def _host_read(self, stats: torch.Tensor) -> list:
        """(pending steps have run: every caller flushes first.)  The one small device->host read of an exact-shapes publication: ``stats`` (int64 list lengths) and, in the same copy, the
        device error word — a prefix hand-off that timed out (FRZ_ERR_SCAN_TIMEOUT: another stream or process kept part of a launch
        from becoming resident) means the jagged offsets about to be used are wrong, so it is raised here rather than left for
        ``check()``.  Invalid-action bits stay for ``check()`` (sync-free contract of ``step``)."""
        values = torch.cat([stats.to(torch.int64), self._error_flags.to(torch.int64)]).tolist()
        if values[-1] & _capi.DEFINES['FRZ_ERR_SCAN_TIMEOUT']:
            raise RuntimeError('device-side prefix hand-off timed out (FRZ_ERR_SCAN_TIMEOUT): the jagged outputs of this step are invalid; '
                               'is another stream or process occupying the GPU?')
        return values[:-1]

    def _check_errors(self) -> None:
        """Raise the data-dependent errors the kernels flagged (reads one word from the device)."""
        self._flush()
        flags = int(self._error_flags.item())
        if flags:
            self._error_flags.zero_()
            names = [k for k, v in _capi.DEFINES.items() if k.startswith('FRZ_ERR_') and flags & v]
            raise ValueError(f'invalid actions / internal error flagged by the device: {names}')

    def check(self) -> None:
        """Explicitly surface device-side error flags (the fast path never synchronises)."""
        self._check_errors()

    def _reset_options(self, options: Optional[Dict[str, Any]]) -> None:
        if options is not None and options.get('max_steps') is not None:
            self._set_max_steps(options['max_steps'])
        self._log_label = options.get('log_label') if options else None
        self.log_description = options.get('log_description') if options and options.get('log_description') else None
        if self.logger is not None:  # env.py:140-143: the logger starts new files on every reset
            self.logger.reset(log_label=self._log_label, log_description=self.log_description, agents=self.possible_agents)

    def _stage_actions(self, actions: Dict[str, torch.Tensor]) -> None:
        """``{agent: [B, 2]}`` -> the stacked int32 ``[A, B, 2]`` buffer the kernels read: one launch when the tensors already are
        int32 on the env's device, a converting copy per agent otherwise."""
        parts = [actions[agent] for agent in self.agents]
        if all(p.dtype == torch.int32 and p.device == self.device and tuple(p.shape) == (self.parallel_envs, 2) for p in parts):
            torch.stack(parts, out=self._actions)
        else:
            for a, part in enumerate(parts):
                self._actions[a].copy_(part)

    # -- logging tap (env.py:191-195, 239-240, 256-271) ---------------------------------------------------------------
    def _log_extra(self, reset: bool):
        """Per-domain extra columns (dict of per-env columns), None when the domain adds none."""
        return None

    def _logs_this_step(self) -> bool:
        """The reference logs a step only when it really steps: not once every env is finished (env.py:211-213).  Reads one word
        from the device — only when logging is on."""
        return self.logger is not None and not bool(torch.all(self.finished))

    def _log_environment(self, reset: bool = False) -> None:
        self.logger.log_environment(state=self.state(), actions=self.actions, rewards=self.rewards,
                                    agent_action_mapping=self.agent_action_mapping,
                                    agent_observation_mapping=self.agent_observation_mapping, num_moves=self.num_moves,
                                    finished=None if reset else self.finished, log_description=self.log_description,
                                    agents=self.possible_agents, extra=self._log_extra(reset), reset=reset)

    # -- action_space(agent).sample_nested() through the policy kernel -----------------------------------------------------------
    policy_seed: int = 0x5EED  # stream of the spaces' device-side sampler; set it for reproducible `sample_nested()` rollouts

    def _space_sampler(self, agent_index: int):
        """``() -> int32 [B, 2]`` for ``BatchedOneOfSpace.sample_nested``: the agent's slice of the env's sample buffer as a
        ``LazySample``.  One draw (policy step ``_sampled_draws``) serves all agents of a step; it is made by the domain's policy kernel
        when a sample is first looked at, or inside the step launch when the samples go to ``step`` untouched."""

        def sample() -> torch.Tensor:
            pending = self.__dict__.get('_pending_samples')
            if pending is None:  # the sample buffer and its per-agent LazySample views: made once, handed out again every step
                self._sampled_actions = torch.zeros_like(self._actions)
                views = tuple(LazySample(self._sampled_actions[a], self) for a in range(len(self.possible_agents)))
                pending = self._pending_samples = [-1, -1, views, True]  # epoch of the draw, its policy step, per-agent tensors, drawn
            if pending[0] != self._epoch_counter:  # first sample of this step: a new draw is due
                pending[0] = self._epoch_counter
                pending[1] += 1
                pending[3] = False
            return pending[2][agent_index]

        return sample

    def _materialize_samples(self) -> None:
        """The policy launch behind a ``LazySample`` that is being looked at (once per draw; after its step was taken the buffer holds
        what that step drew)."""
        self._flush()  # (a sample that a counted step is about to draw: the step runs, the buffer then holds what it drew)
        pending = self.__dict__.get('_pending_samples')
        if pending is not None and not pending[3]:
            pending[3] = True
            self.random_policy_actions(self.policy_seed, pending[1], out=self._sampled_actions)

    def _untouched_samples(self, actions) -> Optional[int]:
        """The policy step of the pending draw when ``actions`` is exactly ``{agent: action_space(agent).sample_nested()}`` of the current
        step and nobody has looked at the samples — ``step`` then draws them inside its own launch — else None."""
        pending = self.__dict__.get('_pending_samples')
        if pending is None or pending[3] or pending[0] != self._epoch_counter or self.logger is not None or len(actions) != len(pending[2]):
            return None
        views = pending[2]
        for a, agent in enumerate(self.agents):
            if actions.get(agent) is not views[a]:
                return None
        pending[3] = True  # the step launch makes the draw
        return pending[1]

    _epoch_counter: int = 0  # bumped by reset / step / rebuild: the task counts changed, samples drawn before are stale

    @property
    def _space_epoch(self) -> int:
        return self._epoch_counter

    def _bump_space_epoch(self) -> None:
        self._epoch_counter += 1

    # -- rollouts: n steps of a rollout loop enqueued by one call (include/frz.h: frz_rollout_spec) ---------------------------------
    @torch.no_grad()
    def rollout(self, steps: int, actions: Optional[torch.Tensor] = None, randomness=None, policy_seed: int = 0, first_step: int = 0,
                reset_first: bool = False, seed_increment: int = 0, auto_reset: bool = False, seed_stride: int = 0,
                record: bool = False, metrics: Optional[torch.Tensor] = None, record_observations: Optional[str] = None,
                record_state: bool = False) -> Dict[str, Any]:
        """
        ``steps`` x ``step`` as ONE call through the C boundary (``frz_<domain>_rollout``) — one multi-step launch where the library has one
        for the shape (after ``set_exclusive_device``), otherwise one launch per step — with the results a loop over ``step()`` leaves.

        Args:
            actions: an ACTION TAPE, int32 ``[steps, A, B, 2]`` on the env's device (what the loop would pass to ``step`` one by one: the
                reference's recorded trajectories, a scripted policy's plan); None: the device-side uniform random policy
                (``policy_seed``, policy steps ``first_step ...``), as ``step_random_policy``.
            randomness: injected randomness tapes ``(a, b)`` — the tensors the env's generator would return, one pair per step
                (wildfire: ``[steps, 3, B, H, W]`` and ``[steps, 5, B, A]``); None: the env's own ``rng``.
            reset_first / seed_increment: the rollout starts with ``reset`` of every env (seeds += seed_increment), inside the same launch.
            auto_reset / seed_stride: an env that finishes at step t is reset (seed += seed_stride) inside step t — `step()` followed by
                `reset_batches(finished)`: continuous rollouts at fixed B.  The reset zeroes that env's rewards / flags like the
                reference's; what the step produced is in the ``record`` tapes and the ``metrics``.
            record: keep EVERY step's outputs — returns ``{'rewards': [steps, A, B], 'dones': [steps, 2, B], 'actions': [steps, A, B, 2]
                (policy), 'lists': uint8 [steps - 1, block_bytes]}`` (``lists[t]`` = a copy of the env's packed-list block after step t).
            metrics: float64 ``[A + 2]``, accumulated in place (see ``frz_rollout_spec.metrics``).
            record_observations: keep every step's observations (what the reference's loop hands back at every step, utils/conversions.py:92-99):
                ``'full'`` — ``out['observations']`` = uint8 ``[steps, obs_block_bytes]``, copies of the dense observation block;
                ``'compact'`` (wildfire) — float32 ``[steps, A, B]``, each agent's suppressant, the one column of the self / others records a
                step changes.  ``recorded_observations(out, t)`` rebuilds step t's ``{agent: TensorDict}`` from either (the jagged ``tasks``
                come from the list record: use ``record=True`` with it).
            record_state: ``out['state']`` = uint8 ``[steps, state_block_bytes]``: what ``env.state()`` shows after each step
                (``recorded_state(out, t)`` wraps step t as the domain's State).
        """
        if not self._has_reset:
            raise RuntimeError('reset() must be called before rollout()')
        self._flush()
        if self.__dict__.get('_global_group', False) is not False and steps > 1:
            raise NotImplementedError('rollout() enqueues several steps without the per-step exchange of set_global_consistency()')
        if self.logger is not None:
            raise NotImplementedError('rollout() does not feed the logging tap: step() does')
        A, B = len(self.agents), self.parallel_envs
        spec = _capi.frz_rollout_spec()
        spec.n_steps = int(steps)
        spec.flags = (_capi.FRZ_ROLLOUT_RESET_FIRST if reset_first else 0) | (_capi.FRZ_ROLLOUT_AUTO_RESET if auto_reset else 0)
        spec.seed_increment, spec.seed_stride = int(seed_increment), int(seed_stride) & 0xFFFFFFFF
        spec.policy_seed, spec.first_step = int(policy_seed), int(first_step)
        keep = []
        out: Dict[str, Any] = {}
        if actions is not None:
            if actions.dtype != torch.int32 or not actions.is_contiguous() or tuple(actions.shape) != (steps, A, B, 2) or actions.device != self.device:
                raise ValueError('the action tape must be a contiguous int32 [steps, A, B, 2] tensor on the env device')
            spec.action_tape = actions.data_ptr()
            keep.append(actions)
        else:
            if record:
                out['actions'] = torch.zeros((steps, A, B, 2), dtype=torch.int32, device=self.device)
                spec.actions_out, spec.record_actions = out['actions'].data_ptr(), 1
            else:
                spec.actions_out = self._actions.data_ptr()
        if randomness is not None:
            a, b = (t.to(device=self.device, dtype=torch.float32).contiguous() for t in randomness)
            self._check_randomness_tapes(steps, a, b)
            spec.rng_mode, spec.randomness_tape_a, spec.randomness_tape_b = _capi.FRZ_RNG_INJECTED, a.data_ptr(), b.data_ptr()
            keep += [a, b]
        else:
            spec.rng_mode = self._fused_rng_mode()
        if record:
            out['rewards'] = torch.zeros((steps, A, B), dtype=torch.float32, device=self.device)
            out['dones'] = torch.zeros((steps, 2, B), dtype=torch.uint8, device=self.device)
            spec.reward_tape, spec.done_tape = out['rewards'].data_ptr(), out['dones'].data_ptr()
            block, nbytes = ctypes.c_void_p(), ctypes.c_int64()
            _capi.check(getattr(self._lib, f'frz_{self._domain}_list_block')(self._handle, ctypes.byref(block), ctypes.byref(nbytes)), 'list_block')
            out['lists'] = torch.zeros((max(steps - 1, 0), nbytes.value), dtype=torch.uint8, device=self.device)
            out['list_block_offset'] = block.value - self._arena.data_ptr()
            spec.list_record = out['lists'].data_ptr() if steps > 1 else None
        if metrics is not None:
            if metrics.dtype != torch.float64 or metrics.numel() != A + 2 or not metrics.is_contiguous() or metrics.device != self.device:
                raise ValueError('metrics must be a contiguous float64 [A + 2] tensor on the env device')
            spec.metrics = metrics.data_ptr()
        obs_tape = state_tape = None
        if record_observations is not None:
            if record_observations not in ('full', 'compact'):
                raise ValueError("record_observations must be None, 'full' or 'compact'")
            if record_observations == 'compact':
                if self._domain != 'wildfire':
                    raise ValueError("record_observations='compact' exists for wildfire (the only column a step changes there is the suppressant)")
                spec.flags |= _capi.FRZ_ROLLOUT_OBS_COMPACT
                out['observations'] = torch.zeros((steps, A, B), dtype=torch.float32, device=self.device)
            else:
                out['observations'] = torch.zeros((steps, self._block_bytes('obs')), dtype=torch.uint8, device=self.device)
            out['observations_form'] = record_observations
            obs_tape = out['observations']
            spec.obs_tape = obs_tape.data_ptr()
        if record_state:
            state_tape = out['state'] = torch.zeros((steps, self._block_bytes('state')), dtype=torch.uint8, device=self.device)
            spec.state_tape = state_tape.data_ptr()
        if self._ops is not None and hasattr(self._ops, f'{self._domain}_rollout'):  # the same call as a dispatcher-visible op: every tape is an argument
            tapes = randomness if randomness is None else tuple(keep[-2:])
            getattr(self._ops, f'{self._domain}_rollout')(
                self._arena, self._handle.value, int(steps), int(spec.rng_mode), int(spec.flags),
                int(seed_increment), int(seed_stride) & 0xFFFFFFFF, int(policy_seed), int(first_step), actions, None if tapes is None else tapes[0],
                None if tapes is None else tapes[1], None if actions is not None else (out['actions'] if record else self._actions), bool(record and actions is None),
                out.get('rewards'), out.get('dones'), out['lists'] if (record and steps > 1) else None, metrics,
                None if obs_tape is None else obs_tape.view(torch.uint8).reshape(-1), None if state_tape is None else state_tape.reshape(-1))
        else:
            symbol = f'frz_{self._domain}_rollout'
            _capi.check(getattr(self._lib, symbol)(self._handle, ctypes.byref(spec), stream_ptr(self.device)), symbol)
        self._rollout_keepalive = keep  # the launch reads the tapes after this call returns
        self._after_rollout()
        return out

    def _check_randomness_tapes(self, steps: int, a: torch.Tensor, b: torch.Tensor) -> None:
        raise NotImplementedError

    def _block_bytes(self, which: str) -> int:
        """Size in bytes of the dense observation block / the state block of the env's arena (include/frz.h: frz_<domain>_obs_block,
        frz_<domain>_state_block): one step of an observation / state tape."""
        raise NotImplementedError(f'{self._domain}_v0 rollouts keep no {which} tape')

    def _after_rollout(self) -> None:
        self._publish()

    @torch.no_grad()
    def reset_finished(self, mask: Optional[torch.Tensor] = None, seed_increment: int = 0) -> None:
        """``reset_batches`` with the selection left on the device (utils/env.py:162-189): envs with ``mask[b] != 0`` (uint8 / bool ``[B]``) —
        or, ``mask=None``, the finished ones — go back to their initial state, bookkeeping zeroed, ``seeds += seed_increment``; no host
        read, no indices on the host.  (The MT19937 streams of the reset envs are re-seeded from the new seeds.)"""
        if not self._has_reset:
            raise RuntimeError('reset() must be called before reset_finished()')
        self._flush()
        symbol = f'frz_{self._domain}_reset_masked'
        if symbol not in _capi.SIGNATURES:  # (the library has the device-mask partial reset for wildfire only)
            raise NotImplementedError(f'{self._domain}_v0 has no device-side partial reset (no {symbol} in include/frz.h): use reset_batches(indices)')
        if mask is not None:
            mask = mask.to(device=self.device).view(torch.uint8) if mask.dtype == torch.bool else mask.to(device=self.device, dtype=torch.uint8)
            mask = mask.contiguous()
            if mask.numel() != self.parallel_envs:
                raise ValueError('mask must have one entry per env')
        if self.rng == 'mt19937':
            selected = (mask != 0) if mask is not None else self.finished
            self._mt_reseed_mask = selected.clone()
        _capi.check(getattr(self._lib, symbol)(self._handle, None if mask is None else mask.data_ptr(), int(seed_increment), stream_ptr(self.device)), symbol)
        if self.rng == 'mt19937' and not self.single_seeding:
            self.generator.reseed_where(self._mt_reseed_mask)
        self._publish()

    # -- refreshing the published outputs after the state was edited in place (planning / search code) -----------------------------
    _rebuild_symbol: Optional[str] = None  # set by the domain envs: the C-ABI entry that rebuilds task lists, observations, spaces

    @torch.no_grad()
    def update_observations(self) -> None:
        """Rebuild observations, task lists and action mappings from the CURRENT state (the reference's hooks of the same names,
        e.g. wildfire.py:584-700, called by code that edits ``env.state()`` between steps): one launch, then re-publish."""
        if not self._has_reset:
            raise RuntimeError('reset() must be called before update_observations()')
        self._flush()
        self._call('rebuild')
        self._publish()

    def refresh(self) -> None:
        """Publish what the buffers hold NOW (after ``graph.replay()`` of a captured rollout, which moves the env without going through
        ``step()``): the exact-shapes outputs are rebuilt on their next access, spaces and samples take a new epoch.  No launch."""
        self._flush()
        self._publish()

    def update_actions(self) -> None:
        """The same launch rebuilds both (update_observations + update_actions of the reference are one kernel here)."""
        self.update_observations()

    def observe(self, agent: Optional[str] = None):
        """``observe(agent)`` (AEC, env.py:274-284) or ``observe()`` -> dict of all agents (adapter, conversions.py:101-108)."""
        if agent is None:
            return {name: self.observations[name] for name in self.agents}
        return self.observations[agent]

    def state(self):
        self._flush()
        return self._state

    # `terminated` / `truncated` = all agents' flags (utils/env.py:331-359 of the reference): one reduction over the [A][B] flag block each
    # (the reference stacks the per-agent tensors first)
    @property
    def terminated(self) -> torch.Tensor:
        self._flush()
        return self._terminations.all(dim=0)

    @property
    def truncated(self) -> torch.Tensor:
        self._flush()
        return self._truncations.all(dim=0)

    @property
    def finished(self) -> torch.Tensor:
        self._flush()
        return torch.logical_or(self._terminations.all(dim=0), self._truncations.all(dim=0))

    def close(self) -> None:
        """Drain the logging tap (every row handed over so far reaches its file)."""
        if self.logger is not None:
            self.logger.close()


# The reference's base class name (utils/env.py:18): one class plays the AEC env and the parallel adapter here.
BatchedAECEnv = BatchedParallelEnv
