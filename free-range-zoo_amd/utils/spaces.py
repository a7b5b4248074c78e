"""Count-based batched action spaces.

The reference builds, per env, a ``free_range_rust.Space.OneOf([Discrete(1, start=s_t) for task t] + task-agnostic
actions)`` (envs/*/env/spaces/actions.py) and callers sample it with ``Space.Vector(...).sample_nested()`` ->
``[[member_index, member_value], ...]``.  free-range-rust is a native third-party dependency that is not part of the
reference tree; this module gives the same *structure* (per-env member lists) computed from the task counts the HIP
kernels maintain, with sampling done on the device.  Sampling distribution: uniform over the OneOf members (parity of
the Rust sampler's stream is unpinned).
"""
from typing import Callable, Dict as DictType, List, Sequence

import torch


class Discrete:
    def __init__(self, n: int, start: int = 0):
        self.n, self.start = n, start

    def __eq__(self, other):
        return isinstance(other, Discrete) and (self.n, self.start) == (other.n, other.start)

    def __repr__(self):
        return f'Discrete({self.n}, start={self.start})'


class OneOf:
    def __init__(self, spaces: Sequence[Discrete]):
        self.spaces = list(spaces)

    def __len__(self):
        return len(self.spaces)

    def __eq__(self, other):
        return isinstance(other, OneOf) and self.spaces == other.spaces

    def __repr__(self):
        return f'OneOf({self.spaces})'


def _bound(value):
    """A box bound as the reference's builders hand it to ``free_range_rust.Space.Box``: a plain number — integral values as int, others as
    float (``max_fire_reduction_power`` may be 1.5) — or None where the configuration leaves it open (``max_steps=None``,
    ``suppressant_states=None``)."""
    if value is None:
        return None
    value = value.item() if hasattr(value, 'item') else value
    as_float = float(value)
    if as_float != as_float or as_float in (float('inf'), float('-inf')):  # an open bound given as inf / nan stays a float
        return as_float
    return int(value) if as_float == int(value) else as_float


def bounds(values) -> tuple:
    """Observation bounds as a hashable tuple of plain numbers / None (the builders are lru-cached on them)."""
    return tuple(_bound(v) for v in values)


class Box:
    """Box ``[low, high]`` per dimension (``free_range_rust.Space.Box``)."""

    def __init__(self, low: Sequence, high: Sequence):
        self.low, self.high = [_bound(v) for v in low], [_bound(v) for v in high]
        if len(self.low) != len(self.high):
            raise ValueError('low and high must have the same length')

    def __len__(self):
        return len(self.high)

    def __eq__(self, other):
        return isinstance(other, Box) and (self.low, self.high) == (other.low, other.high)

    def __hash__(self):
        return hash((tuple(self.low), tuple(self.high)))

    def __repr__(self):
        return f'Box(low={self.low}, high={self.high})'


class Tuple:
    """Fixed sequence of spaces (``free_range_rust.Space.Tuple``)."""

    def __init__(self, spaces: Sequence):
        self.spaces = list(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, index):
        return self.spaces[index]

    def __eq__(self, other):
        return isinstance(other, Tuple) and self.spaces == other.spaces

    def __repr__(self):
        return f'Tuple({self.spaces})'


class Dict:
    """Named spaces (``free_range_rust.Space.Dict``)."""

    def __init__(self, spaces: DictType[str, object]):
        self.spaces = dict(spaces)

    def __getitem__(self, key):
        return self.spaces[key]

    def keys(self):
        return self.spaces.keys()

    def __eq__(self, other):
        return isinstance(other, Dict) and self.spaces == other.spaces

    def __repr__(self):
        return f'Dict({self.spaces})'


class Vector:
    """A list of per-env spaces (``free_range_rust.Space.Vector``)."""

    def __init__(self, spaces: Sequence):
        self.spaces = list(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, index):
        return self.spaces[index]

    def __eq__(self, other):
        other_spaces = other.spaces if isinstance(other, (Vector, BatchedOneOfSpace)) else other
        return self.spaces == other_spaces

    def __repr__(self):
        return f'Vector({self.spaces})'


class Space:
    """Constructors under the names the reference imports from ``free_range_rust`` (``Space.Box``, ``Space.OneOf``, ...): structure and
    equality only — the count-based batched objects below carry the per-env structure without materialising B Python objects."""
    Discrete = Discrete
    OneOf = OneOf
    Box = Box
    Tuple = Tuple
    Dict = Dict
    Vector = Vector


class BatchedSpace:
    """Per-env spaces that are a function of one integer per env (the task count): list-like over the env batch, entries built on demand
    by ``single(count)`` (cached by the builders), so handing one out costs no device read and no O(B) work.  ``observation_space(agent)``
    of the three domains returns one (wildfire.py:736-753, rideshare.py:489-504, cybersecurity.py:553-578)."""

    def __init__(self, task_counts: torch.Tensor, single: Callable[[int], object]):
        self.task_counts = task_counts  # a view of the env's buffer: describes the CURRENT step
        self._single = single
        self._host_counts = None

    def __len__(self):
        return int(self.task_counts.shape[0])

    def _counts(self) -> List[int]:
        if self._host_counts is None:
            self._host_counts = [int(v) for v in self.task_counts.tolist()]  # one device read, on first inspection
        return self._host_counts

    def __getitem__(self, index):
        if isinstance(index, slice):
            return [self._single(n) for n in self._counts()[index]]
        return self._single(self._counts()[index])

    def __iter__(self):
        return (self._single(n) for n in self._counts())

    @property
    def spaces(self) -> List:
        return list(self)

    def __eq__(self, other):
        other_spaces = other.spaces if isinstance(other, (Vector, BatchedSpace)) else other
        return self.spaces == list(other_spaces)

    def __repr__(self):
        return f'BatchedSpace({len(self)} envs)'


class BatchedOneOfSpace:
    """Vector of per-env ``OneOf`` spaces described by (task member starts, task-agnostic tail).

    A space handed out by ``env.action_space(agent)`` describes the env's CURRENT step: its counts are views of the env's buffers and
    its lazily built parts are resolved on first use, so it is valid until the next ``step()`` / ``reset()`` (take a new one then,
    as rollout code does)."""

    def __init__(self, task_counts: torch.Tensor, tail: Sequence[int], task_starts: torch.Tensor = None,
                 tail_mask: torch.Tensor = None, sampler=None, epoch=None):
        """
        task_counts: int tensor [B] — number of task members (Discrete(1, start=task_starts or 0)) per env
        tail:        values of the task-agnostic members appended after the tasks (e.g. [-1] = noop)
        task_starts: optional jagged start values per task member (rideshare: the passenger's state), padded [B, max]
        tail_mask:   optional bool [B, len(tail)] — which tail members exist per env
        """
        self.task_counts = task_counts
        self.tail = list(tail)
        self._task_starts = task_starts  # tensor, None, or a zero-argument callable resolved on first use (it may cost a host read)
        self._tail_mask = tail_mask  # tensor, None, or a zero-argument callable resolved on first use
        self.sampler = sampler  # optional () -> int32 [B, 2]: the env's device-side policy kernel (one launch for all agents)
        # optional () -> int: the env's step counter.  An env hands out ONE space object per agent for its whole life (the counts are views:
        # the object always describes the current step); the parts given as callables are then resolved again once the env has moved on
        self._epoch = epoch
        self._resolved = {}

    def __len__(self):
        return int(self.task_counts.shape[0])

    def _resolve(self, name: str):
        value = getattr(self, name)
        if not callable(value):
            return value
        if self._epoch is None:  # a space of one step (taken anew every step, as rollout code does): resolved once
            value = value()
            setattr(self, name, value)
            return value
        epoch = self._epoch()
        cached = self._resolved.get(name)
        if cached is None or cached[0] != epoch:
            cached = self._resolved[name] = (epoch, value())
        return cached[1]

    @property
    def tail_mask(self):
        return self._resolve('_tail_mask')

    @property
    def task_starts(self):
        return self._resolve('_task_starts')

    @property
    def spaces(self) -> List[OneOf]:
        """Materialise the per-env ``OneOf`` objects (host side, O(B); for inspection and tests)."""
        counts = self.task_counts.tolist()
        starts = self.task_starts.tolist() if self.task_starts is not None else None
        masks = self.tail_mask.tolist() if self.tail_mask is not None else None
        out = []
        for b, n in enumerate(counts):
            members = [Discrete(1, start=(starts[b][t] if starts is not None else 0)) for t in range(int(n))]
            for j, value in enumerate(self.tail):
                if masks is None or masks[b][j]:
                    members.append(Discrete(1, start=value))
            out.append(OneOf(members))
        return out

    def sample_nested(self, generator: torch.Generator = None) -> torch.Tensor:
        """Uniform member per env -> int32 ``[B, 2]`` = (member index, member value), on the counts' device.  Spaces handed out by an
        env sample through its policy kernel (one launch serves every agent of the step; made when a sample is first looked at, or inside
        the step launch: utils/env.py LazySample); an explicit ``generator`` selects the torch path."""
        if self.sampler is not None and generator is None:
            return self.sampler()
        return self._sample_with_torch(generator)

    @torch.no_grad()
    def _sample_with_torch(self, generator: torch.Generator = None) -> torch.Tensor:
        counts = self.task_counts.to(torch.int64)
        device = counts.device
        B = counts.shape[0]
        tail_values = torch.tensor(self.tail, dtype=torch.int64, device=device)
        if self.tail_mask is not None:
            n_tail = self.tail_mask.sum(dim=1).to(torch.int64)
        else:
            n_tail = torch.full((B, ), len(self.tail), dtype=torch.int64, device=device)
        total = counts + n_tail
        u = torch.rand((B, ), device=device, generator=generator)
        member = torch.minimum((u * total).to(torch.int64), total - 1)
        is_task = member < counts
        if self.task_starts is not None and self.task_starts.shape[1] > 0:
            idx = member.clamp(max=self.task_starts.shape[1] - 1).unsqueeze(1)
            task_value = self.task_starts.to(torch.int64).gather(1, idx).squeeze(1)
        else:
            task_value = torch.zeros_like(member)
        k = (member - counts).clamp(min=0)
        if self.tail_mask is not None:  # k-th existing tail member
            order = torch.cumsum(self.tail_mask.to(torch.int64), dim=1) - 1
            pick = ((order == k.unsqueeze(1)) & self.tail_mask).to(torch.int64).argmax(dim=1)
            tail_value = tail_values[pick]
        else:
            tail_value = tail_values[k.clamp(max=len(self.tail) - 1)]
        value = torch.where(is_task, task_value, tail_value)
        return torch.stack([member, value], dim=1).to(torch.int32)

    @torch.no_grad()
    def invalid_actions(self, actions: torch.Tensor, allow_flexible_task_tags: bool = True) -> torch.Tensor:
        """Which envs' ``[task_channel, action_channel]`` pairs the reference's validator rejects
        (wrappers/space_validator.py:47-83), as a bool ``[B]`` tensor computed on the actions' device.  As written there:
        a negative action channel, with ``allow_flexible_task_tags``, only has to equal the start of SOME member of the env's
        space; otherwise the task channel indexes the member list like a Python list (negative indices count from the end,
        out of range is invalid) and the action channel must lie in ``[start, start + n]`` of that member (inclusive upper end)."""
        actions = actions.to(torch.int64)
        device = actions.device
        counts = self.task_counts.to(device=device, dtype=torch.int64)
        B = counts.shape[0]
        task, act = actions[:, 0], actions[:, 1]
        tail_values = torch.tensor(self.tail, dtype=torch.int64, device=device)
        T = len(self.tail)
        mask = self.tail_mask.to(device) if self.tail_mask is not None else torch.ones((B, T), dtype=torch.bool, device=device)
        n_tail = mask.sum(dim=1).to(torch.int64)
        total = counts + n_tail
        starts = None
        if self.task_starts is not None and self.task_starts.shape[1] > 0:
            starts = self.task_starts.to(device=device, dtype=torch.int64)

        # member lookup: index like a Python list
        index = torch.where(task < 0, task + total, task)
        in_range = (task >= -total) & (task < total)
        safe = index.clamp(min=0)
        is_task = safe < counts
        if starts is not None:
            task_start = starts.gather(1, safe.clamp(max=starts.shape[1] - 1).unsqueeze(1)).squeeze(1)
        else:
            task_start = torch.zeros_like(safe)
        k = (safe - counts).clamp(min=0)
        order = torch.cumsum(mask.to(torch.int64), dim=1) - 1
        pick = ((order == k.unsqueeze(1)) & mask).to(torch.int64).argmax(dim=1) if T > 0 else torch.zeros_like(k)
        tail_start = tail_values[pick] if T > 0 else torch.zeros_like(k)
        start = torch.where(is_task, task_start, tail_start)
        by_member = ~in_range | (act < start) | (act > start + 1)

        if not allow_flexible_task_tags:
            return by_member
        # flexible: a negative action channel matches any member's start (task starts included)
        matches_tail = ((act.unsqueeze(1) == tail_values.unsqueeze(0)) & mask).any(dim=1) if T > 0 else torch.zeros(B, dtype=torch.bool, device=device)
        if starts is not None:
            live = torch.arange(starts.shape[1], device=device).unsqueeze(0) < counts.unsqueeze(1)
            matches_task = ((act.unsqueeze(1) == starts) & live).any(dim=1)
        else:
            matches_task = (act == 0) & (counts > 0)
        flexible = act < 0
        return torch.where(flexible, ~(matches_tail | matches_task), by_member)
