"""Per-environment random streams on the GPU (mirrors free_range_zoo/utils/random_generator.py:12-176).

The reference keeps one torch CPU generator state per env and loops over envs in Python for every draw.  Here the
same streams — MT19937 ``init_genrand(seed)``, float32 = (u32 & 0xFFFFFF) * 2**-24, env ``b`` drawing
``events * prod(shape)`` consecutive floats per call — live in HBM (``[624, B]`` words, env innermost) and are advanced
by the hand-written kernels of ``csrc/mt19937.hip``: the same seeds give bit-identical tensors to the reference's CPU path.

``single_seeding=True`` is the reference's one-shared-stream mode (random_generator.py:59-65, 103-106): ONE torch CPU
generator serves all envs, ``torch.rand((parallel_envs, events, *shape))`` per call.  A single sequential stream has no
batch axis to put on the GPU: it is drawn with torch's own CPU generator and uploaded (randomness as an input, the same
path as ``step(..., randomness=...)``).  As in the reference, the stream starts from a freshly constructed generator's
state: the seed value passed to ``seed()`` is stored in ``seeds`` but does not reach the stream.
"""
from typing import Optional, Tuple

import torch

from free_range_zoo_amd import _capi


def _stream_ptr(device) -> int:
    return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())


class RandomGenerator:
    """Random number generator for the environment (same public surface as the reference's class)."""

    def __init__(self, parallel_envs: int, buffer_size: int = 0, single_seeding: bool = False,
                 device: torch.device = torch.device('cuda')):
        device = torch.device(device)
        if device.type != 'cuda':
            raise ValueError(f'Device {device} not supported: the HIP kernels need a GPU device (no CPU fallback)')
        self._single = torch.Generator(device='cpu') if single_seeding else None
        self._single_state = None
        self.parallel_envs = parallel_envs
        self.buffer_size = buffer_size
        self.single_seeding = single_seeding
        self.device = device
        self.seeds = torch.empty((parallel_envs, ), dtype=torch.int32, device=device)
        self.generator_states = None  # [624, B] uint32 words, allocated on first use (or attached by the env)
        self.generator_index = torch.zeros((parallel_envs, ), dtype=torch.int32, device=device)
        self.buffer_count = {}
        self.buffers = {}
        self.has_been_seeded = False
        self._streams_valid = False  # MT19937 states match self.seeds (seeding 624 words per env is done lazily)
        self._pending_partial = None

    def attach(self, seeds: torch.Tensor, states: torch.Tensor, index: torch.Tensor) -> None:
        """Use caller-provided storage (views of an env's device arena) for seeds / MT19937 states / stream positions."""
        seeds.copy_(self.seeds)
        self.seeds, self.generator_states, self.generator_index = seeds, states, index
        self._streams_valid = False

    @torch.no_grad()
    def seed(self, seed: Optional[torch.Tensor] = None, partial_seeding: Optional[torch.Tensor] = None) -> None:
        """Seed all envs (or the envs listed in ``partial_seeding``); random seeds below 1e8 if ``seed`` is None."""
        if seed is None:
            shape = self.seeds.shape if partial_seeding is None else torch.as_tensor(partial_seeding).shape
            seed = torch.randint(100000000, shape, device=self.device)
        seed = self._to_device(seed)
        if self.single_seeding:  # random_generator.py:59-65: a fresh generator's state, whatever the seed
            self.seeds[:] = seed.reshape(-1)[0] if seed.numel() >= 1 else seed
            self._single_state = torch.Generator(device='cpu').get_state()
            self.has_been_seeded = True
            return
        if partial_seeding is None:
            self.seeds[:] = seed
            self._streams_valid = False  # all streams restart: expanded to MT19937 states when first drawn from
            self._pending_partial = None
        else:
            indices = torch.as_tensor(partial_seeding, device=self.device).to(torch.int32).contiguous().reshape(-1)
            self.seeds[indices.long()] = seed
            if self._streams_valid:
                self._seed_streams(indices)
        self.has_been_seeded = True

    def _to_device(self, seed) -> torch.Tensor:
        """``seed`` as an int32 tensor on the device.  Host values are staged in a pinned buffer with a plain ``memmove`` and uploaded
        asynchronously: reset() stays free of synchronisation points, and no torch CPU kernel runs on the way — a CPU tensor op of B
        elements goes through torch's intra-op thread pool (128 threads on the GPU box), whose spinning workers exhaust the CFS quota of a
        16-core cgroup within one 100 ms period; the kernel then freezes every thread of the process, the one enqueuing launches included,
        for the rest of the period (the 85-95 ms stalls of round 3's per-step API leg: tools/dbg/api_stall_probe.py)."""
        if isinstance(seed, torch.Tensor) and seed.device.type == 'cuda':
            return seed.to(device=self.device, dtype=torch.int32)
        host = torch.as_tensor(seed)
        if host.dtype != torch.int32 or not host.is_contiguous():
            host = host.to(torch.int32).contiguous()
        n = host.numel()
        staging = self.__dict__.get('_seed_staging')
        if staging is None or staging[0].numel() < n:
            staging = self._seed_staging = (torch.empty(max(n, 16), dtype=torch.int32, pin_memory=True), torch.cuda.Event())
        else:
            staging[1].synchronize()  # the previous upload out of this buffer has been consumed
        if n:
            import ctypes
            ctypes.memmove(staging[0].data_ptr(), host.data_ptr(), 4 * n)
        out = staging[0][:n].view(host.shape).to(self.device, non_blocking=True)
        staging[1].record(torch.cuda.current_stream(self.device))
        return out

    @torch.no_grad()
    def reseed_where(self, mask: torch.Tensor) -> None:
        """Restart the MT19937 streams of the envs ``mask`` selects from ``self.seeds`` (which a device-side partial reset has already
        moved on): the per-env part of ``seed(..., partial_seeding=...)`` for a selection that lives on the device.  Streams that have
        not been expanded yet are left to the lazy seeding."""
        if self.single_seeding or not self._streams_valid:
            return
        indices = mask.nonzero().reshape(-1).to(torch.int32)
        if indices.numel():
            self._seed_streams(indices.contiguous())

    def _seed_streams(self, indices: Optional[torch.Tensor]) -> None:
        if self.generator_states is None:
            self.generator_states = torch.empty((624, self.parallel_envs), dtype=torch.int32, device=self.device)
        indices_ptr, n = (None, self.parallel_envs) if indices is None else (indices.data_ptr(), indices.numel())
        _capi.check(_capi.lib().frz_mt19937_seed(self.generator_states.data_ptr(), self.generator_index.data_ptr(), self.seeds.data_ptr(),
                                                 indices_ptr, n, self.parallel_envs, _stream_ptr(self.device)), 'frz_mt19937_seed')

    def _ensure_streams(self) -> None:
        if self.single_seeding:
            raise NotImplementedError('single_seeding draws from one host-side stream: use step() (randomness is uploaded per call); '
                                      'the device-side MT19937 streams of graph-captured / fused rollouts are per env')
        if not self._streams_valid:
            self._seed_streams(None)
            self._streams_valid = True

    def _draw(self, events: int, count: int) -> torch.Tensor:
        self._ensure_streams()
        out = torch.empty((events, self.parallel_envs, count), dtype=torch.float32, device=self.device)
        _capi.check(_capi.lib().frz_mt19937_generate(self.generator_states.data_ptr(), self.generator_index.data_ptr(), out.data_ptr(),
                                                     events, count, self.parallel_envs, _stream_ptr(self.device)),
                    'frz_mt19937_generate')
        return out

    @torch.no_grad()
    def generate(self, parallel_envs: int, events: int, shape: Tuple[int], key: str = None) -> torch.Tensor:
        """Random tensor ``[events, parallel_envs, *shape]``; buffered per ``key`` when ``buffer_size > 0``."""
        if not self.has_been_seeded:
            raise ValueError('The environment must be seeded before generating randomness')
        if parallel_envs != self.parallel_envs:
            raise ValueError('parallel_envs does not match the generator')
        if self.single_seeding:
            return self._generate_single(parallel_envs, events, tuple(int(v) for v in shape), key)
        count = 1
        for s in shape:
            count *= int(s)
        if key is None or self.buffer_size == 0:
            return self._draw(events, count).reshape(events, parallel_envs, *shape)
        buffer_key = (key, (parallel_envs, events, *shape))
        if buffer_key not in self.buffers or self.buffer_count[buffer_key] >= self.buffer_size:
            # each env draws buffer_size*events*count consecutive floats (random_generator.py:133-138)
            block = self._draw(self.buffer_size * events, count)
            self.buffers[buffer_key] = block.reshape(self.buffer_size, events, parallel_envs, *shape)
            self.buffer_count[buffer_key] = 0
        out = self.buffers[buffer_key][self.buffer_count[buffer_key]]
        self.buffer_count[buffer_key] += 1
        return out

    def _generate_single(self, parallel_envs: int, events: int, shape: Tuple[int], key: Optional[str]) -> torch.Tensor:
        """One shared host stream (random_generator.py:103-106, 124-131); returns ``[events, parallel_envs, *shape]`` on the device."""
        g = self._single
        if key is None or self.buffer_size == 0:
            g.set_state(self._single_state)
            out = torch.rand((parallel_envs, events, *shape), generator=g)
            self._single_state = g.get_state()
            return out.transpose(1, 0).contiguous().to(self.device)
        buffer_key = (key, (parallel_envs, events, *shape))
        if buffer_key not in self.buffers or self.buffer_count[buffer_key] >= self.buffer_size:
            g.set_state(self._single_state)
            block = torch.rand((self.buffer_size, parallel_envs, events, *shape), generator=g)
            self._single_state = g.get_state()
            self.buffers[buffer_key] = block.transpose(1, 2).contiguous().to(self.device)
            self.buffer_count[buffer_key] = 0
        out = self.buffers[buffer_key][self.buffer_count[buffer_key]]
        self.buffer_count[buffer_key] += 1
        return out

    def state_dict(self) -> dict:
        return {
            'parallel_envs': self.parallel_envs,
            'buffer_size': self.buffer_size,
            'single_seeding': self.single_seeding,
            'device': str(self.device),
            'seeds': self.seeds.clone(),
            'generator_states': self.generator_states.clone() if self._streams_valid else None,
            'streams_valid': self._streams_valid,
            'generator_index': self.generator_index.clone(),
            'buffer_count': dict(self.buffer_count),
            'buffers': {k: v.clone() for k, v in self.buffers.items()},
            'has_been_seeded': self.has_been_seeded,
            'single_state': None if self._single_state is None else self._single_state.clone(),
        }

    def load_state_dict(self, state: dict) -> None:
        self.parallel_envs = state['parallel_envs']
        self.buffer_size = state['buffer_size']
        self.single_seeding = state['single_seeding']
        self.seeds.copy_(state['seeds'])
        self._streams_valid = state.get('streams_valid', True)
        if self._streams_valid:
            if self.generator_states is None:
                self.generator_states = torch.empty((624, self.parallel_envs), dtype=torch.int32, device=self.device)
            self.generator_states.copy_(state['generator_states'])
        self.generator_index.copy_(state['generator_index'])
        self.buffer_count = dict(state['buffer_count'])
        self.buffers = {k: v.clone() for k, v in state['buffers'].items()}
        self.has_been_seeded = state['has_been_seeded']
        single = state.get('single_state')
        self._single_state = None if single is None else single.clone()
        if self.single_seeding and self._single is None:
            self._single = torch.Generator(device='cpu')
