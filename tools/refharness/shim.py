"""Stand-ins for third-party packages the reference imports but this image lacks.

TEST INFRASTRUCTURE, used only in the build container (where /root/reference
exists) to import the *unmodified* reference from /root/reference and record
golden vectors (tools/refharness/make_golden.py).  Nothing here is shipped,
nothing here is arithmetic: every number in a golden vector is produced by the
reference's own torch code.  The stand-ins only provide the class skeletons
the reference subclasses / instantiates:

  typing.Self                    -> typing_extensions.Self (reference needs py>=3.11)
  tensordict.TensorDict          -> dict subclass (keeps batch_size/device attrs)
  gymnasium                      -> empty ``Space`` class (only used in annotations)
  pettingzoo.AECEnv, agent_selector, OrderEnforcingWrapper,
  aec_to_parallel_wrapper, ParallelEnv  -> minimal skeletons written from the
                                    public pettingzoo 1.24 API
  free_range_rust.Space          -> inert placeholder (golden actions are drawn
                                    from the recorded task counts instead)
  supersuit / pygame             -> empty modules (wrappers/rendering are out of scope)
"""
import sys
import types
import typing

REFERENCE_ROOT = '/root/reference'


def _module(name: str) -> types.ModuleType:
    mod = types.ModuleType(name)
    sys.modules[name] = mod
    parent, _, child = name.rpartition('.')
    if parent:
        setattr(sys.modules[parent], child, mod)
    return mod


def install() -> None:
    """Install the stand-ins and put the reference on sys.path (idempotent)."""
    if getattr(install, '_done', False):
        return
    install._done = True

    if not hasattr(typing, 'Self'):
        import typing_extensions
        typing.Self = typing_extensions.Self

    # ---------------------------------------------------------------- tensordict
    td = _module('tensordict')
    tdd = _module('tensordict.tensordict')

    class TensorDict(dict):
        def __init__(self, source=None, batch_size=None, device=None, **kwargs):
            super().__init__(source or {})
            self.batch_size = batch_size
            self.device = device

    td.TensorDict = TensorDict
    tdd.TensorDict = TensorDict

    # ----------------------------------------------------------------- gymnasium
    gym = _module('gymnasium')

    class Space:  # annotation target only
        pass

    gym.Space = Space

    # ---------------------------------------------------------------- pettingzoo
    pz = _module('pettingzoo')
    pzu = _module('pettingzoo.utils')
    pzw = _module('pettingzoo.utils.wrappers')
    pzc = _module('pettingzoo.utils.conversions')
    pze = _module('pettingzoo.utils.env')

    class AECEnv:
        metadata = {}

        def __init__(self, *args, **kwargs):
            pass

        @property
        def num_agents(self):
            return len(self.agents)

        @property
        def max_num_agents(self):
            return len(self.possible_agents)

        @property
        def unwrapped(self):
            return self

    class ParallelEnv(typing.Generic[typing.TypeVar('A'), typing.TypeVar('O'), typing.TypeVar('C')]):
        metadata = {}

        @property
        def num_agents(self):
            return len(self.agents)

        @property
        def max_num_agents(self):
            return len(self.possible_agents)

    class agent_selector:
        """Cyclic agent iterator (pettingzoo.utils.agent_selector semantics)."""

        def __init__(self, agent_order):
            self.reinit(agent_order)

        def reinit(self, agent_order):
            self.agent_order = agent_order
            self._current_agent = 0
            self.selected_agent = 0

        def reset(self):
            self.reinit(self.agent_order)
            return self.next()

        def next(self):
            self._current_agent = (self._current_agent + 1) % len(self.agent_order)
            self.selected_agent = self.agent_order[self._current_agent - 1]
            return self.selected_agent

        def is_last(self):
            return self.selected_agent == self.agent_order[-1]

        def is_first(self):
            return self.selected_agent == self.agent_order[0]

    class OrderEnforcingWrapper(AECEnv):
        """Attribute-forwarding wrapper (the order checks themselves are not needed here)."""

        def __init__(self, env):
            self.__dict__['env'] = env

        def __getattr__(self, name):
            return getattr(self.__dict__['env'], name)

        def __setattr__(self, name, value):
            setattr(self.__dict__['env'], name, value)

        @property
        def unwrapped(self):
            return self.__dict__['env'].unwrapped

        @property
        def metadata(self):
            return self.__dict__['env'].metadata

    class aec_to_parallel_wrapper(ParallelEnv):
        def __init__(self, aec_env):
            assert aec_env.metadata.get('is_parallelizable', False)
            self.aec_env = aec_env
            try:
                self.possible_agents = aec_env.possible_agents
            except AttributeError:
                pass
            self.metadata = aec_env.metadata

        @property
        def unwrapped(self):
            return self.aec_env.unwrapped

        def observation_space(self, agent):
            return self.aec_env.observation_space(agent)

        def action_space(self, agent):
            return self.aec_env.action_space(agent)

        def state(self):
            return self.aec_env.state()

    pz.AECEnv = AECEnv
    pz.ParallelEnv = ParallelEnv
    pzu.agent_selector = agent_selector
    pzu.BaseParallelWrapper = ParallelEnv
    pzw.OrderEnforcingWrapper = OrderEnforcingWrapper
    pzw.BaseWrapper = OrderEnforcingWrapper
    pzc.aec_to_parallel_wrapper = aec_to_parallel_wrapper
    pze.ParallelEnv = ParallelEnv
    pze.AECEnv = AECEnv
    pze.AgentID = typing.Any
    pze.ObsType = typing.Any
    pze.ActionType = typing.Any

    # ----------------------------------------------------------- free_range_rust
    frr = _module('free_range_rust')

    class _SpaceMeta(type):
        def __getattr__(cls, name):  # Space.Vector / OneOf / Discrete / Box / Dict / Tuple
            def build(*args, **kwargs):
                return cls(name, args, kwargs)
            return build

    class RustSpace(metaclass=_SpaceMeta):
        def __init__(self, kind, args, kwargs):
            self.kind, self.args, self.kwargs = kind, args, kwargs

        def __eq__(self, other):
            return (self.kind, self.args, self.kwargs) == (other.kind, other.args, other.kwargs)

        def __hash__(self):
            return hash(self.kind)

    frr.Space = RustSpace

    # ------------------------------------------------------- supersuit / pygame
    ss = _module('supersuit')
    _module('supersuit.generic_wrappers')
    _module('supersuit.generic_wrappers.utils')
    bm = _module('supersuit.generic_wrappers.utils.base_modifier')
    bm.BaseModifier = type('BaseModifier', (), {})
    _module('supersuit.utils')
    wc = _module('supersuit.utils.wrapper_chooser')
    wc.WrapperChooser = type('WrapperChooser', (), {'__init__': lambda self, **kw: None})
    _module('pygame')
    del ss

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
