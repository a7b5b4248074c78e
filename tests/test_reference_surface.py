"""Drop-in surface check against the reference's SOURCE TEXT (no import): configuration dataclasses carry the same field names,
the env base class takes the same constructor keywords, the baselines / wrappers packages export the same class names.

Runs only where /root/reference exists (the build container); parses the files with ``ast`` — nothing of the reference is executed
or copied.  Skipped elsewhere (the GPU box has no reference tree)."""
import ast
import importlib
import inspect
import os

import pytest

REF = '/root/reference/free_range_zoo'
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _tree(relative):
    with open(os.path.join(REF, relative)) as handle:
        return ast.parse(handle.read())


def _dataclass_fields(tree):
    """{class name: [annotated field names]} for every class of a module (dataclass fields are the annotated assignments)."""
    out = {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            out[node.name] = [stmt.target.id for stmt in node.body if isinstance(stmt, ast.AnnAssign) and isinstance(stmt.target, ast.Name)]
    return out


def _function_args(tree, class_name, function_name):
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            for stmt in node.body:
                if isinstance(stmt, ast.FunctionDef) and stmt.name == function_name:
                    names = [a.arg for a in stmt.args.args + stmt.args.kwonlyargs]
                    return [n for n in names if n != 'self']
    raise AssertionError(f'{class_name}.{function_name} not found')


@needs_reference
@pytest.mark.parametrize('domain', ['wildfire', 'rideshare', 'cybersecurity'])
def test_configuration_dataclasses_have_the_reference_fields(domain):
    reference = _dataclass_fields(_tree(f'envs/{domain}/env/structures/configuration.py'))
    mine = importlib.import_module(f'free_range_zoo_amd.envs.{domain}.env.structures.configuration')
    checked = 0
    for name, fields in reference.items():
        if not name.endswith('Configuration') or not fields:
            continue
        cls = getattr(mine, name, None)
        assert cls is not None, f'{domain}: {name} is missing'
        own = list(getattr(cls, '__dataclass_fields__', {}))
        assert own == fields, f'{domain}.{name}: fields {own} != reference {fields}'
        checked += 1
    assert checked >= 4


@needs_reference
@pytest.mark.parametrize('domain', ['wildfire', 'rideshare', 'cybersecurity'])
def test_state_dataclasses_have_the_reference_fields(domain):
    reference = _dataclass_fields(_tree(f'envs/{domain}/env/structures/state.py'))
    mine = importlib.import_module(f'free_range_zoo_amd.envs.{domain}.env.structures.state')
    for name, fields in reference.items():
        if name.endswith('State') and fields:
            own = list(getattr(getattr(mine, name), '__dataclass_fields__', {}))
            assert own == fields, f'{domain}.{name}: fields {own} != reference {fields}'


@needs_reference
def test_env_constructor_keywords_cover_the_reference():
    reference = _function_args(_tree('utils/env.py'), 'BatchedAECEnv', '__init__')
    from free_range_zoo_amd.utils.env import BatchedAECEnv as BatchedEnv
    own = list(inspect.signature(BatchedEnv.__init__).parameters)
    for name in reference:
        assert name in own, f'constructor keyword {name} of the reference env is not accepted'
    for domain, extra in (('wildfire', ['observe_other_suppressant', 'observe_other_power', 'show_bad_actions']),
                          ('cybersecurity', ['observe_other_location', 'observe_other_presence', 'observe_other_power', 'partially_observable',
                                             'show_bad_actions'])):
        args = _function_args(_tree(f'envs/{domain}/env/{domain}.py'), 'raw_env', '__init__')
        for name in extra:
            assert name in args, f'{domain}: the reference constructor no longer takes {name}?'
        module = importlib.import_module(f'free_range_zoo_amd.envs.{domain}.env.{domain}')
        mine = list(inspect.signature(module.raw_env.__init__).parameters)
        for name in args:
            assert name in mine or 'kwargs' in mine, f'{domain}: constructor keyword {name} is not accepted'


def test_env_methods_of_the_reference_exist():
    from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0, wildfire_v0
    for module in (wildfire_v0, rideshare_v0, cybersecurity_v0):
        for name in ('reset', 'reset_batches', 'step', 'observe', 'state', 'action_space', 'observation_space', 'close'):
            assert callable(getattr(module.raw_env, name, None)), f'{module.__name__}.raw_env.{name}'
        for name in ('finished', 'terminated', 'truncated', 'num_agents', 'max_num_agents'):
            assert isinstance(getattr(module.raw_env, name, None), property), f'{module.__name__}.raw_env.{name}'
        assert callable(module.parallel_env) and callable(module.env)


@needs_reference
@pytest.mark.parametrize('domain', ['wildfire', 'rideshare', 'cybersecurity'])
def test_baseline_packages_export_the_reference_classes(domain):
    tree = _tree(f'envs/{domain}/baselines/__init__.py')
    names = [alias.asname or alias.name for node in tree.body if isinstance(node, ast.ImportFrom) for alias in node.names]
    mine = importlib.import_module(f'free_range_zoo_amd.envs.{domain}.baselines')
    assert names, domain
    for name in names:
        cls = getattr(mine, name, None)
        assert inspect.isclass(cls), f'{domain}.baselines.{name} is missing'
        assert callable(getattr(cls, 'act', None)) and callable(getattr(cls, 'observe', None))


@needs_reference
def test_random_generator_and_agent_interfaces():
    reference = _tree('utils/random_generator.py')
    from free_range_zoo_amd.utils.random_generator import RandomGenerator
    for method in ('seed', 'generate'):
        args = _function_args(reference, 'RandomGenerator', method)
        own = list(inspect.signature(getattr(RandomGenerator, method)).parameters)
        for name in args:
            assert name in own, f'RandomGenerator.{method}({name}) missing'
    assert _function_args(reference, 'RandomGenerator', '__init__') == [p for p in inspect.signature(RandomGenerator.__init__).parameters if p != 'self']
    agent = _function_args(_tree('utils/agent.py'), 'Agent', '__init__')
    from free_range_zoo_amd.utils.agent import Agent
    assert agent == [p for p in inspect.signature(Agent.__init__).parameters if p != 'self']


# ------------------------------------------------------------------------------------------------------------------
# configurations pickled with the reference package load into this package's classes (utils/compat.py)
# ------------------------------------------------------------------------------------------------------------------
import pickle  # noqa: E402

import torch  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _same(a, b, path=''):
    assert type(a).__name__ == type(b).__name__, path
    for name, value in vars(a).items():
        other = getattr(b, name)
        if isinstance(value, torch.Tensor):
            assert value.dtype == other.dtype and torch.equal(value.cpu(), other.cpu()), f'{path}.{name}'
        elif hasattr(value, '__dataclass_fields__'):
            _same(value, other, f'{path}.{name}')
        else:
            assert value == other, f'{path}.{name}'


def test_reference_pickles_load_into_this_package(request):
    """tests/golden/reference_configuration_<domain>.pkl = pickle.dump of reference Configuration objects
    (tools/refharness/make_golden.py pickles): the files name `free_range_zoo...` classes and load as this package's."""
    import configs
    from free_range_zoo_amd.utils.compat import load_reference_pickle
    expected = {'wildfire': configs.WILDFIRE_GOLDEN['rich_localized'][0](), 'cybersecurity': configs.CYBER_GOLDEN['rich'][0](),
                'rideshare': configs.RIDESHARE_GOLDEN['busy_waiting_costs']()}
    for domain, want in expected.items():
        path = os.path.join(GOLDEN, f'reference_configuration_{domain}.pkl')
        with open(path, 'rb') as handle:
            raw = handle.read()
        assert b'free_range_zoo.envs' in raw and b'free_range_zoo_amd' not in raw
        for source in (path, raw):
            got = load_reference_pickle(source)
            assert type(got).__module__.startswith('free_range_zoo_amd.envs.' + domain)
            _same(got, want, domain)


def test_reference_pickle_loader_refuses_everything_else():
    from free_range_zoo_amd.utils.compat import load_reference_pickle
    for payload in (pickle.dumps(os.system), pickle.dumps(print), b"cfree_range_zoo.utils.env\nBatchedAECEnv\n."):
        with pytest.raises(pickle.UnpicklingError):
            load_reference_pickle(payload)
    assert load_reference_pickle(pickle.dumps({'a': torch.arange(3), 'b': (1, 2.5, 'x')}))['a'].tolist() == [0, 1, 2]


class _Run:
    """A pickle whose load would call ``fn(*args)``."""

    def __init__(self, fn, *args):
        self.fn, self.args = fn, args

    def __reduce__(self):
        return self.fn, self.args


def test_reference_pickle_loader_refuses_code_under_allowed_packages(tmp_path):
    """Globals are allowed by exact name, not by package prefix: a callable that lives under torch / numpy / collections is refused,
    and a tensor's storage bytes are read with weights_only=True (torch.storage._load_from_bytes would run a nested full unpickle)."""
    import collections
    import io
    import numpy
    import torch.utils.collect_env
    from free_range_zoo_amd.utils.compat import load_reference_pickle
    marker = tmp_path / 'ran'
    payloads = [
        pickle.dumps(_Run(torch.utils.collect_env.run, f'touch {marker}')),
        pickle.dumps(_Run(numpy.load, str(marker), None, True)),
        pickle.dumps(_Run(collections.namedtuple, 'x', 'a b')),
        pickle.dumps(_Run(torch.load, str(marker))),
        pickle.dumps(_Run(torch.hub.load, 'x', 'y')),
    ]
    # the nested route: bytes handed to the storage loader that are themselves a program
    inner = io.BytesIO()
    pickle.dump(_Run(os.system, f'touch {marker}'), inner)
    payloads.append(pickle.dumps(_Run(torch.storage._load_from_bytes, inner.getvalue())))
    for payload in payloads:
        with pytest.raises(Exception) as caught:  # UnpicklingError from find_class, or torch's weights_only refusal for the nested bytes
            load_reference_pickle(payload)
        assert isinstance(caught.value, (pickle.UnpicklingError, RuntimeError)), caught.value
        assert not marker.exists()
    # what a configuration does hold still loads: tensors of several dtypes, a numpy array and scalar, an OrderedDict
    ok = {'t': torch.arange(6, dtype=torch.int32).reshape(2, 3), 'b': torch.ones(2, dtype=torch.bool), 'f': torch.zeros(2, dtype=torch.float64),
          'n': numpy.arange(3), 's': numpy.float32(1.5), 'o': collections.OrderedDict(a=1)}
    got = load_reference_pickle(pickle.dumps(ok))
    assert got['t'].tolist() == [[0, 1, 2], [3, 4, 5]] and got['n'].tolist() == [0, 1, 2] and float(got['s']) == 1.5 and got['o']['a'] == 1
