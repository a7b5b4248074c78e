"""Loading configurations pickled with the reference package.

The reference distributes its competition configurations as pickles of its own ``Configuration`` dataclasses
(docs/source/events/moasei-2026/quickstart_fire.md: "Configurations must be loaded using pickle").  Those files name classes of the
``free_range_zoo`` package; ``load_reference_pickle`` resolves them to the classes of this package — same names, same fields — so the
file a user already has keeps working:

    configuration = load_reference_pickle('<path to configuration>.pkl')
    env = wildfire_v0.parallel_env(configuration=configuration, ...)

Only what such a file needs is resolvable — an EXACT (module, name) list, not module prefixes: configuration / state classes of this
package under their reference names, the tensor / array rebuild helpers, ``collections.OrderedDict`` and a few builtin containers.
Tensor storages travel inside such a pickle as bytes that ``torch.storage._load_from_bytes`` would hand to an unrestricted
``torch.load``; here they are read with ``weights_only=True``.  Anything else raises ``pickle.UnpicklingError`` (a pickle is a program).
"""
import importlib
import io
import pickle
from typing import Any, BinaryIO, Union

REFERENCE_PACKAGE = 'free_range_zoo'
OWN_PACKAGE = 'free_range_zoo_amd'
# every global a tensor / ndarray / scalar inside a configuration needs, by exact name (numpy 1.x and 2.x module paths)
_ALLOWED_GLOBALS = {
    ('collections', 'OrderedDict'),
    ('torch._utils', '_rebuild_tensor_v2'),
    ('torch._utils', '_rebuild_parameter'),
    ('torch', 'Size'),
    ('torch', 'device'),
    ('numpy', 'dtype'),
    ('numpy', 'ndarray'),
    ('numpy.core.multiarray', '_reconstruct'),
    ('numpy.core.multiarray', 'scalar'),
    ('numpy._core.multiarray', '_reconstruct'),
    ('numpy._core.multiarray', 'scalar'),
}
_TORCH_DTYPES = {'float16', 'bfloat16', 'float32', 'float64', 'uint8', 'int8', 'int16', 'int32', 'int64', 'bool', 'complex64', 'complex128'}
_ALLOWED_BUILTINS = {'set', 'frozenset', 'list', 'dict', 'tuple', 'int', 'float', 'bool', 'str', 'bytes', 'complex', 'slice', 'range', 'bytearray'}


class _ReferenceUnpickler(pickle.Unpickler):

    def find_class(self, module: str, name: str):
        if module == REFERENCE_PACKAGE or module.startswith(REFERENCE_PACKAGE + '.'):
            own = OWN_PACKAGE + module[len(REFERENCE_PACKAGE):]
            try:
                cls = getattr(importlib.import_module(own), name)
            except (ImportError, AttributeError) as error:
                raise pickle.UnpicklingError(f'{module}.{name} has no counterpart in {OWN_PACKAGE}') from error
            from free_range_zoo_amd.utils.configuration import Configuration
            from free_range_zoo_amd.utils.state import State
            if not (isinstance(cls, type) and issubclass(cls, (Configuration, State))):
                raise pickle.UnpicklingError(f'{module}.{name}: only configuration / state classes are loaded from reference pickles')
            return cls
        if module == 'builtins' and name in _ALLOWED_BUILTINS:
            return super().find_class(module, name)
        if (module, name) == ('torch.storage', '_load_from_bytes'):
            return _load_storage_from_bytes
        if (module, name) in _ALLOWED_GLOBALS or (module == 'torch' and name in _TORCH_DTYPES):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'{module}.{name} is not something a configuration pickle needs')


def _load_storage_from_bytes(data: bytes):
    """Stand-in for ``torch.storage._load_from_bytes`` (a nested, unrestricted ``torch.load``): the same bytes, tensors / storages only."""
    import torch
    return torch.load(io.BytesIO(data), weights_only=True)


def load_reference_pickle(source: Union[str, bytes, BinaryIO]) -> Any:
    """Load an object pickled with the reference package (path, bytes or binary file object); the configuration is validated."""
    if isinstance(source, (bytes, bytearray)):
        obj = _ReferenceUnpickler(io.BytesIO(source)).load()
    elif isinstance(source, str):
        with open(source, 'rb') as handle:
            obj = _ReferenceUnpickler(handle).load()
    else:
        obj = _ReferenceUnpickler(source).load()
    if hasattr(obj, 'validate'):
        obj.validate()
    return obj
