"""Import name of the package that lives in ``free-range-zoo_amd/`` (a directory name Python cannot import): this module replaces
itself in ``sys.modules`` with that directory loaded as the package ``free_range_zoo_amd`` (importlib's load-from-location recipe)."""
import importlib.util as _util
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'free-range-zoo_amd')
_spec = _util.spec_from_file_location(__name__, _os.path.join(_real, '__init__.py'), submodule_search_locations=[_real])
_module = _util.module_from_spec(_spec)
_sys.modules[__name__] = _module
_spec.loader.exec_module(_module)
