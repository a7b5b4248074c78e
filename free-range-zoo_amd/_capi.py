"""ctypes binding of the C-ABI in include/frz.h to libfrz_hip.so (hand-written HIP kernels, gfx950).

There is NO CPU fallback: if the shared library is missing or a symbol cannot be resolved, loading fails loudly.
"""
import ctypes
import os
from typing import Any, Dict

from ._cstruct import parse_header

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_PKG_DIR)
HEADER = os.path.join(REPO_ROOT, 'include', 'frz.h')
# FRZ_HIP_LIB selects another build of the same C-ABI (diagnostic builds only, e.g. libfrz_hip_stamps.so)
LIB_PATH = os.environ.get('FRZ_HIP_LIB') or os.path.join(_PKG_DIR, 'csrc', 'libfrz_hip.so')

DEFINES, STRUCTS = parse_header(HEADER)
globals().update({k: v for k, v in DEFINES.items()})

frz_wildfire_cfg = STRUCTS['frz_wildfire_cfg']
frz_wildfire_bufs = STRUCTS['frz_wildfire_bufs']
frz_cybersecurity_cfg = STRUCTS['frz_cybersecurity_cfg']
frz_cybersecurity_bufs = STRUCTS['frz_cybersecurity_bufs']
frz_rideshare_cfg = STRUCTS['frz_rideshare_cfg']
frz_rideshare_bufs = STRUCTS['frz_rideshare_bufs']
frz_rollout_spec = STRUCTS['frz_rollout_spec']
frz_wildfire_saved_state = STRUCTS['frz_wildfire_saved_state']
frz_cybersecurity_saved_state = STRUCTS['frz_cybersecurity_saved_state']

_lib = None

# name -> (restype, argtypes); every function include/frz.h declares
_P = ctypes.c_void_p
SIGNATURES = {
    'frz_abi_version': (ctypes.c_int, []),
    'frz_handle_kind': (ctypes.c_int, [_P]),
    'frz_handle_shape': (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'frz_wildfire_create': (ctypes.c_int, [_P, ctypes.POINTER(_P)]),
    'frz_wildfire_destroy': (None, [_P]),
    'frz_wildfire_arena_bytes': (ctypes.c_int64, [_P]),
    'frz_wildfire_bind': (ctypes.c_int, [_P, _P, _P]),
    'frz_wildfire_get_bufs': (ctypes.c_int, [_P, _P]),
    'frz_wildfire_reset': (ctypes.c_int, [_P, _P]),
    'frz_wildfire_reset_reseed': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'frz_wildfire_rebuild': (ctypes.c_int, [_P, _P]),
    'frz_wildfire_step': (ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P, _P]),
    'frz_wildfire_random_policy': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    'frz_wildfire_step_random_policy': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, _P, ctypes.c_int, _P, _P, _P]),
    'frz_wildfire_rollout_random_policy': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, _P, ctypes.c_int, _P]),
    'frz_wildfire_extreme_fire_policy': (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64,
                                                       ctypes.c_int64, _P, _P]),
    'frz_rideshare_task_policy': (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64,
                                                ctypes.c_int64, _P, _P, _P]),
    'frz_cybersecurity_focus_policy': (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, _P, ctypes.c_int32, ctypes.c_int64,
                                                     ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64,
                                                     ctypes.c_uint64, ctypes.c_int64, _P, _P, _P, _P, _P]),
    'frz_wildfire_episode_metrics': (ctypes.c_int, [_P, _P, _P]),
    'frz_wildfire_rollout_random_policy_metrics': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, _P, ctypes.c_int, _P, _P]),
    'frz_wildfire_timed_rollout': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, _P, ctypes.c_int, _P,
                                                 ctypes.POINTER(ctypes.c_float)]),
    'frz_wildfire_set_exclusive_device': (ctypes.c_int, [_P, ctypes.c_int]),
    'frz_exclusive_launch_fits': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'frz_wildfire_export_totals': (ctypes.c_int, [_P, _P, _P]),
    'frz_wildfire_import_totals': (ctypes.c_int, [_P, _P, _P]),
    'frz_wildfire_rollout_launches': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int]),
    'frz_wildfire_timed_rollout_launch': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, _P, ctypes.c_int, _P,
                                                        ctypes.POINTER(ctypes.c_float)]),
    'frz_wildfire_rollout': (ctypes.c_int, [_P, _P, _P]),
    'frz_wildfire_timed_rollout_spec': (ctypes.c_int, [_P, _P, _P, ctypes.POINTER(ctypes.c_float)]),
    'frz_wildfire_list_block': (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64)]),
    'frz_wildfire_obs_block': (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'frz_wildfire_state_block': (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64)]),
    'frz_wildfire_reset_masked': (ctypes.c_int, [_P, _P, ctypes.c_int32, _P]),
    'frz_wildfire_set_saved_initial': (ctypes.c_int, [_P, _P]),
    'frz_cybersecurity_create': (ctypes.c_int, [_P, ctypes.POINTER(_P)]),
    'frz_cybersecurity_destroy': (None, [_P]),
    'frz_cybersecurity_arena_bytes': (ctypes.c_int64, [_P]),
    'frz_cybersecurity_bind': (ctypes.c_int, [_P, _P, _P]),
    'frz_cybersecurity_get_bufs': (ctypes.c_int, [_P, _P]),
    'frz_cybersecurity_reset': (ctypes.c_int, [_P, _P]),
    'frz_cybersecurity_rebuild': (ctypes.c_int, [_P, _P]),
    'frz_cybersecurity_step': (ctypes.c_int, [_P, _P, ctypes.c_int, _P, _P, _P]),
    'frz_cybersecurity_random_policy': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    'frz_cybersecurity_step_random_policy': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, _P, ctypes.c_int, _P, _P, _P]),
    'frz_cybersecurity_rollout_random_policy': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, _P, ctypes.c_int, _P]),
    'frz_cybersecurity_rollout': (ctypes.c_int, [_P, _P, _P]),
    'frz_rideshare_rollout': (ctypes.c_int, [_P, _P, _P]),
    'frz_rideshare_list_block': (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64)]),
    'frz_cybersecurity_list_block': (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64)]),
    'frz_cybersecurity_reset_masked': (ctypes.c_int, [_P, _P, ctypes.c_int32, _P]),
    'frz_cybersecurity_set_saved_initial': (ctypes.c_int, [_P, _P]),
    'frz_cybersecurity_obs_block': (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64)]),
    'frz_cybersecurity_state_block': (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64)]),
    'frz_cybersecurity_set_exclusive_device': (ctypes.c_int, [_P, ctypes.c_int]),
    'frz_cybersecurity_rollout_launches': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int]),
    'frz_rideshare_create': (ctypes.c_int, [_P, _P, ctypes.POINTER(_P)]),
    'frz_rideshare_destroy': (None, [_P]),
    'frz_rideshare_arena_bytes': (ctypes.c_int64, [_P]),
    'frz_rideshare_bind': (ctypes.c_int, [_P, _P, _P]),
    'frz_rideshare_get_bufs': (ctypes.c_int, [_P, _P]),
    'frz_rideshare_reset': (ctypes.c_int, [_P, _P]),
    'frz_rideshare_rebuild': (ctypes.c_int, [_P, _P]),
    'frz_rideshare_step': (ctypes.c_int, [_P, _P, _P]),
    'frz_rideshare_random_policy': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    'frz_rideshare_step_random_policy': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    'frz_rideshare_timed_rollout': (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, _P, _P, ctypes.POINTER(ctypes.c_float)]),
    'frz_mt19937_seed': (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int64, ctypes.c_int64, _P]),
    'frz_mt19937_generate': (ctypes.c_int, [_P, _P, _P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _P]),
    'frz_mt19937_generate_pair': (ctypes.c_int, [_P, _P, _P, ctypes.c_int64, ctypes.c_int64, _P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _P]),
}


class FrzError(RuntimeError):
    """A C-ABI entry point returned a negative FRZ_E_* code."""


def lib() -> ctypes.CDLL:
    """Load libfrz_hip.so once; raise (never fall back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f'{LIB_PATH} is missing: build the HIP extension first '
                              f'(python -c "import __graft_entry__ as g; g.build()" or make -C free-range-zoo_amd/csrc). '
                              f'There is no CPU fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.frz_abi_version() != DEFINES['FRZ_ABI_VERSION']:
            raise ImportError('libfrz_hip.so ABI version does not match include/frz.h; rebuild')
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        names = {v: k for k, v in DEFINES.items() if k.startswith('FRZ_E_')}
        raise FrzError(f'{what} failed: {names.get(code, code)}')


def struct_to_dict(s: ctypes.Structure) -> Dict[str, Any]:
    """Plain-python view of a scalar/array struct (used to store configurations in golden fixtures)."""
    out = {}
    for name, ctype in s._fields_:
        value = getattr(s, name)
        if isinstance(value, ctypes.Array):
            def unroll(v):
                return [unroll(x) for x in v] if isinstance(v, ctypes.Array) else v
            value = unroll(value)
        out[name] = value
    return out


def struct_from_dict(cls, values: Dict[str, Any]) -> ctypes.Structure:
    s = cls()
    for name, ctype in cls._fields_:
        if name not in values:
            continue
        value = values[name]
        if isinstance(value, (list, tuple)):
            arr = getattr(s, name)
            for i, row in enumerate(value):
                if isinstance(row, (list, tuple)):
                    for j, x in enumerate(row):
                        arr[i][j] = x
                else:
                    arr[i] = row
        else:
            setattr(s, name, value)
    return s
