"""Build ctypes mirrors of the plain-C structs declared in include/frz.h (and oracle/frz_oracle.h).

The headers are the single source of truth for the C-ABI; parsing them at import keeps the Python
bindings from drifting.  Only the declaration forms used in those headers are understood:
``#define NAME <int>``, ``typedef struct name { ... } name;`` with scalar, fixed-array and pointer members.
"""
import ctypes
import re
from typing import Dict, Tuple

_SCALARS = {
    'int32_t': ctypes.c_int32,
    'uint32_t': ctypes.c_uint32,
    'int64_t': ctypes.c_int64,
    'uint64_t': ctypes.c_uint64,
    'uint8_t': ctypes.c_uint8,
    'int8_t': ctypes.c_int8,
    'float': ctypes.c_float,
    'double': ctypes.c_double,
    'int': ctypes.c_int,
    'void': None,
}


def _strip_comments(text: str) -> str:
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    return re.sub(r'//[^\n]*', ' ', text)


def parse_header(path: str, known: Dict[str, type] = None) -> Tuple[Dict[str, int], Dict[str, type]]:
    """Return (defines, structs) found in the header at ``path``."""
    text = _strip_comments(open(path).read())
    defines: Dict[str, int] = {}
    for name, value in re.findall(r'^\s*#define\s+(\w+)\s+\(?(-?\d+)u?\)?\s*$', text, flags=re.M):
        defines[name] = int(value)
    structs: Dict[str, type] = dict(known or {})
    for body, name in re.findall(r'typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;', text, flags=re.S):
        fields = []
        for decl in body.split(';'):
            decl = ' '.join(decl.split())
            if not decl:
                continue
            decl = decl.replace('const ', '')
            m = re.match(r'(\w+)\s*(\**)\s*(.*)$', decl)
            base, stars, rest = m.group(1), m.group(2), m.group(3)
            for item in rest.split(','):
                item = item.strip()
                item_stars = stars
                while item.startswith('*'):
                    item_stars += '*'
                    item = item[1:].strip()
                dims = [defines[d] if d in defines else int(d) for d in re.findall(r'\[(\w+)\]', item)]
                fname = re.match(r'\w+', item).group(0)
                if item_stars:
                    ctype = ctypes.c_void_p
                else:
                    ctype = structs[base] if base in structs else _SCALARS[base]
                    for d in reversed(dims):
                        ctype = ctype * d
                fields.append((fname, ctype))
        structs[name] = type(name, (ctypes.Structure, ), {'_fields_': fields})
    return defines, structs
