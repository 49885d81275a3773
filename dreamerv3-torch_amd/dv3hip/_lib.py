"""ctypes loader for libdv3hip.so.

The argument types of every entry point are parsed from `include/dv3hip.h`, so the header is the
single source of truth for the C ABI.  Loading fails loudly (ImportError / RuntimeError) when the
library has not been built: there is no CPU or PyTorch fallback for the product path.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
_REPO = os.path.dirname(_PKG)
# DV3HIP_LIB: development override (A/B two builds of the library inside one GPU session)
LIB_PATH = os.environ.get("DV3HIP_LIB") or os.path.join(_HERE, "libdv3hip.so")
HEADER_PATH = os.path.join(_REPO, "include", "dv3hip.h")

_CTYPES = {
    "const float*": ctypes.c_void_p,
    "float*": ctypes.c_void_p,
    "const unsigned char*": ctypes.c_void_p,
    "const unsigned long long*": ctypes.c_void_p,
    "unsigned long long*": ctypes.c_void_p,
    "int*": ctypes.c_void_p,
    "const int*": ctypes.c_void_p,
    "unsigned int*": ctypes.c_void_p,
    "const unsigned int*": ctypes.c_void_p,
    "void*": ctypes.c_void_p,
    "const float* const*": ctypes.c_void_p,  # (host arrays of the grouped entry points)
    "float* const*": ctypes.c_void_p,
    "const long*": ctypes.c_void_p,
    "unsigned long long": ctypes.c_ulonglong,
    "long": ctypes.c_long,
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


def parse_header(path: str = HEADER_PATH) -> Dict[str, List[Tuple[str, str]]]:
    """-> {function name: [(ctype string, arg name), ...]} for every `int dv3_*(...)` declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint\s+(dv3_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        out = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.+?)\s*(\w+)$", a)
                ty, nm = mm.group(1).strip(), mm.group(2)
                ty = ty.replace(" *", "*")
                if ty not in _CTYPES:
                    raise ValueError(f"{name}: unsupported C type {ty!r}")
                out.append((ty, nm))
        decls[name] = out
    return decls


class DV3Error(RuntimeError):
    pass


_lib = None
_decls = None


def load() -> ctypes.CDLL:
    global _lib, _decls
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; import it FIRST so this process has exactly one HIP runtime and
    # the streams / device pointers torch hands us belong to the runtime our kernels launch on.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not built. Run `python dreamerv3-torch_amd/csrc/build.py` (needs hipcc, gfx950). "
            "The MI355X hot path has no fallback implementation."
        )
    lib = ctypes.CDLL(LIB_PATH)
    _decls = parse_header()
    for name, args in _decls.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale w.r.t. the header
        fn.restype = ctypes.c_int
        fn.argtypes = [_CTYPES[t] for t, _ in args]
    _lib = lib
    return lib


def declarations() -> Dict[str, List[Tuple[str, str]]]:
    load()
    return _decls


def check(code: int, what: str) -> None:
    if code != 0:
        if code == 10001:
            raise DV3Error(f"{what}: argument rejected by libdv3hip (DV3_ERR_ARG)")
        raise DV3Error(f"{what}: HIP error {code}")
