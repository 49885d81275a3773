"""dv3hip -- Python binding of libdv3hip (gfx950 HIP kernels behind a C ABI).

`ops`   : one thin wrapper per C entry point (shape/dtype/device checks, current-stream launch).
`_lib`  : ctypes loader; parses include/dv3hip.h for the signatures.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
