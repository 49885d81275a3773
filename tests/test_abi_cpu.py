"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports exactly the symbols
`include/dv3hip.h` declares (no compute calls without a GPU)."""
import ctypes
import os
import subprocess

import pytest

from dv3hip import _lib


def test_library_is_built_and_loads():
    if not os.path.exists(_lib.LIB_PATH):
        import importlib.util

        spec = importlib.util.spec_from_file_location(
            "dv3_build", os.path.join(os.path.dirname(os.path.dirname(_lib.LIB_PATH)), "csrc", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    lib = _lib.load()
    assert lib.dv3_version() >= 1


def test_every_declared_symbol_is_exported_with_c_linkage():
    decls = _lib.parse_header()
    assert len(decls) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in include/dv3hip.h but not exported"
    # and nothing is exported that the header does not declare
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("dv3_")}
    assert exported == set(decls), exported ^ set(decls)


def test_header_signatures_use_plain_c_types_only():
    for name, args in _lib.parse_header().items():
        for ty, _ in args:
            assert ty in _lib._CTYPES, (name, ty)
            assert "torch" not in ty and "Tensor" not in ty


def test_argument_rejection_happens_before_any_launch():
    """DV3_ERR_ARG paths return without touching the device (safe to call with no GPU)."""
    lib = _lib.load()
    assert lib.dv3_gemm_f32(0, 1, 4, 4, 4, None, 4, None, 0, 0, None, 4, None, 4, None, 0, -1, None) == 10001
    assert lib.dv3_ln_act_fwd(None, 0, None, None, None, 0, None, None, 4, 4096, 1, 0, None) == 10001
    assert lib.dv3_onehot_sample_fwd(None, None, None, 0, None, None, 4, 128, 0.01, 0, None) == 10001
    assert lib.dv3_gemm_f32(0, 1, 0, 4, 4, None, 4, None, 0, 0, None, 4, None, 4, None, 0, -1, None) == 0  # empty


def test_ops_fail_loudly_without_gpu_tensors():
    import torch

    from dv3hip import ops

    with pytest.raises(TypeError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4), torch.zeros(4, 4))
