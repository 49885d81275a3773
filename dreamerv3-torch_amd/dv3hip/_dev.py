"""Development (A/B measurement) switches of the host code.

The shipped library is built without DV3_DEV_SWITCHES: `dv3_dev_switches()` returns 0 and every switch below is
its default whatever the environment holds -- an integrator's stray DV3_* variable cannot change which kernel
runs.  `python dreamerv3-torch_amd/csrc/build.py --dev` + DV3HIP_LIB=.../libdv3hip_dev.so turns them on for
tools/*_bench.py."""
from __future__ import annotations

import os

from . import _lib


def enabled() -> bool:
    return bool(_lib.load().dv3_dev_switches())


def flag(name: str, default: bool) -> bool:
    if not enabled():
        return default
    return os.environ.get(name, "1" if default else "0") != "0"


def value(name: str, default, cast=int):
    if not enabled():
        return default
    return cast(os.environ.get(name, default))
