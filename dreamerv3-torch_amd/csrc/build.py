#!/usr/bin/env python3
"""Build libdv3hip.so (gfx950 only) in-tree with hipcc.

    python dreamerv3-torch_amd/csrc/build.py [--force] [--dev]

Sources: every *.hip in this directory.  Output: ../dv3hip/libdv3hip.so (git-ignored; it
travels to the GPU box with the gpurun snapshot).  Objects are cached under ./_obj and
rebuilt when the source or any header is newer.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "dv3hip", "libdv3hip.so")
OBJ = os.path.join(HERE, "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I", HERE, "-I", os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.exists(c) or c == "hipcc"):
            return c
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = True, dev: bool = False) -> str:
    """dev=True: -DDV3_DEV_SWITCHES (the DV3_* A/B switches of tools/*_bench.py become live) into libdv3hip_dev.so,
    which dv3hip._lib loads only when DV3HIP_LIB points at it.  DV3_DEV_DEFINES="-DFOO ..." adds compile-time
    definitions to the dev build only (A/B of source-level variants against the shipped library)."""
    global OUT, OBJ
    if dev:
        OUT, OBJ = OUT.replace("libdv3hip.so", "libdv3hip_dev.so"), os.path.join(HERE, "_obj_dev")
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    hdrs = glob.glob(os.path.join(HERE, "*.h")) + glob.glob(
        os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "*.h"))
    hdr_time = max([os.path.getmtime(h) for h in hdrs] + [0.0])
    cc = hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [cc] + FLAGS + (["-DDV3_DEV_SWITCHES"] + os.environ.get("DV3_DEV_DEFINES", "").split() if dev else []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return o

    if jobs:
        if verbose:
            print(f"[dv3hip] compiling {len(jobs)} file(s) for {ARCH} ...", flush=True)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    need_link = bool(jobs) or not os.path.exists(OUT) or any(
        os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs)
    if need_link:
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[dv3hip] linked {OUT}", flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, dev="--dev" in sys.argv)
