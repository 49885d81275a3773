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
    out, obj = OUT, OBJ
    extra = []
    if dev:
        out, obj = OUT.replace("libdv3hip.so", "libdv3hip_dev.so"), os.path.join(HERE, "_obj_dev")
        extra = ["-DDV3_DEV_SWITCHES"] + os.environ.get("DV3_DEV_DEFINES", "").split()
    os.makedirs(obj, exist_ok=True)
    # the objects depend on the compile-time definitions as well as on the sources: another DV3_DEV_DEFINES set rebuilds
    # them (an A/B run must never measure stale objects)
    stamp = os.path.join(obj, "flags.stamp")
    flags_now = " ".join(FLAGS + extra)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    hdrs = glob.glob(os.path.join(HERE, "*.h")) + glob.glob(
        os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "*.h"))
    hdr_time = max([os.path.getmtime(h) for h in hdrs] + [0.0])
    cc = hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(obj, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [cc] + FLAGS + extra + ["-c", s, "-o", o]
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
    with open(stamp, "w") as f:
        f.write(flags_now)
    need_link = bool(jobs) or not os.path.exists(out) or any(
        os.path.getmtime(o) > os.path.getmtime(out) for o in objs)
    if need_link:
        tmp = out + ".tmp"  # (linked beside the target and renamed: a reader never maps a half-written library)
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        os.replace(tmp, out)
        if verbose:
            print(f"[dv3hip] linked {out}", flush=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, dev="--dev" in sys.argv)
