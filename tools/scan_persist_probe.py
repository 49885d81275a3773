#!/usr/bin/env python3
"""Bounded experiment (VERDICT r03 "Next" #6): the per-step floor of a PERSISTENT forward observe scan on MI355X -- one
launch for all 64 steps, the weights resident in LDS, a device-wide hand-off between the five dependent layers of a step
(tools/scan_persist_probe.hip: the scan's data flow without its row operations) -- against the 35.5 us per step the
captured sequence of five launches per step takes (tools/scan_bench.py).

    python tools/scan_persist_probe.py            # builds tools/_probe/libscanprobe.so if missing (hipcc, gfx950)
"""
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

SRC = os.path.join(REPO, "tools", "scan_persist_probe.hip")
LIB = os.path.join(REPO, "tools", "_probe", "libscanprobe.so")


class Params(ctypes.Structure):
    _fields_ = [("act0", ctypes.c_void_p), ("act1", ctypes.c_void_p), ("counters", ctypes.c_void_p),
                ("steps", ctypes.c_int), ("phases", ctypes.c_int), ("G", ctypes.c_int), ("K", ctypes.c_int),
                ("cols", ctypes.c_int), ("mode", ctypes.c_int), ("verify", ctypes.c_int), ("variant", ctypes.c_int),
                ("nbuf", ctypes.c_int), ("ring", ctypes.c_void_p), ("wsrc", ctypes.c_void_p), ("spin_limit", ctypes.c_long)]


def build():
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", SRC, "-o", LIB],
                   check=True)


def main():
    build()
    lib = ctypes.CDLL(LIB)
    lib.probe_launch.restype = ctypes.c_int
    lib.probe_launch.argtypes = [ctypes.POINTER(Params), ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    K, steps, phases = 1024, 64, 5
    act = [torch.zeros(16, K, device=dev), torch.zeros(16, K, device=dev)]
    w = torch.randn(16, K, device=dev) * 0.03
    counters = torch.zeros(64, dtype=torch.int32, device=dev)
    streams = {"whole chip": torch.cuda.Stream()}
    try:
        from dv3hip import engine

        ln = engine.Lanes.get(dev)
        if ln is not None:
            streams["128-CU lane"] = ln.streams["scan"]
    except Exception as e:  # the probe is meaningful without the lanes
        print("no CU-masked lane:", e)

    ring = torch.zeros(steps * phases + 1, 16, K, device=dev)

    def run(G, mode, verify, stream, variant=0, fresh=False):
        counters.zero_()
        act[0].zero_(), act[1].zero_()
        p = Params(act[0].data_ptr(), act[1].data_ptr(), counters.data_ptr(), steps, phases, G, K, K // G, mode, verify,
                   variant, steps * phases + 1 if fresh else 2, ring.data_ptr(), w.data_ptr(), 400000)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            a.record()
            rc = lib.probe_launch(ctypes.byref(p), stream.cuda_stream)
            b.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        c = counters.cpu().numpy()
        return a.elapsed_time(b) * 1e3 / steps, int(c[1]), int(c[2])

    print(f"persistent forward-scan skeleton: {steps} steps x {phases} dependent layers, 16 rows x K = {K}, weights in LDS")
    names = {0: "sc1 stores + sc1 loads", 1: "sc1 stores + sc0 sc1 loads", 2: "sc1 stores, acquire fence + plain loads",
             3: "plain stores + release fence, acquire fence + plain loads"}
    print("hand-off forms (128 workgroups, one counter; stale = values a verify run read that the previous layer had not written):")
    for variant in (0, 1, 2, 3):
        for fresh in (False, True):
            us, err, stale = run(128, 0, 1, streams["whole chip"], variant, fresh)
            ts = sorted(run(128, 0, 0, streams["whole chip"], variant, fresh)[0] for _ in range(5))
            print(f"  {names[variant]:58s} {'fresh addresses every layer' if fresh else 'two buffers in rotation  '} "
                  f"{ts[2]:6.2f} us per step   stale {stale}  error flag {err}")
    if "--forms" in sys.argv:
        return
    VAR = int(sys.argv[sys.argv.index("--variant") + 1]) if "--variant" in sys.argv else 3
    for sname, st in streams.items():
        for G in (64, 128, 256):
            if sname != "whole chip" and G > 128:
                continue  # (a lane holds 128 workgroups at one per CU: more would wait for a CU and the barrier never completes)
            for mode, mname in ((0, "one counter"), (1, "per-XCD counters")):
                us, err, stale = run(G, mode, 1, st, VAR)
                ok = "protocol ok" if (err == 0 and stale == 0) else f"ERROR flag {err}, {stale} stale reads"
                ts = sorted(run(G, mode, 0, st, VAR)[0] for _ in range(7))
                print(f"  {sname:12s} G = {G:3d} workgroups, barrier: {mname:17s} {ts[3]:6.2f} us per step "
                      f"({ts[3] / phases:5.2f} us per layer; min {ts[0]:.2f})   [{ok}; verify run {us:.2f}]")
                if err:
                    return


if __name__ == "__main__":
    main()
