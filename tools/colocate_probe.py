#!/usr/bin/env python3
"""Can dense work share the scan lane's compute units with the scan?  (MI355X only, development aid.)

The pipelined update keeps both 128-CU lanes full, but the world-model lane spends 4.4 of its 13.2 ms in the two observe
scans, whose 16-row launches need the lane's compute units for their L2 bandwidth and leave their ALUs idle.  This
probe puts a SECOND queue on the same 128 compute units (same CU mask) and runs dense GEMMs there while the chain of
dependent 16-row launches runs on the first one: how much does the chain slow down, how much of their lane speed do the
GEMMs keep -- by LDS footprint of the GEMM tile (a workgroup of the chain can only be placed on a CU that has LDS and wave
slots left: the 128 x 128 tile takes all 160 KB with two workgroups, the 64 x 64 one leaves 37 KB).

    python tools/colocate_probe.py
"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import torch  # noqa: E402

from dv3hip import _lib, engine, ops  # noqa: E402


def same_mask_stream(lanes, lane):
    """A second stream with the CU mask of `lane` (rebuilt the way engine.Lanes builds it)."""
    lib = _lib.load()
    n = ctypes.c_int()
    _lib.check(lib.dv3_device_cu_count(ctypes.byref(n)), "dv3_device_cu_count")
    n_cu = n.value
    groups, want = n_cu // 8, lanes.cus["scan"] // 8
    scan_bits = [((g + 1) * want) // groups != (g * want) // groups for g in range(groups)]
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(n_cu):
        if scan_bits[i // 8] == (lane == "scan"):
            mask[i // 32] |= 1 << (i % 32)
    out = ctypes.c_ulonglong()
    _lib.check(lib.dv3_stream_create_cu_masked(words, mask, ctypes.byref(out)), "dv3_stream_create_cu_masked")
    ops.LANE_STREAMS[out.value] = lanes.cus[lane]
    return torch.cuda.ExternalStream(out.value), out.value


def main():
    dev = torch.device("cuda", 0)
    lanes = engine.Lanes.get(dev)
    X = lanes.streams["scan"]
    X2, h2 = same_mask_stream(lanes, "scan")
    Y = lanes.streams["side"]
    home = torch.cuda.Stream()
    torch.cuda.set_stream(home)
    x = torch.randn(16, 1024, device=dev)
    W = torch.randn(1536, 1024, device=dev) * 0.03
    W2 = torch.randn(1024, 1536, device=dev) * 0.03
    y = torch.empty(16, 1536, device=dev)

    def chain(n=150):
        for _ in range(n):
            ops.gemm(x, W, y, transB=True)
            ops.gemm(y, W2, x, transB=True)

    chain(2)
    torch.cuda.synchronize()
    gc = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gc, stream=X):
        chain()
    A = torch.randn(14336, 512, device=dev)
    B = torch.randn(512, 512, device=dev)
    C = torch.empty(14336, 512, device=dev)
    A1 = torch.randn(1024, 1024, device=dev)
    B1 = torch.randn(1536, 1024, device=dev)
    C1 = torch.empty(1024, 1536, device=dev)

    def timed_pair(ga, sa, gb, sb):
        """Both graphs at once -> (ms of a, ms of b)."""
        ea, eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)], [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        go = torch.cuda.Event()
        go.record(home)
        sa.wait_event(go)
        ea[0].record(sa)
        if gb is not None:
            sb.wait_event(go)
            eb[0].record(sb)
        with torch.cuda.stream(sa):
            ga.replay()
        ea[1].record(sa)
        if gb is not None:
            with torch.cuda.stream(sb):
                gb.replay()
            eb[1].record(sb)
        torch.cuda.synchronize()
        return ea[0].elapsed_time(ea[1]), (eb[0].elapsed_time(eb[1]) if gb is not None else None)

    t_chain = min(timed_pair(gc, X, None, None)[0] for _ in range(3))
    print(f"chain of 300 dependent 16-row GEMMs on the scan lane, alone: {t_chain:.3f} ms = {t_chain * 1e3 / 300:.2f} us per launch")
    for label, (a, b, c), tiles in (("14336 x 512 x 512", (A, B, C), (9, 13, 14, 15)), ("1024 x 1536 x 1024", (A1, B1, C1), (9, 12, 13, 14))):
        for tile in tiles:
            n = 30 if a.shape[0] > 2048 else 60
            fn = lambda: [ops.gemm(a, b, c, transB=True, tile=tile) for _ in range(n)]
            fn()
            torch.cuda.synchronize()
            gd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gd, stream=X2):
                fn()
            t_alone = min(timed_pair(gd, X2, None, None)[0] for _ in range(3))
            res = [timed_pair(gc, X, gd, X2) for _ in range(3)]
            tc, td = min(r_[0] for r_ in res), min(r_[1] for r_ in res)
            res_y = [timed_pair(gc, X, gd, Y) for _ in range(3)]
            tcy, tdy = min(r_[0] for r_ in res_y), min(r_[1] for r_ in res_y)
            # while both run: the chain takes tc; the dense graph runs beside it for min(tc, td) of its td
            print(f"  {label} tile {tile:2d}: {n} GEMMs alone on the lane {t_alone:.3f} ms | on the SAME 128 CUs as the chain: chain "
                  f"{tc:.3f} ms (x{tc / t_chain:.2f}), GEMMs {td:.3f} ms (x{td / t_alone:.2f}) | on the OTHER lane: chain {tcy:.3f} "
                  f"(x{tcy / t_chain:.2f}), GEMMs {tdy:.3f} (x{tdy / t_alone:.2f})", flush=True)
    _lib.load().dv3_stream_destroy(h2)


if __name__ == "__main__":
    main()
