#!/usr/bin/env python3
"""Microbenchmark of dv3_gemm_f32 on the shapes the hot path issues (MI355X only).

    python tools/gemm_bench.py [--reps 50]

Prints, per (shape, layout, tile): average device time per launch (`reps` launches captured into one hipGraph, HIP events around its replay)
and TFLOP/s against the 157.3 TFLOP/s fp32-MFMA peak.  Development aid for csrc/mfma_gemm.h.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import torch  # noqa: E402

from dv3hip import ops  # noqa: E402

SHAPES = [
    # M, N, K, transA, transB, note
    (1024, 512, 1536, 0, 1, "actor/head L0 fwd"),
    (1024, 1536, 1024, 0, 1, "GRU fwd"),
    (1024, 512, 512, 0, 1, "img_out / L1 fwd"),
    (1024, 512, 1030, 0, 1, "img_in fwd"),
    (1024, 1024, 512, 0, 1, "stat layer fwd"),
    (1024, 512, 1536, 0, 0, "dgrad gru->x"),
    (1024, 1024, 512, 0, 0, "dgrad"),
    (15360, 512, 1536, 0, 1, "behaviour head L0"),
    (14336, 512, 512, 0, 0, "behaviour dgrad"),
    (512, 512, 14336, 1, 0, "wgrad"),
    (512, 1536, 15360, 1, 0, "wgrad head L0"),
    (1536, 1024, 1024, 1, 0, "wgrad GRU"),
    (1024, 512, 1024, 1, 0, "wgrad stat"),
    (512, 4608, 1024, 1, 0, "wgrad obs_out"),
    (255, 512, 14336, 1, 0, "wgrad reward head"),
    (1024, 512, 512, 0, 0, "imagine dgrad"),
    (1024, 1536, 1024, 0, 0, "imagine dgrad gru"),
    (4096, 4096, 4096, 0, 1, "square 4k"),
    (15360, 512, 512, 0, 1, "beh: head L1 fwd"),
    (14336, 512, 512, 0, 0, "beh: head dgrad [K][N]"),
    (14336, 1024, 512, 0, 0, "beh: head dgrad to stoch [K][N]"),
    (14336, 1024, 512, 0, 1, "beh: head dgrad to stoch, W^T"),
    (15360, 255, 512, 0, 1, "beh: head out fwd"),
    (2048, 3072, 1536, 0, 1, "big: cfg3 GRU fwd"),
    (2048, 1024, 1024, 0, 1, "big: cfg3 stacked"),
    (4096, 1024, 1024, 0, 1, "big: cfg5 hidden"),
    (4096, 2048, 2048, 0, 1, "big: cfg5 stacked"),
    (4096, 6144, 3072, 0, 1, "big: cfg5 GRU fwd"),
    (4096, 12288, 5120, 0, 1, "big: cfg4 GRU fwd"),
    (4096, 5120, 12288, 0, 1, "big: cfg4 GRU dgrad (W^T)"),
]


ZEROS = False
LANE = None  # --lane: replay on the 128-CU side lane (engine.Lanes) -- what a kernel sees in the pipelined update


def bench(M, N, K, tA, tB, tile, reps):
    dev = "cuda"
    pad = int(os.environ.get("DV3_BENCH_PAD", "0"))  # extra floats per row (leading-dimension experiment)
    A = torch.randn((K, M + pad) if tA else (M, K + pad), device=dev)[:, :(M if tA else K)]
    B = torch.randn((N, K + pad) if tB else (K, N + pad), device=dev)[:, :(K if tB else N)]
    if ZEROS:  # clock check: zero operands draw less power, the chip holds a higher clock (MI355X_MICROARCH, DVFS)
        A.zero_()
        B.zero_()
    C = torch.zeros(M, N, device=dev)
    acc = bool(tA)
    fn = lambda: ops.gemm(A, B, C, transA=bool(tA), transB=bool(tB), tile=tile, accumulate=acc)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    # `reps` launches captured into one hipGraph: device time per launch, as the captured update pays it
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(LANE if LANE is not None else torch.cuda.current_stream()):
        a.record()
        g.replay()
        b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--tiles", default="0,1")
    ap.add_argument("--only", default="", help="substring of the note column")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands (clock / power check)")
    ap.add_argument("--lane", action="store_true", help="replay on the 128-CU side lane instead of the whole chip")
    ap.add_argument("--shapes", default="", help='"M,N,K,tA,tB;..." instead of the built-in list')
    args = ap.parse_args()
    shapes = SHAPES
    if args.shapes:
        shapes = [tuple(int(x) for x in sh.split(",")) + ("",) for sh in args.shapes.split(";") if sh]
    global ZEROS, LANE
    ZEROS = args.zeros
    if args.lane:
        from dv3hip import engine

        LANE = engine.Lanes.get(torch.device("cuda", 0)).streams["side"]
    tiles = [int(t) for t in args.tiles.split(",")]
    print(f"{'shape':>22s} {'tA tB':>6s} {'tile':>5s} {'us':>9s} {'TFLOP/s':>8s} {'frac':>6s}  note")
    for M, N, K, tA, tB, note in shapes:
        if args.only and args.only not in note:
            continue
        for t in tiles:
            if (M * N * K > 1e11 and t == 1) or (t == 9 and tA) or (t == 10 and not tA):
                continue
            try:
                us, tf = bench(M, N, K, tA, tB, t, args.reps)
            except Exception:  # the tile does not take this shape / layout
                continue
            print(f"{M:6d}x{N:5d}x{K:6d} {tA:3d}{tB:3d} {t:5d} {us:9.1f} {tf:8.1f} {tf / 157.3:6.2f}  {note}", flush=True)


if __name__ == "__main__":
    main()
