#!/usr/bin/env python3
"""Microbenchmark of the implicit-GEMM convolution kernels on the cfg-2 layer shapes (MI355X only).

    python tools/conv_bench.py [--reps 20] [--frames 1024]

Every case is captured `reps` times into one hipGraph; prints device time per launch and TFLOP/s (useful flops
of the convolution) next to a dense GEMM of the same M x N x K through dv3_gemm_f32 -- the gap between the two
is the cost of the im2col gather.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import torch  # noqa: E402

from dv3hip import ops  # noqa: E402


LANE = None  # --lane: replay on the 128-CU side lane (engine.Lanes)


def graph_us(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
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
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=32, help="cnn_depth: 32 (cfg 2) or 96 (cfg 4 / 5)")
    ap.add_argument("--only", default="", help="substring of the op name (conv_s2, convT_s2, conv_wgrad)")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-GEMM comparison column")
    ap.add_argument("--lane", action="store_true", help="replay on the 128-CU side lane instead of the whole chip")
    args = ap.parse_args()
    if args.lane:
        global LANE
        from dv3hip import engine

        LANE = engine.Lanes.get(torch.device("cuda", 0)).streams["side"]
    N = args.frames
    r = lambda *s: torch.randn(*s, device="cuda")
    # (H = fine grid side, C fine, C coarse): encoder conv fine->coarse; decoder convT coarse->fine
    d = args.depth
    layers = [(64, 3, d), (32, d, 2 * d), (16, 2 * d, 4 * d), (8, 4 * d, 8 * d)]
    print(f"{'op':28s} {'M':>8s} {'N':>5s} {'K':>5s} {'us':>8s} {'TF/s':>7s} | dense GEMM us  TF/s")
    for H, Cf, Cc in layers:
        fine, coarse = r(N, H, H, Cf), r(N, H // 2, H // 2, Cc)
        w = r(Cc, Cf, 4, 4)
        flops = 2.0 * N * (H // 2) ** 2 * Cc * 16 * Cf

        def dense(M, Nn, K, tA=False):
            if args.no_dense:
                return 0.0, 0.0
            A = r(K, M) if tA else r(M, K)
            B = r(K, Nn) if tA else r(Nn, K)
            C = torch.zeros(M, Nn, device="cuda")
            us = graph_us(lambda: ops.gemm(A, B, C, transA=tA, transB=not tA, accumulate=tA), args.reps)
            return us, 2.0 * M * Nn * K / us / 1e6

        if Cf != 3 and (not args.only or args.only in "conv_s2 convT_s2"):
            wp = torch.empty(Cc, 16 * Cf, device="cuda")
            ops.pack_conv_weight(w, wp, transposed=False)
            y = torch.empty_like(coarse)
            us = graph_us(lambda: ops.conv_s2_fwd(fine, wp, y, Ci=Cf, Co=Cc), args.reps)
            M, Nn, K = N * (H // 2) ** 2, Cc, 16 * Cf
            du, dt = dense(M, Nn, K)
            print(f"conv_s2  {H:2d}x{H:<2d} {Cf:3d}->{Cc:<3d}      {M:8d} {Nn:5d} {K:5d} {us:8.1f} {flops / us / 1e6:7.1f} | {du:8.1f} {dt:7.1f}")
            wpt = torch.empty(4, Cf, 4 * Cc, device="cuda")
            ops.pack_conv_weight(w, wpt, transposed=True)
            dx = torch.empty_like(fine)
            us = graph_us(lambda: ops.convT_s2_fwd(coarse, wpt, dx, Ci=Cc, Co=Cf), args.reps)
            M, Nn, K = N * (H // 2) ** 2, Cf, 4 * Cc
            du, dt = dense(M * 4, Nn, K)
            print(f"convT_s2 {H // 2:2d}x{H // 2:<2d} {Cc:3d}->{Cf:<3d} (x4 cls) {M:8d} {Nn:5d} {K:5d} {us:8.1f} {flops / us / 1e6:7.1f} | {du:8.1f} {dt:7.1f}")
        if Cf == 3 and (not args.only or args.only in "conv_s2 convT_s2 c3") and Cc in ops.C3_WIDTHS:
            # the 3-channel image layers: encoder conv 3 -> d, decoder convT d -> 3 (weights in the reference layout)
            y = torch.empty_like(coarse)
            us = graph_us(lambda: ops.conv_s2_c3_fwd(fine, w, y, CW=Cc), args.reps)
            gb = 4.0 * (fine.numel() + coarse.numel()) / 1e9
            print(f"conv_s2_c3  {H:2d}x{H:<2d} 3->{Cc:<3d}     {N * (H // 2) ** 2:8d} {Cc:5d} {48:5d} {us:8.1f} {flops / us / 1e6:7.1f} | {gb / us * 1e6:6.0f} GB/s")
            wt = r(Cc, 3, 4, 4)
            yo = torch.empty(N, H, H, 3, device="cuda")
            us = graph_us(lambda: ops.convT_s2_c3_fwd(coarse, wt, yo, CW=Cc, out_add=0.5), args.reps)
            print(f"convT_s2_c3 {H // 2:2d}x{H // 2:<2d} {Cc:3d}->3      {N * (H // 2) ** 2:8d} {3:5d} {4 * Cc:5d} {us:8.1f} {flops / us / 1e6:7.1f} | {gb / us * 1e6:6.0f} GB/s")
        if args.only and args.only not in "conv_wgrad" and not (args.only == "c3" and Cf == 3):
            continue
        dw = torch.zeros(Cc, Cf, 4, 4, device="cuda")
        us = graph_us(lambda: ops.conv_s2_wgrad(coarse, fine, dw), args.reps)
        M, Nn, K = Cc, 16 * Cf, N * (H // 2) ** 2
        du, dt = dense(M, Nn, K, tA=True)
        print(f"conv_wgrad {H:2d}x{H:<2d} Cf{Cf:<3d} Cc{Cc:<3d}  {M:8d} {Nn:5d} {K:5d} {us:8.1f} {flops / us / 1e6:7.1f} | {du:8.1f} {dt:7.1f}")


if __name__ == "__main__":
    main()
