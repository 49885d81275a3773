#!/usr/bin/env python3
"""Microbenchmark of the row kernels at the observe-scan shapes (few rows, one launch per step) -- MI355X only.

    python tools/rowops_bench.py [--reps 200]

Each case is captured `reps` times into one hipGraph and replayed, so the figure is device time per launch
(graph-internal dependencies included, host launch cost excluded) -- what the captured update pays.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import torch  # noqa: E402

from dv3hip import ops  # noqa: E402


def graph_us(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    dev = "cuda"
    r = lambda *s: torch.randn(*s, device=dev)
    for M in (16, 32, 1024):
        De = 512
        p, h, dhn = r(M, 3 * De), r(M, De), r(M, De)
        g, b = r(3 * De), r(3 * De)
        mean, rstd = r(M), r(M).abs() + 0.5
        hn, dp, dh = r(M, De), r(M, 3 * De), r(M, De)
        dg, db = torch.zeros(3 * De, device=dev), torch.zeros(3 * De, device=dev)
        print(f"gru_fwd          M={M:5d}: {graph_us(lambda: ops.gru_fwd(p, g, b, h, hn, mean, rstd), args.reps):7.2f} us")
        print(f"gru_bwd          M={M:5d}: {graph_us(lambda: ops.gru_bwd(dhn, p, g, b, h, mean, rstd, dp, dh), args.reps):7.2f} us")
        print(f"gru_bwd +dgamma  M={M:5d}: {graph_us(lambda: ops.gru_bwd(dhn, p, g, b, h, mean, rstd, dp, dh, dg, db), args.reps):7.2f} us")
        for N in (512, 1024):
            x, dy, dx, y = r(M, N), r(M, N), r(M, N), r(M, N)
            ga, be = r(N), r(N)
            dga, dbe = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
            print(f"ln_fwd    N={N:5d} M={M:5d}: {graph_us(lambda: ops.ln_act_fwd(x, ga, be, y, mean, rstd), args.reps):7.2f} us")
            print(f"ln_bwd    N={N:5d} M={M:5d}: {graph_us(lambda: ops.ln_act_bwd(dy, x, ga, be, mean, rstd, dx), args.reps):7.2f} us")
            print(f"ln_bwd+dg N={N:5d} M={M:5d}: {graph_us(lambda: ops.ln_act_bwd(dy, x, ga, be, mean, rstd, dx, dga, dbe), args.reps):7.2f} us")
        SD = 1024
        dsin, ddin, first = r(M, SD), r(M, De), torch.zeros(M, device=dev)
        gs, gd, s0, d0 = r(M, SD), r(M, De), torch.zeros(SD, device=dev), torch.zeros(De, device=dev)
        print(f"obs_blend_bwd    M={M:5d}: {graph_us(lambda: ops.obs_blend_bwd(dsin, ddin, first, gs, gd, s0, d0), args.reps):7.2f} us")
        # skinny GEMMs of the scan
        for (N, K, tB, note) in ((512, 1536, False, "dgrad gru->x"), (1536, 1024, True, "gru fwd"),
                                 (512, 1030, True, "img_in fwd"), (1024, 512, False, "dgrad")):
            A = r(M, K)
            B = r(N, K) if tB else r(K, N)
            C = r(M, N)
            for mode in (False, True, "atomic"):
                us = graph_us(lambda: ops.gemm(A, B, C, transB=tB, accumulate=mode), args.reps)
                print(f"gemm {note:14s} M={M:5d} N={N} K={K} acc={str(mode):6s}: {us:7.2f} us  "
                      f"{2.0 * M * N * K / us / 1e6:6.2f} TF/s")


if __name__ == "__main__":
    main()
