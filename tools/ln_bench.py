#!/usr/bin/env python3
"""Channel LayerNorm + SiLU of the conv stacks (1 M x 32 ... 16 k x 256 rows x channels), forward and backward with
d-gamma / d-beta, on the whole chip and on a 128-CU lane: us per launch inside a hipGraph and GB/s of the tensors that
have to cross HBM (forward: read x, write y; backward: read dy and x, write dx).  MI355X only.

    python tools/ln_bench.py [--reps 20]

The last column is the backward WITHOUT the parameter gradients: the difference is what d-gamma / d-beta cost.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import torch  # noqa: E402

from dv3hip import engine, ops  # noqa: E402


def graph_us(fn, reps, stream):
    fn()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            a.record()
            g.replay()
            b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lanes = engine.Lanes.get(dev)
    streams = [("whole chip", lanes.streams["whole"]), ("128-CU lane", lanes.streams["side"])]
    for R, N in ((1048576, 32), (262144, 64), (65536, 128), (16384, 256), (15360, 512)):
        x, dy = torch.randn(R, N, device=dev), torch.randn(R, N, device=dev)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        g, b = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
        dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        for label, st in streams:
            f = graph_us(lambda: ops.ln_act_fwd(x, g, b, y, mean, rstd), args.reps, st)
            w = graph_us(lambda: ops.ln_act_bwd(dy, x, g, b, mean, rstd, dx, dg, db), args.reps, st)
            w0 = graph_us(lambda: ops.ln_act_bwd(dy, x, g, b, mean, rstd, dx), args.reps, st)
            print(f"{R:8d} x {N:3d}  {label:11s}  fwd {f:7.1f} us = {8.0 * R * N / f / 1e3:6.0f} GB/s   "
                  f"bwd {w:7.1f} us = {12.0 * R * N / w / 1e3:6.0f} GB/s   bwd without d-gamma {w0:7.1f} us = "
                  f"{12.0 * R * N / w0 / 1e3:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
