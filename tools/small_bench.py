#!/usr/bin/env python3
"""Device time of the small single-workgroup kernels on the behaviour path (quantile select, logged statistics), each
captured `reps` times into one hipGraph (MI355X only).   python tools/small_bench.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import torch  # noqa: E402

from dv3hip import ops  # noqa: E402


def graph_us(fn, reps=50):
    for _ in range(3):
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
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    for n in (14336, 28672, 57344, 458752):
        x = 1.0 + 0.05 * torch.randn(n, device="cuda")
        ema = torch.zeros(2, device="cuda")
        print(f"quantile2_ema n={n:7d}: {graph_us(lambda: ops.quantile2_ema(x, 0.05, 0.95, ema=ema, alpha=0.01)):7.1f} us")
    xs = [torch.randn(n, device="cuda") for n in (14336, 14336, 15360, 92160, 14336)]
    out = torch.empty(5, 4, device="cuda")
    one = torch.empty(4, device="cuda")
    print(f"tensorstats x5 single launches: {graph_us(lambda: [ops.tensorstats(x, one) for x in xs]):7.1f} us")
    print(f"tensorstats_multi (5 tensors):  {graph_us(lambda: ops.tensorstats_multi([(x, None, None) for x in xs], out)):7.1f} us")


if __name__ == "__main__":
    main()
