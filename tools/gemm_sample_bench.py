#!/usr/bin/env python3
"""Device time of the prior-logit GEMM with and without the fused sampling epilogue (MI355X only): back-to-back launches
captured into one hipGraph, HIP events around the replay."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import torch  # noqa: E402

from dv3hip import ops  # noqa: E402


def graph_us(fn, reps=30):
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
    M, N, K = 1024, 1024, 512
    A, W, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / 20, torch.randn(N, device="cuda")
    lg, st = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    q = torch.empty(M, N, device="cuda").exponential_()
    idx = torch.empty(M * N // 32, dtype=torch.int32, device="cuda")
    rng = ops.RngStream("cuda", 1)
    print(f"gemm only            {graph_us(lambda: ops.gemm(A, W, lg, bias=b, tile=9)):7.2f} us")
    print(f"sample only (rng)    {graph_us(lambda: ops.onehot_sample(lg.view(M, N // 32, 32), st.view(M, N // 32, 32), rng=rng, idx=idx)):7.2f} us")
    print(f"sample only (noise)  {graph_us(lambda: ops.onehot_sample(lg.view(M, N // 32, 32), st.view(M, N // 32, 32), noise=q.view(M, N // 32, 32), idx=idx)):7.2f} us")
    print(f"gemm+sample (mode)   {graph_us(lambda: ops.gemm_sample(A, W, lg, st, bias=b, mode=True, idx=idx)):7.2f} us")
    print(f"gemm+sample (noise)  {graph_us(lambda: ops.gemm_sample(A, W, lg, st, bias=b, noise=q, idx=idx)):7.2f} us")
    print(f"gemm+sample (rng)    {graph_us(lambda: ops.gemm_sample(A, W, lg, st, bias=b, rng=rng, idx=idx)):7.2f} us")


if __name__ == "__main__":
    main()
