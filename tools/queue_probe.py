#!/usr/bin/env python3
"""Which hardware queue does a regular HIP stream land on once CU-masked streams exist?  (MI355X only, development aid.)

    python tools/queue_probe.py [--prefill K]

ROCclr keeps one pool of HSA queues per priority and hands regular streams a pooled queue once the pool is "full"; the queues
of CU-masked streams are in that pool too.  A regular stream that lands on a masked queue runs on that queue's CUs (and in
its order).  The probe times a chip-filling product on freshly made torch streams: 1.0 = whole chip, 2.0 = a 128-CU queue.
--prefill K: touch K regular streams BEFORE the masked streams are made.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import torch  # noqa: E402

from dv3hip import engine, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prefill", type=int, default=0)
    ap.add_argument("--after", type=int, default=10)
    args = ap.parse_args()
    A = torch.randn(4096, 2048, device="cuda")
    B = torch.randn(4096, 2048, device="cuda")
    C = torch.empty(4096, 4096, device="cuda")

    def timed(stream):
        best = 1e9
        with torch.cuda.stream(stream):
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                for _ in range(4):
                    ops.gemm(A, B, C, transB=True)
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
        return best

    base = timed(torch.cuda.current_stream())
    pre = [torch.cuda.Stream() for _ in range(args.prefill)]
    print("before the masked streams:", " ".join(f"{timed(s) / base:.2f}" for s in pre))
    ln = engine.Lanes.get("cuda:0")
    print("lanes:", {k: f"{timed(s) / base:.2f}" for k, s in ln.streams.items()})
    print("the same regular streams afterwards:", " ".join(f"{timed(s) / base:.2f}" for s in pre))
    post = [torch.cuda.Stream() for _ in range(args.after)]
    print("regular streams made afterwards:", " ".join(f"{timed(s) / base:.2f}" for s in post))
    print("NULL stream:", f"{timed(torch.cuda.default_stream()) / base:.2f}")


if __name__ == "__main__":
    main()
