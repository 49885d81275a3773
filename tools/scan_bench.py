#!/usr/bin/env python3
"""Device time of the observe scan (forward) inside a hipGraph, per step, on cfg-2 shapes (MI355X only).

    python tools/scan_bench.py

Measured r01: 2.69 ms = 42 us per step for 9 dependent launches; leaving the two LayerNorm launches out of the
loop (wrong values, timing only) gave 2.31 ms, the reset blend 2.53 ms, both 2.14 ms -- the ceiling for fusing them
into their neighbours.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from tests import helpers as Hh  # noqa: E402
from tests.golden import common  # noqa: E402


def main():
    name = "cfg2"
    _, wm, _ = Hh.build_models(name)
    s = common.SHAPES[name]
    B, T = s["B"], s["T"]
    eng = wm.dynamics.engine
    E = eng.E
    embed = torch.randn(T, B, E, device="cuda")
    action = torch.randn(T, B, s["A"], device="cuda")
    first = torch.zeros(T, B, device="cuda")
    rng = wm.dynamics._rng()

    def run():
        eng.observe_fwd(embed, action, first, rng=rng)
        rng.commit()

    run()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            run()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"observe_fwd {ts[len(ts) // 2]:.3f} ms "
          f"= {ts[len(ts) // 2] * 1e3 / T:.1f} us per step (incl. the batched prior head and init state)")


if __name__ == "__main__":
    main()
