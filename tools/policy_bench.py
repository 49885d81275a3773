#!/usr/bin/env python3
"""Latency of one acting step (Dreamer._policy: encoder -> obs_step -> actor) at cfg-2 sizes.

    python tools/policy_bench.py [--envs 1 4 16]

Prints the host-visible time per call (launch + device + the final D2H of the action): hipGraph replay
(dv3hip.graph.PolicyRunner, the default of Dreamer._policy) and eager.  Mean and p99 beside the median: under a CPU quota
(cgroup cpu.max) a host-side torch op that goes OpenMP-parallel makes one step in twelve take the throttling period
(~100 ms) -- r04 found the acting step's pinned-buffer fill doing that (mean 5 ms at a median of 0.29).
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests import helpers as Hh  # noqa: E402
from tests.golden import common  # noqa: E402


class _Logger:
    step = 0

    def scalar(self, *a):
        pass

    def video(self, *a, **k):
        pass

    def write(self, fps=False):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[1, 4, 16])
    ap.add_argument("--reps", type=int, default=1000)
    args = ap.parse_args()
    import dreamer

    name = "cfg2"
    cfg = Hh.make_config(name)
    cfg.pretrain = 0

    def ds():
        while True:
            yield common.make_batch(name)

    agent = dreamer.Dreamer(Hh.obs_space(name), None, cfg, _Logger(), ds()).to(cfg.device)
    agent.requires_grad_(False)
    rs = np.random.RandomState(0)
    for E in args.envs:
        obs = {"image": rs.randint(0, 256, (E, 64, 64, 3)).astype(np.uint8), "is_first": np.zeros((E,), bool),
               "is_terminal": np.zeros((E,), bool)}
        first = dict(obs, is_first=np.ones((E,), bool))
        out, state = agent._policy(first, None, training=True)
        for _ in range(5):
            out, state = agent._policy(obs, state, training=True)
        torch.cuda.synchronize()

        def med(fn):
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            ts = np.asarray(ts) * 1e3
            return float(np.median(ts)), float(ts.mean()), float(np.percentile(ts, 99)), float(ts.max())

        def graph_step():
            nonlocal state
            out, state = agent._policy(obs, state, training=True)
            return out["action"].cpu()  # what the env loop needs back

        def eager_step():
            nonlocal state
            out, state = agent._policy_eager(obs, state, training=True)
            return out["action"].cpu()

        g, e = med(graph_step), med(eager_step)
        fmt = lambda r: f"median {r[0]:6.3f} mean {r[1]:6.3f} p99 {r[2]:6.3f} max {r[3]:7.2f} ms"
        print(f"envs={E:3d}: hipGraph replay {fmt(g)} | eager {fmt(e)}   (per acting step, incl. H2D of the image and D2H "
              f"of the action; {args.reps} steps)")


if __name__ == "__main__":
    main()
