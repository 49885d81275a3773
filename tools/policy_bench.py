#!/usr/bin/env python3
"""Latency of one acting step (Dreamer._policy: encoder -> obs_step -> actor) at cfg-2 sizes.

    python tools/policy_bench.py [--envs 1 4 16]

Prints the host-visible time per call (launch + device + the final D2H of the action): hipGraph replay
(dv3hip.graph.PolicyRunner, the default of Dreamer._policy) and eager.
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
    ap.add_argument("--reps", type=int, default=50)
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
            return float(np.median(ts)) * 1e3, float(np.max(ts)) * 1e3

        def graph_step():
            nonlocal state
            out, state = agent._policy(obs, state, training=True)
            return out["action"].cpu()  # what the env loop needs back

        def eager_step():
            nonlocal state
            out, state = agent._policy_eager(obs, state, training=True)
            return out["action"].cpu()

        g, gmax = med(graph_step)
        e, emax = med(eager_step)
        print(f"envs={E:3d}: median {g:6.3f} ms per acting step with hipGraph replay (max {gmax:.2f}), {e:6.3f} ms eager "
              f"(max {emax:.2f}); incl. H2D of the image and D2H of the action")


if __name__ == "__main__":
    main()
