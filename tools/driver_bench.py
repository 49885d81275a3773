#!/usr/bin/env python3
"""Time per update of a driver written against the reference's classes (its Dreamer._train, dreamer.py:192-199:
WorldModel._train(host batch) then ImagBehavior._train(post, reward-head lambda)) on the MI355X modules, with the
hipGraph replay behind the two calls on and off (config key hip_graph).  MI355X only.

    python tools/driver_bench.py [cfg2] [--steps 20]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from tests import helpers as Hh  # noqa: E402
from tests.golden import common  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "cfg2"
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20
    data = common.make_batch(name)
    for hip_graph in (True, False):
        cfg, wm, beh = Hh.build_models(name)
        cfg.hip_graph = hip_graph
        reward = lambda f, s, a: wm.heads["reward"](wm.dynamics.get_feat(s)).mode()  # noqa: E731  (dreamer.py:196-198)

        def update():
            post, context, mets = wm._train(data)
            return mets, beh._train(post, reward)[-1]

        for _ in range(5):
            update()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m1, m2 = update()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        print(f"{name}: hip_graph={hip_graph}: {dt:.2f} ms per update (host batch staged every call; model_loss "
              f"{float(m1['model_loss']):.2f}, actor_loss {float(m2['actor_loss']):.4f})")


if __name__ == "__main__":
    main()
