#!/usr/bin/env python3
"""Port-to-reference CPU ratio -- build container only.

Times, on the same host cores and the same synthetic minibatch, (a) the REFERENCE's own
WorldModel._train + ImagBehavior._train (imported from /root/reference exactly as tests/golden/make_golden.py does)
and (b) the CPU oracle's full update (tests/helpers.oracle_update, the `cpu_baseline` of bench.py).  The ratio
T(reference) / T(oracle) turns the oracle time measured on the GPU box's host into a derived reference-CPU figure
(the reference's Python never travels there).  Prints one JSON line; the numbers are recorded in BASELINE.md.

    python tools/cpu_ratio.py [cfg2] [--reps 3]
"""
import contextlib
import io
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "cfg2"
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    from tests import helpers as Hh
    from tests.golden import common
    from tests.golden import make_golden as MG

    # (b) oracle
    Hh.oracle_update(name)
    t_or = []
    for _ in range(reps):
        t0 = time.perf_counter()
        Hh.oracle_update(name)
        t_or.append(time.perf_counter() - t0)
    # (a) reference
    tools, networks, models = MG.import_reference()
    torch.autograd.set_detect_anomaly(False)  # dreamer.py:30 turns it on; timed without
    cfg, wm, beh, w = MG.build_reference(name, tools, networks, models)
    data = common.make_batch(name)
    reward_fn = lambda f, st, a: wm.heads["reward"](wm.dynamics.get_feat(st)).mode()
    quiet = contextlib.redirect_stdout(io.StringIO())

    def ref_update():
        with quiet:
            post, _, _ = wm._train({k: v.copy() for k, v in data.items()})
            beh._train(post, reward_fn)

    ref_update()
    t_ref = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ref_update()
        t_ref.append(time.perf_counter() - t0)
    s = common.SHAPES[name]
    units = s["B"] * s["T"] * s["H"]
    out = {"config": name, "threads": threads, "reps": reps, "T_reference_s": float(np.median(t_ref)),
           "T_oracle_s": float(np.median(t_or)), "ratio_reference_over_oracle": float(np.median(t_ref) / np.median(t_or)),
           "reference_steps_per_s": units / float(np.median(t_ref)), "oracle_steps_per_s": units / float(np.median(t_or))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
