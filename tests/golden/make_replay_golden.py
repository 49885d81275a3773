#!/usr/bin/env python3
"""Golden batches of the reference's replay sampler (tools.sample_episodes + tools.from_generator,
tools.py:310-371) on a synthetic episode store -> tests/golden/replay.npz.

Runs in the build container only (imports /root/reference/tools.py with the tensorboard stub of SURVEY 8c).
Every transition carries a unique id (reward = 1000*episode + index), so the stored `reward` and `is_first`
arrays pin which episode / offset every draw picked and where sequences were joined.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import common  # noqa: E402


def main():
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    sys.path.insert(0, "/root/reference")
    import tools  # the reference's

    out = {}
    for seed in (0, 3):
        eps = common.make_episodes()
        gen = tools.from_generator(tools.sample_episodes(eps, common.REPLAY_LENGTH, seed=seed), common.REPLAY_BATCH)
        with contextlib.redirect_stdout(io.StringIO()):  # the fork prints a counter per sample
            for i in range(3):
                b = next(gen)
                assert "log_entropy" not in b
                out[f"s{seed}/b{i}/reward"] = b["reward"]
                out[f"s{seed}/b{i}/is_first"] = b["is_first"]
                out[f"s{seed}/b{i}/image_sum"] = b["image"].astype(np.int64).sum((2, 3, 4))
                out[f"s{seed}/b{i}/keys"] = np.array(sorted(b.keys()))
    path = os.path.join(HERE, "replay.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] wrote {path}: {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
