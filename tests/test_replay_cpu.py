"""Replay sampler against batches drawn by the reference's own tools.sample_episodes /
tools.from_generator (tests/golden/replay.npz, written by tests/golden/make_replay_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from tests.golden import common

GOLD = os.path.join(os.path.dirname(__file__), "golden", "replay.npz")


@pytest.mark.parametrize("seed", [0, 3])
def test_sampler_draws_the_reference_batches(seed):
    import tools  # dreamerv3-torch_amd/tools.py (conftest puts the package dir on sys.path)

    g = np.load(GOLD, allow_pickle=False)
    eps = common.make_episodes()
    gen = tools.from_generator(tools.sample_episodes(eps, common.REPLAY_LENGTH, seed=seed), common.REPLAY_BATCH)
    for i in range(3):
        b = next(gen)
        assert sorted(b.keys()) == list(g[f"s{seed}/b{i}/keys"])
        assert b["image"].dtype == np.uint8 and b["is_first"].dtype == np.bool_
        assert b["reward"].shape == (common.REPLAY_BATCH, common.REPLAY_LENGTH)
        np.testing.assert_array_equal(b["reward"], g[f"s{seed}/b{i}/reward"])        # which episode / offset
        np.testing.assert_array_equal(b["is_first"], g[f"s{seed}/b{i}/is_first"])    # joins and forced starts
        np.testing.assert_array_equal(b["image"].astype(np.int64).sum((2, 3, 4)), g[f"s{seed}/b{i}/image_sum"])
        assert b["is_first"][:, 0].all()


def test_ranks_draw_different_sequences():
    import tools

    eps = common.make_episodes()
    a = next(tools.sample_episodes(eps, common.REPLAY_LENGTH, seed=0))
    b = next(tools.sample_episodes(eps, common.REPLAY_LENGTH, seed=1))
    assert not np.array_equal(a["reward"], b["reward"])


def test_short_store_is_joined_and_single_step_episodes_are_skipped():
    import tools

    eps = {"a": {"reward": np.zeros(1, np.float32), "is_first": np.ones(1, bool)},
           "b": {"reward": np.arange(3, dtype=np.float32) + 10, "is_first": np.array([True, False, False])}}
    s = next(tools.sample_episodes(eps, 8, seed=0))
    assert s["reward"].shape == (8,) and (s["reward"] >= 10).all()
    # after the first (possibly partial) piece every join restarts episode "b" at its first step
    starts = np.flatnonzero(s["is_first"])
    assert starts[0] == 0 and all(s["reward"][j] == 10 for j in starts[1:])
