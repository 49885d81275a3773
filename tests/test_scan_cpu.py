"""tools.static_scan (tools.py:806-850) on the CPU: the fold-and-stack combinator the
reference drives RSSM.observe / imagine_with_action / _imagine with.  Checked (i) structurally against a direct
loop for every output form the reference uses (dict, tuple of dicts / tensors), and (ii) end to end: the observe scan
written exactly as networks.py:127-143 writes it -- static_scan over obs_step with a (state, state) start --
reproduces the posterior / prior the REFERENCE produced for the tiny config (tests/golden/tiny.npz)."""
import os

import numpy as np
import torch

import tools
from oracle import dv3_oracle as O
from tests.golden import common

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_static_scan_output_forms():
    xs = torch.arange(12.0).reshape(4, 3)
    # dict state -> [dict of stacked]
    out = tools.static_scan(lambda last, x: {"s": last["s"] + x, "p": x * 2}, (xs,), {"s": torch.zeros(3), "p": None})
    assert isinstance(out, list) and len(out) == 1
    assert torch.equal(out[0]["s"], torch.cumsum(xs, 0)) and torch.equal(out[0]["p"], xs * 2)
    # tuple of (dict, tensor, dict) with two inputs; start entries may be None (models.py:517 passes (start, None, None))
    ys = torch.ones(4, 3)

    def step(last, x, y):
        prev = last[0]["a"] if last[0] is not None else torch.zeros(3)
        return {"a": prev + x}, x + y, {"b": x - y}

    o = tools.static_scan(step, (xs, ys), (None, None, None))
    assert len(o) == 3 and torch.equal(o[0]["a"], torch.cumsum(xs, 0))
    assert torch.equal(o[1], xs + ys) and torch.equal(o[2]["b"], xs - ys)
    assert o[1].shape == (4, 3)


def test_observe_written_with_static_scan_matches_reference_golden():
    name = "tiny"
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    cfg = common.path_config(name)
    p = {k: torch.from_numpy(v) for k, v in common.make_weights(name).items()}
    n = {k: torch.from_numpy(v) for k, v in common.make_noise(name).items()}
    data = common.make_batch(name)
    obs = O.preprocess(cfg, data)
    with torch.no_grad():
        embed = O.conv_encoder(cfg, p, obs["image"])
    swap = lambda x: x.permute([1, 0] + list(range(2, len(x.shape))))
    T = embed.shape[1]
    steps = torch.arange(T)

    def fn(prev_state, prev_act, emb, first, t):  # networks.py:132-139: prev_state[0] is the posterior
        return O.obs_step(cfg, p, prev_state[0], prev_act, emb, first, n["q_prior"][int(t)], n["q_post"][int(t)])

    with torch.no_grad():
        post, prior = tools.static_scan(fn, (swap(obs["action"]), swap(embed), swap(obs["is_first"]), steps),
                                        (None, None))
    post = {k: swap(v) for k, v in post.items()}
    prior = {k: swap(v) for k, v in prior.items()}
    for k in ("stoch", "deter", "logit"):
        assert np.allclose(post[k].numpy(), g["post/" + k], atol=2e-5), "post " + k
        assert np.allclose(prior[k].numpy(), g["prior/" + k], atol=2e-5), "prior " + k
