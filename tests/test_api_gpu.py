"""Public (reference-surface) methods around the hot path that the training kernels do not go through themselves:
tools.lambda_return, RSSM.observe with a carried state, tools.Optimizer.__call__, the `objective` argument of
ImagBehavior._train.  Each against the CPU oracle / torch on the same inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import dv3_oracle as O
from tests import helpers as Hh
from tests.golden import common

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_lambda_return_public_api_matches_oracle_and_reference():
    """tools.lambda_return(reward[1:], value[:-1], disc[1:], bootstrap=value[-1], lambda_, axis=0) as
    models.py:627-634 calls it, on the imagined trajectory of the tiny config."""
    import tools

    name = "tiny"
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    exp = Hh.oracle_update(name, piecewise=True)
    b = exp["beh0"]
    r, v, d = b["reward"].cuda(), b["value"].cuda(), b["discount"].cuda()
    got = tools.lambda_return(r[1:], v[:-1], d[1:], bootstrap=v[-1], lambda_=0.95, axis=0)
    assert got.shape == b["target"].shape
    assert torch.allclose(got.cpu(), b["target"], atol=1e-5)
    assert np.allclose(got.cpu().numpy(), g["imag/target"], atol=2e-5)


@pytest.mark.parametrize("name", ["tiny", "cfg2"])
def test_observe_with_carried_state_matches_oracle(name):
    """RSSM.observe(embed, action, is_first, state) (networks.py:127-143): step 0 continues from `state` on rows
    whose is_first is 0 and resets the others."""
    cfg, wm, _ = Hh.build_models(name)
    s = common.SHAPES[name]
    pc = common.path_config(name)
    p = {k: torch.from_numpy(v) for k, v in common.make_weights(name).items()}
    rs = np.random.RandomState(3)
    B, T, E = min(s["B"], 5), 4, wm.embed_size
    embed = torch.from_numpy(rs.randn(B, T, E).astype(np.float32))
    action = torch.from_numpy(rs.uniform(-1, 1, (B, T, s["A"])).astype(np.float32))
    first = torch.zeros(B, T)
    first[1, 0] = 1.0  # one row resets at step 0, one mid-sequence
    first[2, 2] = 1.0
    S, D = s["stoch"], s["discrete"]
    idx = rs.randint(0, D, (B, S))
    state = {"stoch": torch.nn.functional.one_hot(torch.from_numpy(idx), D).float(),
             "deter": torch.from_numpy(rs.randn(B, s["deter"]).astype(np.float32)) * 0.5,
             "logit": torch.from_numpy(rs.randn(B, S, D).astype(np.float32))}
    q1 = torch.from_numpy(np.maximum(rs.exponential(size=(T, B, S, D)), 1e-20).astype(np.float32))
    q2 = torch.from_numpy(np.maximum(rs.exponential(size=(T, B, S, D)), 1e-20).astype(np.float32))
    post, prior = wm.dynamics.observe(embed.cuda(), action.cuda(), first.cuda(), {k: v.cuda() for k, v in state.items()},
                                      noise=dict(q_prior=q1.cuda(), q_post=q2.cuda()))
    with torch.no_grad():
        epost, eprior = O.observe(pc, p, embed, action, first, q1, q2, state=state)
    assert torch.equal(post["stoch"].cpu(), epost["stoch"]), "posterior draws"
    for k in ("deter", "logit"):
        assert torch.allclose(post[k].cpu(), epost[k], atol=1e-4), "post " + k
        assert torch.allclose(prior[k].cpu(), eprior[k], atol=1e-4), "prior " + k
    # and without a state the call still starts every row from the initial state
    post0, _ = wm.dynamics.observe(embed.cuda(), action.cuda(), first.cuda(), noise=dict(q_prior=q1.cuda(), q_post=q2.cuda()))
    with torch.no_grad():
        e0, _ = O.observe(pc, p, embed, action, first, q1, q2)
    assert torch.allclose(post0["deter"].cpu(), e0["deter"], atol=1e-4)


def test_optimizer_call_with_an_autograd_loss_is_clip_plus_adam():
    """The reference's Optimizer.__call__(loss, params) (tools.py:760-776) for a loss built with torch autograd."""
    import tools

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3)).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3)).cuda()
    ref.load_state_dict(net.state_dict())
    opt = tools.Optimizer("test", list(net.parameters()), lr=1e-2, eps=1e-5, clip=1.0)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2, eps=1e-5)
    x = torch.randn(11, 7, device="cuda")
    for _ in range(3):
        loss = (net(x) ** 2).sum() * 10
        mets = opt(loss, net.parameters())
        rloss = (ref(x) ** 2).sum() * 10
        ropt.zero_grad()
        rloss.backward()
        norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        ropt.step()
        assert float(mets["test_grad_norm"]) == pytest.approx(float(norm), rel=1e-4)
        assert float(mets["test_loss"]) == pytest.approx(float(rloss), rel=1e-5)
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, atol=2e-6), (a - b).abs().max()
    with pytest.raises(RuntimeError, match="no autograd graph"):
        opt(torch.zeros((), device="cuda"), net.parameters())


def test_objective_and_policy_arguments():
    """ImagBehavior._train(start, objective) / _imagine(start, policy, horizon) take what the reference's take
    (models.py:327-331, 448): the reward-head lambda of dreamer.py:196-199 is recognised and stays on the fused path; any
    other objective trains through the autograd hook (tests/test_autograd_gpu.py pins the numbers against the reference's
    Plan2Explore); a foreign policy rolls out through the public methods."""
    name = "tiny"
    s = common.SHAPES[name]
    cfg, wm, beh = Hh.build_models(name)
    n = {k: torch.from_numpy(v).cuda() for k, v in common.make_noise(name).items()}
    post, _, _ = wm._train(common.make_batch(name), noise=dict(q_prior=n["q_prior"], q_post=n["q_post"]))
    post = {k: v.clone() for k, v in post.items()}
    reward = lambda f, s_, a: wm.heads["reward"](wm.dynamics.get_feat(s_)).mode()  # dreamer.py:196-198
    out = beh._train(post, reward)
    assert np.isfinite(float(out[-1]["actor_loss"])) and list(beh._objective_kinds.values()) == [True]
    # an objective of the imagined deter and action: trained through its own autograd graph
    out = beh._train(post, lambda f, s_, a: s_["deter"].pow(2).mean(-1, keepdim=True) + a.sum(-1, keepdim=True))
    assert np.isfinite(float(out[-1]["actor_loss"])) and np.isfinite(float(out[-1]["actor_grad_norm"]))
    assert sorted(beh._objective_kinds.values()) == [False, True]
    with pytest.raises(ValueError, match="one reward per imagined state"):
        beh._train(post, lambda f, s_, a: torch.ones(3, device=f.device))
    feats, states, actions = beh._imagine(post, lambda feat: beh.actor(feat), 3)
    N = s["B"] * s["T"]
    assert feats.shape[:2] == (3, N) and actions.shape == (3, N, s["A"]) and states["stoch"].shape[:2] == (3, N)
    # as in the reference (models.py:513-517) the feats are detached; the states keep their graph
    import tools

    with tools.RequiresGrad(wm.dynamics):
        feats, states, actions = beh._imagine(post, lambda feat: beh.actor(feat), 3)
        assert not feats.requires_grad and states["deter"].requires_grad
        w = wm.dynamics._cell.layers.GRU_linear.weight
        (gw,) = torch.autograd.grad(states["deter"][-1].pow(2).sum(), [w])
    assert float(gw.abs().max()) > 0.0



@pytest.mark.parametrize("name", ["tiny", "cfg2"])
def test_a_driver_written_against_the_reference_classes_replays_graphs(name):
    """The reference's Dreamer._train (dreamer.py:192-199) calls WorldModel._train(data) and then
    ImagBehavior._train(post, reward-head lambda).  Against these modules the two calls meet in one UpdateRunner: after
    two eager warm-up updates they replay hipGraphs (per-lane graphs of the world model, the behaviour, the optimizers)
    and compute what the eager calls compute; the metrics handed back are snapshots (a later update does not rewrite
    them); `hip_graph: False` keeps every call eager."""
    import tools

    def drive(hip_graph, n=5):
        cfg, wm, beh = Hh.build_models(name)
        cfg.hip_graph = hip_graph
        tools.default_rng("cuda:0", seed=21)
        reward = lambda f, s_, a: wm.heads["reward"](wm.dynamics.get_feat(s_)).mode()  # dreamer.py:196-198
        seen = []
        for i in range(n):
            data = common.make_batch(name)  # host arrays, as the replay sampler hands them over
            post, context, mets = wm._train(data)
            out = beh._train(post, reward)
            assert set(post) == {"stoch", "deter", "logit"} and len(out) == 5
            seen.append((mets["model_loss"], out[-1]["actor_loss"], out[-1]["value_loss"]))
        torch.cuda.synchronize()
        import models

        r = models.runner_of(wm)
        vals = [tuple(float(v) for v in t) for t in seen]
        return r, vals, wm.dynamics.W.detach().clone()

    r1, v1, w1 = drive(True)
    r0, v0, w0 = drive(False)
    assert r0 is None and r1 is not None and r1._g_wm is not None and r1._g_beh is not None  # graphs were captured
    assert all(np.isfinite(v1).ravel())
    assert len({v[0] for v in v1}) == len(v1), "the metrics of earlier updates were rewritten by later replays"
    for a, b in zip(v1, v0):
        assert abs(a[0] - b[0]) <= 2e-3 * abs(b[0]), (v1, v0)
        assert abs(a[2] - b[2]) <= 2e-2 * max(1.0, abs(b[2])), (v1, v0)
    err = float((w1 - w0).abs().max())
    assert err <= 1e-4 * max(1.0, float(w0.abs().max())), err
