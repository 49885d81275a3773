"""The autograd bridge (dv3hip/autograd.py): callers that build a loss on the PUBLIC class surface
and call `loss.backward()` / `tools.Optimizer.__call__` -- the reference's exploration.Plan2Explore and its causal world
models (scm_world_model.py:500-560, causal_VAE.py:1045-1120 re-state WorldModel._train in that style) -- get the same
numbers as the reference, with the HIP kernels running forward AND backward.

* test_public_surface_world_model_update: the reference's WorldModel._train written against the public methods
  (encoder -> observe -> kl_loss -> heads -> log_prob -> Optimizer(loss, params)), compared with the REFERENCE's own
  gradients and post-Adam parameters (tests/golden/tiny*.npz).
* the piece tests: each public method's values and input / parameter gradients against the CPU oracle's torch autograd.
* test_plan2explore_*: exploration.Plan2Explore.train against the reference's exploration.Plan2Explore.train
  (tests/golden/tiny_p2e*.npz, written by tests/golden/make_golden.py run_p2e) and against the oracle.
fp32; values 1e-4 (north star), gradients 3e-4 of the tensor's scale."""
import numpy as np
import pytest
import torch

from oracle import dv3_oracle as O
from tests import helpers as Hh
from tests.golden import common
from tests.test_path_gpu import adam_close, close, gpu_noise

pytestmark = pytest.mark.gpu
GTOL = 3e-4


def _gold(name):
    import os

    return np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"), allow_pickle=False)


def _cpu_params(name, grad=True):
    return {k: torch.from_numpy(v.copy()).requires_grad_(grad) for k, v in common.make_weights(name).items()}


# ---------------------------------------------------------------------------------------------
# the reference's WorldModel._train, autograd style, on the public surface
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "tiny_proprio"])
def test_public_surface_world_model_update(name):
    import tools

    g = _gold(name)
    cfg, wm, _ = Hh.build_models(name)
    wm_noise, _ = gpu_noise(name)
    data = common.make_batch(name)
    with tools.RequiresGrad(wm):
        obs = wm.preprocess(data)
        embed = wm.encoder(obs)
        post, prior = wm.dynamics.observe(embed, obs["action"], obs["is_first"], noise=wm_noise)
        kl_loss, kl_value, dyn_loss, rep_loss = wm.dynamics.kl_loss(post, prior, cfg.kl_free, cfg.dyn_scale,
                                                                    cfg.rep_scale)
        assert kl_loss.shape == embed.shape[:2]
        preds = {}
        for hname, head in wm.heads.items():
            feat = wm.dynamics.get_feat(post)
            pred = head(feat)
            preds.update(pred if isinstance(pred, dict) else {hname: pred})
        losses = {k: -pred.log_prob(obs[k]) for k, pred in preds.items()}
        for k, v in losses.items():
            assert v.shape == embed.shape[:2], (k, v.shape)
        model_loss = sum(losses.values()) + kl_loss
        mets = wm._model_opt(torch.mean(model_loss), wm.parameters())
    torch.cuda.synchronize()
    close(post["stoch"], torch.from_numpy(g["post/stoch"]), what="post stoch")
    close(post["logit"], torch.from_numpy(g["post/logit"]), what="post logit")
    close(prior["logit"], torch.from_numpy(g["prior/logit"]), what="prior logit")
    close(embed, torch.from_numpy(g["embed"]), what="embed")
    for k, v in losses.items():
        close(v, torch.from_numpy(g["loss/" + k]), what="loss/" + k)
    close(kl_value, torch.from_numpy(g["kl_value"]), what="kl value")
    close(torch.tensor(float(mets["model_loss"])), torch.from_numpy(g["model_loss"]), tol=1e-5, what="model_loss")
    close(torch.tensor(float(mets["model_grad_norm"])), torch.from_numpy(np.asarray(g["model_grad_norm"])), tol=2e-5,
          what="model_grad_norm")
    for k, p in wm.named_parameters():
        close(p.grad, torch.from_numpy(g["grad/" + k]), tol=GTOL, what="grad/" + k)
    for k, v in wm.state_dict().items():
        adam_close(v, torch.from_numpy(g["after/" + k]), 1e-4, "after/" + k)


# ---------------------------------------------------------------------------------------------
# pieces: values + gradients against the oracle's torch autograd on the CPU
# ---------------------------------------------------------------------------------------------
def _grads_close(got_named, ref_named, what):
    for k, r in ref_named.items():
        assert got_named[k] is not None, f"{what}: no gradient for {k}"
        close(got_named[k], r, tol=GTOL, what=f"{what} d/d{k}")


def test_rssm_components_as_the_causal_models_call_them():
    """scm_world_model.py:129-165: `_img_in_layers(x)`, `_cell(x, [deter])`, `_obs_out_layers`, `_img_out_layers`,
    `_suff_stats_layer(name, x)`, `get_dist(stats).sample()` called one by one, gradients by loss.backward()."""
    import tools

    name = "tiny"
    cfg = common.path_config(name)
    s = common.SHAPES[name]
    M, S, D, A = 5, s["stoch"], s["discrete"], s["A"]
    p = _cpu_params(name)
    _, wm, _ = Hh.build_models(name)
    dyn = wm.dynamics
    rs = np.random.RandomState(3)
    stoch_idx = rs.randint(0, D, (M, S))
    stoch = torch.from_numpy(np.eye(D, dtype=np.float32)[stoch_idx])
    deter = torch.from_numpy(rs.randn(M, s["deter"]).astype(np.float32))
    action = torch.from_numpy(rs.uniform(-1, 1, (M, A)).astype(np.float32))
    embed = torch.from_numpy(rs.randn(M, 2 * 8 * 16).astype(np.float32))
    q1 = torch.from_numpy(np.maximum(rs.exponential(size=(M, S, D)), 1e-20).astype(np.float32))
    w_out = torch.from_numpy(rs.randn(M, S, D).astype(np.float32))
    # oracle
    dc, ac, ec = (t.clone().requires_grad_(True) for t in (deter, action, embed))
    pri = O.img_step(cfg, p, stoch, dc, ac, q1)
    x = O.dense_ln_silu(torch.cat([pri["deter"], ec], -1), p["dynamics._obs_out_layers.0.weight"],
                        p["dynamics._obs_out_layers.1.weight"], p["dynamics._obs_out_layers.1.bias"])
    lg = (x @ p["dynamics._obs_stat_layer.weight"].t() + p["dynamics._obs_stat_layer.bias"]).reshape(M, S, D)
    post = O.onehot_sample(lg, q1, cfg.unimix)
    loss_ref = (post * w_out).sum() + (pri["stoch"] * w_out).sum() * 0.5 + pri["deter"].pow(2).sum()
    names = [k for k in p if k.startswith("dynamics.") and k != "dynamics.W"]
    ref = torch.autograd.grad(loss_ref, [dc, ac, ec] + [p[k] for k in names])
    # GPU, piece by piece
    dg, ag, eg = (t.cuda().requires_grad_(True) for t in (deter, action, embed))
    with tools.RequiresGrad(dyn):
        h = dyn._img_in_layers(torch.cat([stoch.cuda().reshape(M, S * D), ag], -1))
        _, dl = dyn._cell(h, [dg])
        d2 = dl[0]
        pst = dyn._suff_stats_layer("ims", dyn._img_out_layers(d2))
        pri_st = dyn.get_dist(pst).sample(noise=q1.cuda())
        ost = dyn._suff_stats_layer("obs", dyn._obs_out_layers(torch.cat([d2, eg], -1)))
        post_g = dyn.get_dist(ost).sample(noise=q1.cuda())
        loss = (post_g * w_out.cuda()).sum() + (pri_st * w_out.cuda()).sum() * 0.5 + d2.pow(2).sum()
        loss.backward()
    close(post_g, post, what="posterior sample")
    close(pst["logit"], pri["logit"], what="prior logit")
    close(loss, loss_ref, tol=1e-5, what="loss")
    _grads_close(dict(deter=dg.grad, action=ag.grad, embed=eg.grad), dict(deter=ref[0], action=ref[1], embed=ref[2]),
                 "rssm pieces")
    got = {"dynamics." + k: v.grad for k, v in dyn.named_parameters()}
    _grads_close(got, dict(zip(names, ref[3:])), "rssm pieces")


@pytest.mark.parametrize("resets", ["none", "some", "all"])
def test_obs_step_and_img_step_autograd(resets):
    name = "tiny"
    cfg = common.path_config(name)
    s = common.SHAPES[name]
    M, S, D, A = 6, s["stoch"], s["discrete"], s["A"]
    p = _cpu_params(name)
    _, wm, _ = Hh.build_models(name)
    dyn = wm.dynamics
    rs = np.random.RandomState(11)
    stoch = torch.from_numpy(np.eye(D, dtype=np.float32)[rs.randint(0, D, (M, S))])
    deter = torch.from_numpy(rs.randn(M, s["deter"]).astype(np.float32))
    logit = torch.from_numpy(rs.randn(M, S, D).astype(np.float32))
    action = torch.from_numpy(rs.uniform(-1, 1, (M, A)).astype(np.float32))
    embed = torch.from_numpy(rs.randn(M, 2 * 8 * 16).astype(np.float32))
    first = {"none": np.zeros(M), "some": (np.arange(M) % 3 == 0), "all": np.ones(M)}[resets].astype(np.float32)
    first = torch.from_numpy(first)
    qa = torch.from_numpy(np.maximum(rs.exponential(size=(M, S, D)), 1e-20).astype(np.float32))
    qb = torch.from_numpy(np.maximum(rs.exponential(size=(M, S, D)), 1e-20).astype(np.float32))
    w = torch.from_numpy(rs.randn(M, S, D).astype(np.float32))
    dc, ec = deter.clone().requires_grad_(True), embed.clone().requires_grad_(True)
    post, prior = O.obs_step(cfg, p, {"stoch": stoch, "deter": dc, "logit": logit}, action, ec, first, qa, qb)
    loss_ref = (post["stoch"] * w).sum() + post["deter"].sum() + (prior["logit"] * w).sum()
    names = [k for k in p if k.startswith("dynamics.")]
    ref = torch.autograd.grad(loss_ref, [dc, ec] + [p[k] for k in names])
    import tools

    dg, eg = deter.cuda().requires_grad_(True), embed.cuda().requires_grad_(True)
    with tools.RequiresGrad(dyn):
        gpost, gprior = dyn.obs_step({"stoch": stoch.cuda(), "deter": dg, "logit": logit.cuda()}, action.cuda(), eg,
                                     first.cuda(), noise=dict(prior=qa.cuda(), post=qb.cuda()))
        loss = (gpost["stoch"] * w.cuda()).sum() + gpost["deter"].sum() + (gprior["logit"] * w.cuda()).sum()
        loss.backward()
    close(gpost["stoch"], post["stoch"], what="post stoch")
    close(gpost["logit"], post["logit"], what="post logit")
    close(gprior["stoch"], prior["stoch"], what="prior stoch")
    _grads_close(dict(deter=dg.grad, embed=eg.grad), dict(deter=ref[0], embed=ref[1]), "obs_step " + resets)
    got = {"dynamics." + k: v.grad for k, v in dyn.named_parameters()}
    _grads_close(got, dict(zip(names, ref[2:])), "obs_step " + resets)


@pytest.mark.parametrize("name", ["tiny", "tiny_onehot", "tiny_proprio"])
def test_head_distributions_autograd(name):
    """Every head the shipped configs build: actor (normal: sample / entropy / log_prob; onehot: sample / entropy /
    log_prob), reward + value (symlog_disc: mode / log_prob), cont (binary), the vector decoder (symlog_mse)."""
    import tools

    cfg = common.path_config(name)
    s = common.SHAPES[name]
    p = _cpu_params(name)
    _, wm, beh = Hh.build_models(name)
    rs = np.random.RandomState(5)
    R = 7
    F_ = s["stoch"] * s["discrete"] + s["deter"]
    feat = torch.from_numpy(rs.randn(2, R, F_).astype(np.float32))
    A = s["A"]
    # ---- actor
    fc = feat.clone().requires_grad_(True)
    akeys = [k for k in p if k.startswith("actor.")]
    fg = feat.cuda().requires_grad_(True)
    if s["actor_dist"] == "normal":
        eps = torch.from_numpy(rs.randn(2, R, A).astype(np.float32))
        given = torch.from_numpy(rs.uniform(-1, 1, (2, R, A)).astype(np.float32))
        a_ref = O.actor_sample(cfg, p, fc, eps)
        ref_loss = (a_ref ** 2).sum() + O.actor_entropy(cfg, p, fc).sum() + O.actor_logprob(cfg, p, fc, given).sum()
        with tools.RequiresGrad(beh.actor):
            dist = beh.actor(fg)
            a = dist.sample(noise=eps.cuda())
            loss = (a ** 2).sum() + dist.entropy().sum() + dist.log_prob(given.cuda()).sum()
            loss.backward()
    else:
        q = torch.from_numpy(np.maximum(rs.exponential(size=(2, R, A)), 1e-20).astype(np.float32))
        wgt = torch.from_numpy(rs.randn(2, R, A).astype(np.float32))
        given = torch.from_numpy(np.eye(A, dtype=np.float32)[rs.randint(0, A, (2, R))])
        a_ref = O.actor_sample(cfg, p, fc, q)
        ref_loss = (a_ref * wgt).sum() + O.actor_entropy(cfg, p, fc).sum() + O.actor_logprob(cfg, p, fc, given).sum()
        with tools.RequiresGrad(beh.actor):
            dist = beh.actor(fg)
            a = dist.sample(noise=q.cuda())
            loss = (a * wgt.cuda()).sum() + dist.entropy().sum() + dist.log_prob(given.cuda()).sum()
            loss.backward()
    ref = torch.autograd.grad(ref_loss, [fc] + [p[k] for k in akeys])
    close(a, a_ref, what="actor sample")
    close(loss, ref_loss, tol=1e-5, what="actor loss")
    close(fg.grad, ref[0], tol=GTOL, what="actor d/dfeat")
    _grads_close({"actor." + k: v.grad for k, v in beh.actor.named_parameters()}, dict(zip(akeys, ref[1:])), "actor")
    # ---- reward / value (symlog_disc) and cont (binary)
    x = torch.from_numpy(rs.randn(2, R).astype(np.float32) * 3)
    c = torch.from_numpy((rs.rand(2, R, 1) > 0.5).astype(np.float32))
    fc = feat.clone().requires_grad_(True)
    rl = O.head_logits(p, "heads.reward.", "Reward", cfg.reward_layers, fc)
    cl = O.head_logits(p, "heads.cont.", "Cont", cfg.cont_layers, fc)
    ref_loss = O.disc_mode(rl).sum() + O.disc_logprob(rl, x).sum() + O.bernoulli_logprob(cl, c).sum()
    hk = [k for k in p if k.startswith(("heads.reward.", "heads.cont."))]
    ref = torch.autograd.grad(ref_loss, [fc] + [p[k] for k in hk])
    fg = feat.cuda().requires_grad_(True)
    with tools.RequiresGrad(wm.heads):
        rd, cd = wm.heads["reward"](fg), wm.heads["cont"](fg)
        loss = rd.mode().sum() + rd.log_prob(x.cuda()).sum() + cd.log_prob(c.cuda()).sum()
        loss.backward()
    close(loss, ref_loss, tol=1e-5, what="head loss")
    close(fg.grad, ref[0], tol=GTOL, what="heads d/dfeat")
    got = {"heads." + k: v.grad for k, v in wm.heads.named_parameters()}
    _grads_close(got, dict(zip(hk, ref[1:])), "heads")
    # ---- vector decoder (symlog_mse)
    if s["encoder"] == "mlp":
        tgt = {k: torch.from_numpy(rs.randn(2, R, w).astype(np.float32) * 2) for k, w in common.PROPRIO_KEYS}
        fc = feat.clone().requires_grad_(True)
        modes = O.mlp_decoder_modes(cfg, p, fc)
        ref_loss = sum(O.symlog_mse_logprob(modes[k], tgt[k]).sum() for k in modes)
        dk = [k for k in p if k.startswith("heads.decoder.")]
        ref = torch.autograd.grad(ref_loss, [fc] + [p[k] for k in dk])
        fg = feat.cuda().requires_grad_(True)
        with tools.RequiresGrad(wm.heads["decoder"]):
            dists = wm.heads["decoder"](fg)
            loss = sum(dists[k].log_prob(tgt[k].cuda()).sum() for k in dists)
            loss.backward()
        close(loss, ref_loss, tol=1e-5, what="decoder loss")
        close(fg.grad, ref[0], tol=GTOL, what="decoder d/dfeat")
        got = {"heads.decoder." + k: v.grad for k, v in wm.heads["decoder"].named_parameters()}
        _grads_close(got, dict(zip(dk, ref[1:])), "decoder")


def test_imagine_with_a_foreign_policy():
    """ImagBehavior._imagine(start, policy, horizon) with a policy that is NOT the behaviour's own actor (models.py:448
    takes any callable): step-by-step rollout through the public methods, differentiable w.r.t. the policy."""
    import networks
    import tools

    name = "tiny"
    s = common.SHAPES[name]
    cfgo = common.path_config(name)
    cfg, wm, beh = Hh.build_models(name)
    p = _cpu_params(name)
    B, T, H, S, D, A = s["B"], s["T"], s["H"], s["stoch"], s["discrete"], s["A"]
    out = O.wm_forward(cfgo, p, common.make_batch(name), *[torch.from_numpy(common.make_noise(name)[k])
                                                             for k in ("q_prior", "q_post")])
    start = {k: v.detach() for k, v in out["post"].items()}
    n = common.make_noise(name, seed=9)
    # the foreign policy: a fresh actor-shaped MLP with its own weights (the oracle reads them under "actor.")
    torch.manual_seed(4)
    pol = networks.MLP(S * D + s["deter"], (A,), 2, s["units"], dist="normal", std="learned", absmax=1.0,
                       name="Actor").cuda()
    pw = {"actor." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in pol.state_dict().items()}
    q = {**{k: v for k, v in p.items() if not k.startswith("actor.")}, **pw}
    feats_r, states_r, actions_r = O.imagine(cfgo, q, start, torch.from_numpy(n["act"]), torch.from_numpy(n["q_img"]))
    loss_ref = (states_r["deter"] ** 2).sum() + (actions_r ** 2).sum()
    ref = torch.autograd.grad(loss_ref, list(pw.values()))
    # GPU: rows are b*T+t as in the reference (generic path); feed the same noise through a scripted policy wrapper
    step = {"t": 0}

    class Scripted:
        def __init__(self, dist, t):
            self.dist, self.t = dist, t

        def sample(self):
            return self.dist.sample(noise=torch.from_numpy(n["act"][self.t]).cuda())

    def policy(feat):
        d = Scripted(pol(feat), step["t"])
        step["t"] += 1
        return d

    qi = iter(range(H))
    stock = wm.dynamics.img_step
    wm.dynamics.img_step = lambda st, a: stock(st, a, noise=torch.from_numpy(n["q_img"][next(qi)]).cuda())
    with tools.RequiresGrad(pol):
        feats, states, actions = beh._imagine({k: v.cuda() for k, v in start.items()}, policy, H)
        loss = (states["deter"] ** 2).sum() + (actions ** 2).sum()
        loss.backward()
    del wm.dynamics.img_step
    close(states["stoch"], states_r["stoch"], what="imag stoch")
    close(states["deter"], states_r["deter"], what="imag deter")
    close(actions, actions_r, what="imag action")
    close(feats, feats_r, what="imag feat")
    got = {"actor." + k: v.grad for k, v in pol.named_parameters()}
    _grads_close(got, dict(zip(pw.keys(), ref)), "foreign policy")


# ---------------------------------------------------------------------------------------------
# Plan2Explore
# ---------------------------------------------------------------------------------------------
def _build_p2e(name, wm, cfg):
    import exploration

    extr = lambda f, st, a: wm.heads["reward"](f).mean()  # dreamer.py:80
    p2e = exploration.Plan2Explore(cfg, wm, extr).cuda()
    pw = common.make_p2e_weights(name)
    sd = p2e.state_dict()
    for k, v in pw.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v)
        if k.startswith("_behavior.actor."):
            sd[k[len("_behavior."):]] = sd[k]  # exploration.py:47: the same module under a second name
    p2e.load_state_dict(sd)
    p2e.requires_grad_(False)
    return p2e


@pytest.fixture(scope="module", params=["tiny_p2e", "tiny_p2e_ac"])
def p2e_run(request):
    name = request.param
    cfg, wm, _ = Hh.build_models(name)
    p2e = _build_p2e(name, wm, cfg)
    wm_noise, _ = gpu_noise(name)
    _, x_noise = gpu_noise(name, seed=5)
    data = common.make_batch(name)
    post, context, _ = wm._train(data, noise=wm_noise)
    stock = p2e._behavior._train
    p2e._behavior._train = lambda st, obj: stock(st, obj, noise=x_noise)
    _, mets = p2e.train(post, context, data)
    torch.cuda.synchronize()
    grads = {k: v.grad.clone() for k, v in p2e.named_parameters()
             if v.grad is not None and not k.startswith(("_behavior._world_model", "actor."))}
    return dict(name=name, p2e=p2e, mets=mets, grads=grads, g=_gold(name), exp=Hh.oracle_p2e_update(name))


def test_plan2explore_ensemble_update(p2e_run):
    g, exp, mets, p2e = p2e_run["g"], p2e_run["exp"], p2e_run["mets"], p2e_run["p2e"]
    for ref, src in ((torch.from_numpy(np.asarray(g["train/explorer_loss"])), "reference"), (exp["explorer_loss"], "oracle")):
        close(torch.tensor(float(mets["explorer_loss"])), ref, tol=1e-5, what=f"explorer_loss vs {src}")
    close(torch.tensor(float(mets["explorer_grad_norm"])), torch.from_numpy(np.asarray(g["train/explorer_grad_norm"])),
          tol=GTOL, what="explorer_grad_norm")
    n = 0
    for k, gr in p2e_run["grads"].items():
        if k.startswith("_networks."):
            close(gr, torch.from_numpy(g["grad/" + k]), tol=GTOL, what="reference grad/" + k)
            close(gr, exp["explorer_grads"][k], tol=GTOL, what="oracle grad/" + k)
            n += 1
    assert n == len(exp["explorer_grads"])
    sd = p2e.state_dict()
    for k in exp["explorer_grads"]:
        adam_close(sd[k], torch.from_numpy(g["after/" + k]), 1e-4, "after/" + k)


def test_plan2explore_behaviour_update(p2e_run):
    name, g, exp, mets, p2e = p2e_run["name"], p2e_run["g"], p2e_run["exp"], p2e_run["mets"], p2e_run["p2e"]
    s = common.SHAPES[name]
    B, T = s["B"], s["T"]
    beh = p2e._behavior
    unperm = lambda x: Hh.from_time_major_rows(x, B, T)
    close(unperm(beh._last["reward"]), torch.from_numpy(g["imag/reward"]).squeeze(-1), what="intrinsic reward (reference)")
    close(unperm(beh._last["reward"]), exp["beh"]["reward"].squeeze(-1), what="intrinsic reward (oracle)")
    close(unperm(beh._last["target"]), exp["beh"]["target"].squeeze(-1), what="lambda-return")
    for k in ("actor_loss", "value_loss", "EMA_005", "EMA_095", "actor_entropy", "imag_reward_mean", "target_mean"):
        close(torch.tensor(float(mets[k])), torch.from_numpy(np.asarray(g["train/" + k])), tol=2e-5, what=k)
    close(torch.tensor(float(mets["actor_grad_norm"])), torch.from_numpy(np.asarray(g["train/actor_grad_norm"])),
          tol=GTOL, what="actor_grad_norm")
    close(torch.tensor(float(mets["value_grad_norm"])), torch.from_numpy(np.asarray(g["train/value_grad_norm"])),
          tol=GTOL, what="value_grad_norm")
    n = 0
    for k, gr in p2e_run["grads"].items():
        if k.startswith(("_behavior.actor.", "_behavior.value.")):
            close(gr, torch.from_numpy(g["grad/" + k]), tol=GTOL, what="reference grad/" + k)
            n += 1
    assert n == len(exp["actor_grads"]) + len(exp["value_grads"])
    sd = p2e.state_dict()
    for k in sd:
        if k.startswith(("_behavior.actor.", "_behavior.value.")):
            adam_close(sd[k], torch.from_numpy(g["after/" + k]), 3e-5, "after/" + k)
        elif k.startswith("_behavior._slow_value."):
            close(sd[k], torch.from_numpy(g["after/" + k]), tol=1e-6, what="after/" + k)
    # the objective was NOT mistaken for the reward head, and the task behaviour's fused path is untouched by it
    assert beh._objective_kinds and not any(beh._objective_kinds.values())


def test_reward_head_objective_stays_on_the_fused_path():
    """dreamer.py:196-199 passes a lambda around the reward head: recognised once (by value), remembered by code
    object, and trained through the fused kernels -- same result as objective=None."""
    name = "tiny"
    res = []
    for use_lambda in (False, True):
        cfg, wm, beh = Hh.build_models(name)
        wm_noise, im_noise = gpu_noise(name)
        post, _, _ = wm._train(common.make_batch(name), noise=wm_noise)
        post = {k: v.clone() for k, v in post.items()}
        obj = (lambda f, st, a: wm.heads["reward"](wm.dynamics.get_feat(st)).mode()) if use_lambda else None
        mets = beh._train(post, obj, noise=im_noise)[-1]
        if use_lambda:
            assert list(beh._objective_kinds.values()) == [True]
        res.append((float(mets["actor_loss"]), float(mets["value_loss"]),
                    {k: v.grad.clone() for k, v in beh.actor.named_parameters()}))
    # (not bit-equal: the reverse observe scan adds its split-K partial tiles atomically, so the two world models
    # differ in the last bits after their Adam step)
    assert abs(res[0][0] - res[1][0]) <= 1e-6 and abs(res[0][1] - res[1][1]) <= 1e-6
    for k in res[0][2]:
        close(res[1][2][k], res[0][2][k], tol=1e-5, what=k)


def test_dreamer_trains_with_plan2explore():
    """dreamer.Dreamer with expl_behavior 'plan2explore': _train runs the world model, the task behaviour (hipGraph
    replay) and the explorer; the exploration actor acts while training, the task actor at evaluation."""
    import dreamer

    name = "tiny_p2e"
    cfg = Hh.make_config(name)
    cfg.log_every, cfg.train_ratio, cfg.reset_every, cfg.expl_until, cfg.action_repeat = 1e9, 1, 0, 0, 1
    cfg.pretrain, cfg.video_pred_log = 1, False
    agent = dreamer.Dreamer(Hh.obs_space(name), None, cfg, None, None).cuda()
    agent.requires_grad_(False)
    losses = []
    for i in range(4):
        agent._train(common.make_batch(name, seed=i))
    agent._flush_metrics()
    m = agent._metrics
    for k in ("model_loss", "actor_loss", "expl_explorer_loss", "expl_actor_loss", "expl_value_loss", "expl_imag_reward_mean"):
        assert k in m and np.isfinite(np.mean(m[k])), (k, m.get(k))
    obs = {k: v[:, 0] for k, v in common.make_batch(name).items()}
    obs = {k: obs[k] for k in ("image", "is_first", "is_terminal")}
    out_t, st = agent._policy(obs, None, training=True)
    out_e, _ = agent._policy(obs, None, training=False)
    assert out_t["action"].shape == out_e["action"].shape == (common.SHAPES[name]["B"], common.SHAPES[name]["A"])
    assert torch.isfinite(out_t["logprob"]).all() and torch.isfinite(out_e["logprob"]).all()
    assert agent._exploring()
