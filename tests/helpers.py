"""Shared test scaffolding: build the MI355X-native models for a named shape config, load the
deterministic weights, and compute the oracle's expectation of one full update on the CPU."""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from oracle import dv3_oracle as O
from tests.golden import common

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dreamerv3-torch_amd")
WM_PREFIXES = ("encoder", "dynamics", "heads")


from dv3hip.shapes import make_config, obs_space  # noqa: E402,F401  (moved into the package: bench.py needs no tests/)


def build_models(name, device="cuda:0", weights=None):
    """-> (config, WorldModel, ImagBehavior) on the GPU with common.make_weights(name) loaded."""
    import models

    cfg = make_config(name, device)
    wm = models.WorldModel(obs_space(name), None, 0, cfg).to(device)
    beh = models.ImagBehavior(cfg, wm).to(device)
    w = weights if weights is not None else common.make_weights(name)
    sd = {k: torch.from_numpy(v) for k, v in w.items() if k.split(".")[0] in WM_PREFIXES}
    missing = set(wm.state_dict().keys()) ^ set(sd.keys())
    assert not missing, missing
    wm.load_state_dict(sd)
    bsd = beh.state_dict()
    for k in list(bsd):
        if k.startswith("_world_model.") or k == "ema_vals":
            continue
        bsd[k] = torch.from_numpy(w[k])
    beh.load_state_dict(bsd)
    wm.requires_grad_(False)
    beh.requires_grad_(False)
    return cfg, wm, beh


def to_time_major_rows(x, B, T):
    """[..., N=b*T+t, ...] on dim 1 -> rows ordered t*B+b (how the GPU path lays out imagination rows)."""
    sh = x.shape
    return x.reshape((sh[0], B, T) + tuple(sh[2:])).transpose(1, 2).reshape(sh)


def from_time_major_rows(x, B, T):
    sh = x.shape
    return x.reshape((sh[0], T, B) + tuple(sh[2:])).transpose(1, 2).reshape(sh)


def oracle_update(name, threads=None, piecewise=False):
    """One full update in the oracle (WM step, slow-critic EMA, behaviour on the UPDATED world model,
    as dreamer.py:194-200).  Returns everything the GPU tests compare against.  piecewise=True also
    evaluates the behaviour losses and gradients on the world model BEFORE its Adam step and without the
    slow-critic EMA update (res["beh0"], ...): that is the setting of the golden files' imag/* and
    grad/actor.*, grad/value.* entries (tests/golden/make_golden.py runs the pieces before the _train calls)."""
    if threads:
        torch.set_num_threads(threads)
    cfg = common.path_config(name)
    p = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in common.make_weights(name).items()}
    n = {k: torch.from_numpy(v) for k, v in common.make_noise(name).items()}
    data = common.make_batch(name)
    out = O.wm_forward(cfg, p, data, n["q_prior"], n["q_post"])
    wkeys = [k for k in p if k.split(".")[0] in WM_PREFIXES]
    grads = torch.autograd.grad(out["model_loss"], [p[k] for k in wkeys])
    res = dict(cfg=cfg, wm=out, wm_grads={k: g.clone() for k, g in zip(wkeys, grads)})
    start = {k: v.detach() for k, v in out["post"].items()}
    akeys = [k for k in p if k.startswith("actor.")]
    vkeys = [k for k in p if k.startswith("value.")]
    if piecewise:
        ema0 = torch.zeros(2)
        b0 = O.behavior_forward(cfg, p, start, n["act"], n["q_img"], ema0)
        ga0 = torch.autograd.grad(b0["actor_loss"], [p[k] for k in akeys], retain_graph=True)
        gv0 = torch.autograd.grad(b0["value_loss"], [p[k] for k in vkeys])
        res.update(beh0={k: (v.detach() if isinstance(v, torch.Tensor) else {kk: vv.detach() for kk, vv in v.items()})
                         for k, v in b0.items()},
                   ema0=ema0, actor_grads0=dict(zip(akeys, ga0)), value_grads0=dict(zip(vkeys, gv0)))
        del b0
    with torch.no_grad():
        st = dict(step=0, m=[torch.zeros_like(p[k]) for k in wkeys], v=[torch.zeros_like(p[k]) for k in wkeys])
        res["model_grad_norm"] = O.clip_and_adam([p[k] for k in wkeys], list(grads), st, lr=1e-4, eps=1e-8, clip=1000.0)
        for k in list(p):
            if k.startswith("value."):
                sk = "_slow_value." + k[len("value."):]
                p[sk].copy_(cfg.slow_target_fraction * p[k] + (1 - cfg.slow_target_fraction) * p[sk])
    # the state the behaviour update runs on: world model after its Adam step, slow critic after its EMA step
    res["params_mid"] = {k: v.detach().clone() for k, v in p.items()
                         if k.split(".")[0] in WM_PREFIXES or k.startswith("_slow_value.")}
    ema = torch.zeros(2)
    bout = O.behavior_forward(cfg, p, start, n["act"], n["q_img"], ema)
    ga = torch.autograd.grad(bout["actor_loss"], [p[k] for k in akeys], retain_graph=True)
    gv = torch.autograd.grad(bout["value_loss"], [p[k] for k in vkeys])
    res.update(beh=bout, ema=ema, actor_grads=dict(zip(akeys, ga)), value_grads=dict(zip(vkeys, gv)))
    with torch.no_grad():
        for keys, gr, nm in ((akeys, ga, "actor"), (vkeys, gv, "value")):
            st = dict(step=0, m=[torch.zeros_like(p[k]) for k in keys], v=[torch.zeros_like(p[k]) for k in keys])
            res[nm + "_grad_norm"] = O.clip_and_adam([p[k] for k in keys], list(gr), st, lr=3e-5, eps=1e-5, clip=100.0)
    res["params_after"] = {k: v.detach() for k, v in p.items()}
    res["data"], res["noise"] = data, n
    return res


def oracle_updates(name, n_updates):
    """n consecutive full updates in the oracle (fresh batch and noise per update: seeds 0, 1, ...), carrying the three
    Adam states, the slow critic and the return-normalisation EMA across them as dreamer.py:192-208 does.
    -> dict(losses=[(model_loss, actor_loss, value_loss)], params_after, ema)."""
    cfg = common.path_config(name)
    p = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in common.make_weights(name).items()}
    wkeys = [k for k in p if k.split(".")[0] in WM_PREFIXES]
    akeys = [k for k in p if k.startswith("actor.")]
    vkeys = [k for k in p if k.startswith("value.")]
    mk = lambda keys: dict(step=0, m=[torch.zeros_like(p[k]) for k in keys], v=[torch.zeros_like(p[k]) for k in keys])
    st_w, st_a, st_v = mk(wkeys), mk(akeys), mk(vkeys)
    ema = torch.zeros(2)
    losses = []
    for i in range(n_updates):
        n = {k: torch.from_numpy(v) for k, v in common.make_noise(name, seed=i).items()}
        data = common.make_batch(name, seed=i)
        out = O.wm_forward(cfg, p, data, n["q_prior"], n["q_post"])
        grads = torch.autograd.grad(out["model_loss"], [p[k] for k in wkeys])
        start = {k: v.detach() for k, v in out["post"].items()}
        with torch.no_grad():
            O.clip_and_adam([p[k] for k in wkeys], list(grads), st_w, lr=1e-4, eps=1e-8, clip=1000.0)
            for k in list(p):
                if k.startswith("value."):
                    sk = "_slow_value." + k[len("value."):]
                    p[sk].copy_(cfg.slow_target_fraction * p[k] + (1 - cfg.slow_target_fraction) * p[sk])
        bout = O.behavior_forward(cfg, p, start, n["act"], n["q_img"], ema)
        ga = torch.autograd.grad(bout["actor_loss"], [p[k] for k in akeys], retain_graph=True)
        gv = torch.autograd.grad(bout["value_loss"], [p[k] for k in vkeys])
        with torch.no_grad():
            O.clip_and_adam([p[k] for k in akeys], list(ga), st_a, lr=3e-5, eps=1e-5, clip=100.0)
            O.clip_and_adam([p[k] for k in vkeys], list(gv), st_v, lr=3e-5, eps=1e-5, clip=100.0)
        losses.append(tuple(float(x.detach()) for x in (out["model_loss"], bout["actor_loss"], bout["value_loss"])))
    return dict(losses=losses, params_after={k: v.detach() for k, v in p.items()}, ema=ema.clone())


def onehot_index(x):
    """one-hot [..., D] -> int32 class index [...] (contiguous, for teacher forcing)."""
    return x.detach().argmax(-1).to(torch.int32).contiguous()


def forced_draws(name, exp, beh_key="beh", device="cuda"):
    """Teacher-forcing inputs of the GPU path from an oracle run: the classes the oracle drew, laid out as the GPU
    path indexes them (observe: time-major [T,B,S]; imagination: rows t*B+b, entry t = the draw that produces
    state t+1).  -> (wm_force, im_force) dicts to merge into the `noise` arguments."""
    s = common.SHAPES[name]
    B, T, H = s["B"], s["T"], s["H"]
    w = exp["wm"]
    wm_force = dict(force_post=onehot_index(w["post"]["stoch"]).transpose(0, 1).contiguous().to(device),
                    force_prior=onehot_index(w["prior"]["stoch"]).transpose(0, 1).contiguous().to(device))
    b = exp[beh_key]
    st = onehot_index(b["states"]["stoch"])  # [H,N,S], state t; rows b*T+t
    st = to_time_major_rows(st, B, T)
    f_img = torch.zeros_like(st)
    f_img[:-1] = st[1:]  # the draw of step t yields state t+1; the discarded H-th successor is never drawn
    im_force = dict(force_img=f_img.contiguous().to(device))
    if s["actor_dist"] == "onehot":
        im_force["force_act"] = to_time_major_rows(onehot_index(b["actions"]), B, T).contiguous().to(device)
    return wm_force, im_force


def oracle_p2e_update(name):
    """dreamer.py:194-203 in the oracle for expl_behavior 'plan2explore': the world model's update, then one
    exploration update (ensemble regression + Adam, then the exploration behaviour on the intrinsic reward).
    -> dict with the quantities tests/golden/<name>.npz holds from the reference's exploration.Plan2Explore.train."""
    s = common.SHAPES[name]
    cfg = common.path_config(name)
    c = O.P2EConfig(**s["p2e"])
    p = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in common.make_weights(name).items()}
    pp = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in common.make_p2e_weights(name).items()}
    n = {k: torch.from_numpy(v) for k, v in common.make_noise(name).items()}
    nx = {k: torch.from_numpy(v) for k, v in common.make_noise(name, seed=5).items()}
    data = common.make_batch(name)
    out = O.wm_forward(cfg, p, data, n["q_prior"], n["q_post"])
    wkeys = [k for k in p if k.split(".")[0] in WM_PREFIXES]
    grads = torch.autograd.grad(out["model_loss"], [p[k] for k in wkeys])
    start = {k: v.detach() for k, v in out["post"].items()}
    feat = O.get_feat(cfg, start)
    embed = out["embed"].detach()
    with torch.no_grad():
        st = dict(step=0, m=[torch.zeros_like(p[k]) for k in wkeys], v=[torch.zeros_like(p[k]) for k in wkeys])
        O.clip_and_adam([p[k] for k in wkeys], list(grads), st, lr=1e-4, eps=1e-8, clip=1000.0)
    # ---- ensemble regression (exploration.py:88-103)
    B, T = s["B"], s["T"]
    target = {"stoch": start["stoch"].reshape(B, T, -1), "deter": start["deter"], "embed": embed}[c.disag_target]
    inputs = feat
    if c.disag_action_cond:
        inputs = torch.cat([inputs, torch.from_numpy(data["action"])], -1)
    ekeys = [k for k in pp if k.startswith("_networks.")]
    eloss = O.p2e_ensemble_loss(c, pp, inputs, target)
    eg = torch.autograd.grad(eloss, [pp[k] for k in ekeys])
    res = dict(explorer_loss=eloss.detach(), explorer_grads=dict(zip(ekeys, eg)), post=start)
    with torch.no_grad():
        st = dict(step=0, m=[torch.zeros_like(pp[k]) for k in ekeys], v=[torch.zeros_like(pp[k]) for k in ekeys])
        res["explorer_grad_norm"] = O.clip_and_adam([pp[k] for k in ekeys], list(eg), st, lr=1e-4, eps=1e-8, clip=1000.0)
    # ---- the exploration behaviour (its own actor / critic) on the intrinsic reward
    q = dict(p)
    for k, v in pp.items():
        if k.startswith("_behavior."):
            q[k[len("_behavior."):]] = v
    with torch.no_grad():
        for k in list(q):
            if k.startswith("value."):
                sk = "_slow_value." + k[len("value."):]
                q[sk].copy_(cfg.slow_target_fraction * q[k] + (1 - cfg.slow_target_fraction) * q[sk])

    def reward_fn(sfeat, states, actions):
        extr = None
        if c.expl_extr_scale:
            extr = O.disc_mode(O.head_logits(q, "heads.reward.", "Reward", cfg.reward_layers, sfeat))
        return O.p2e_intrinsic_reward(c, pp, sfeat, actions, extr)

    ema = torch.zeros(2)
    b = O.behavior_forward(cfg, q, start, nx["act"], nx["q_img"], ema, reward_fn=reward_fn)
    akeys = [k for k in q if k.startswith("actor.")]
    vkeys = [k for k in q if k.startswith("value.")]
    ga = torch.autograd.grad(b["actor_loss"], [q[k] for k in akeys], retain_graph=True)
    gv = torch.autograd.grad(b["value_loss"], [q[k] for k in vkeys])
    res.update(beh=b, ema=ema, actor_grads=dict(zip(akeys, ga)), value_grads=dict(zip(vkeys, gv)))
    with torch.no_grad():
        for keys, gr, nm in ((akeys, ga, "actor"), (vkeys, gv, "value")):
            st = dict(step=0, m=[torch.zeros_like(q[k]) for k in keys], v=[torch.zeros_like(q[k]) for k in keys])
            res[nm + "_grad_norm"] = O.clip_and_adam([q[k] for k in keys], list(gr), st, lr=3e-5, eps=1e-5, clip=100.0)
    res["p2e_after"] = {k: v.detach() for k, v in pp.items()}
    res["wm_after"] = {k: p[k].detach() for k in wkeys}
    res["data"], res["noise"], res["noise_x"], res["cfg"], res["p2e_cfg"] = data, n, nx, cfg, c
    return res
