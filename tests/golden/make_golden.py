#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Imports `tools`, `networks`, `models` unmodified from /root/reference (with
its three import accommodations: a stub `torch.utils.tensorboard`, PyYAML + float coercion
instead of ruamel, and `device="cpu"` as the `networks.MLP` default).  The reference never
travels to the GPU box; only the .npz files written here do.  What is stored is data only:
inputs, injected noise (or the seed that regenerates it) and the reference's outputs.

Noise injection: `tools.OneHotDist.sample` draws through torch.multinomial, whose single-draw
path is argmax(probs / q), q ~ Exp(1).  We replace the draw of q by popping the next array of
a numpy-generated tape (and do the same for the actor's N(0,1) draw), after first checking,
under a fixed torch seed, that the replacement with torch's own q is bit-identical to the
stock sampler.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from tests.golden import common  # noqa: E402


# ---------------------------------------------------------------------------------------
# reference import (accommodations 1-3 of SURVEY.md §8c)
# ---------------------------------------------------------------------------------------
def import_reference():
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:  # never used on this path
        def __init__(self, *a, **k):
            pass

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    sys.path.insert(0, REF)
    import tools  # noqa
    import networks  # noqa
    import models  # noqa

    # gotcha 5: networks.MLP defaults device="cuda"; there is no GPU in this container
    d = list(networks.MLP.__init__.__defaults__)
    d[d.index("cuda")] = "cpu"
    networks.MLP.__init__.__defaults__ = tuple(d)
    return tools, networks, models


def load_config(blocks, overrides):
    import yaml

    with open(os.path.join(REF, "configs.yaml")) as f:
        raw = yaml.safe_load(f)

    def coerce(x):
        if isinstance(x, dict):
            return {k: coerce(v) for k, v in x.items()}
        if isinstance(x, str) and re.fullmatch(r"-?\d+(\.\d*)?[eE][-+]?\d+", x):
            return float(x)
        return x

    def merge(base, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(base.get(k), dict):
                merge(base[k], v)
            else:
                base[k] = v

    cfg = {}
    for name in ["defaults"] + list(blocks):
        merge(cfg, coerce(raw[name]))
    for k, v in overrides.items():
        if isinstance(v, dict):
            merge(cfg.setdefault(k, {}), v)
        else:
            cfg[k] = v
    return argparse.Namespace(**cfg)


class Space:
    def __init__(self, shape):
        self.shape = shape


class ObsSpace:
    def __init__(self, spaces):
        self.spaces = spaces


class Tape:
    def __init__(self, arrays):
        self.arrays = list(arrays)
        self.pos = 0

    def next(self, shape):
        a = self.arrays[self.pos]
        self.pos += 1
        assert tuple(a.shape) == tuple(shape), (a.shape, tuple(shape), self.pos)
        return torch.from_numpy(np.ascontiguousarray(a))


TAPE = None  # current noise tape (None -> draw from torch RNG)


def install_noise_hooks(tools):
    import torch.distributions as torchd

    def sample(self, sample_shape=(), seed=None):
        assert sample_shape == () and seed is None
        probs = self._categorical.probs  # == super().probs in tools.py:456
        d = probs.shape[-1]
        if TAPE is None:
            q = torch.empty(probs.reshape(-1, d).shape).exponential_(1).reshape(probs.shape)
        else:
            q = TAPE.next(probs.shape)
        idx = torch.argmax(probs.detach() / q, -1)
        out = torch.nn.functional.one_hot(idx, d).to(probs.dtype)
        return out + (probs - probs.detach())

    stock = tools.OneHotDist.sample
    # bit-exactness of the replacement against the stock sampler, same torch seed
    logits = torch.randn(7, 5, 9)
    torch.manual_seed(123)
    a = stock(tools.OneHotDist(logits, unimix_ratio=0.01))
    torch.manual_seed(123)
    b = sample(tools.OneHotDist(logits, unimix_ratio=0.01))
    assert torch.equal(a, b), "argmax(p/q) replacement is not bit-identical to torch.multinomial"
    print("[golden] OneHotDist.sample == argmax(probs/Exp(1)) bit-exact under torch seed: OK")
    tools.OneHotDist.sample = sample

    stock_normal = torchd.normal._standard_normal

    def std_normal(shape, dtype, device):
        if TAPE is None:
            return stock_normal(shape, dtype, device)
        return TAPE.next(shape).to(dtype)

    torchd.normal._standard_normal = std_normal


# ---------------------------------------------------------------------------------------
def build_reference(name, tools, networks, models):
    s = common.SHAPES[name]
    blocks = ["dmc_proprio"] if s["encoder"] == "mlp" else ["dmc_vision"]
    ov = dict(
        device="cpu", compile=False, num_actions=s["A"], dyn_stoch=s["stoch"], dyn_discrete=s["discrete"],
        dyn_deter=s["deter"], dyn_hidden=s["hidden"], units=s["units"], batch_size=s["B"],
        batch_length=s["T"], imag_horizon=s["H"], imag_gradient=s["imag_gradient"],
        encoder=dict(cnn_depth=s["cnn_depth"]), decoder=dict(cnn_depth=s["cnn_depth"]),
        causal_world_model=False, imag_gradient_mix=s.get("imag_gradient_mix", 0.0),
        actor=dict(layers=s.get("actor_layers", 2)), critic=dict(layers=s.get("critic_layers", 2)),
        reward_head=dict(layers=s.get("reward_layers", 2)), cont_head=dict(layers=s.get("cont_layers", 2)),
    )
    if "p2e" in s:
        ov.update(expl_behavior="plan2explore", **s["p2e"])
    if s["actor_dist"] == "onehot":
        ov["actor"].update(dist="onehot", std="none")
    if s["encoder"] in ("mlp", "both"):
        ov["encoder"].update(mlp_units=s["enc_mlp_units"], mlp_layers=s["enc_mlp_layers"])
        ov["decoder"].update(mlp_units=s["enc_mlp_units"], mlp_layers=s["enc_mlp_layers"])
    if s["encoder"] == "both":  # vision block + the vector keys routed to the MLPs too (as the `minecraft` block does)
        keys = "|".join(k for k, _ in common.PROPRIO_KEYS)
        ov["encoder"].update(mlp_keys=keys, cnn_keys="image")
        ov["decoder"].update(mlp_keys=keys, cnn_keys="image")
    cfg = load_config(blocks, ov)
    spaces = {}
    if s["encoder"] in ("mlp", "both"):
        for k, w in common.PROPRIO_KEYS:
            spaces[k] = Space((w,))
    spaces["image"] = Space((64, 64, 3))
    spaces["is_first"] = Space((1,))
    spaces["is_terminal"] = Space((1,))
    with contextlib.redirect_stdout(io.StringIO()):
        wm = models.WorldModel(ObsSpace(spaces), None, 0, cfg)
        beh = models.ImagBehavior(cfg, wm)
    wm.requires_grad_(False)
    beh.requires_grad_(False)
    # load the deterministic weights
    w = common.make_weights(name)
    sd_wm = {k: torch.from_numpy(v) for k, v in w.items()
             if k.split(".")[0] in ("encoder", "dynamics", "heads")}
    missing = set(wm.state_dict().keys()) ^ set(sd_wm.keys())
    assert not missing, missing
    wm.load_state_dict(sd_wm)
    sd_beh = beh.state_dict()
    for k in list(sd_beh.keys()):
        if k.startswith("_world_model."):
            continue
        if k == "ema_vals":
            continue
        assert k in w, k
        sd_beh[k] = torch.from_numpy(w[k])
    own = {k for k in sd_beh if not k.startswith("_world_model.") and k != "ema_vals"}
    extra = {k for k in w if k.split(".")[0] in ("actor", "value", "_slow_value")} ^ own
    assert not extra, extra
    beh.load_state_dict(sd_beh)
    return cfg, wm, beh, w


def to_np(x):
    return x.detach().cpu().numpy()


def run_config(name, tools, networks, models, full: bool):
    """full=True: store every tensor (tiny configs).  full=False: slices + checksums."""
    global TAPE
    s = common.SHAPES[name]
    cfg, wm, beh, w = build_reference(name, tools, networks, models)
    data = common.make_batch(name)
    noise = common.make_noise(name)
    out = {}
    quiet = contextlib.redirect_stdout(io.StringIO())

    def keep(key, arr, rows=None):
        arr = np.asarray(arr)
        out["sum/" + key] = common.checksum(arr)
        if full:
            out[key] = arr
        elif rows is not None:
            out[key] = arr[rows]

    # ---- world model forward, piece by piece (reference modules, injected noise) ------------
    for prm in list(wm.parameters()) + list(beh.parameters()):
        prm.requires_grad_(True)
    TAPE = Tape(common.observe_tape(noise))
    with quiet:
        obs = wm.preprocess({k: v.copy() for k, v in data.items()})
    embed = wm.encoder(obs)
    action_in = obs["action"].clone()
    post, prior = wm.dynamics.observe(embed, action_in, obs["is_first"])
    assert TAPE.pos == len(TAPE.arrays)
    kl_loss, kl_value, dyn_loss, rep_loss = wm.dynamics.kl_loss(
        post, prior, cfg.kl_free, cfg.dyn_scale, cfg.rep_scale)
    feat = wm.dynamics.get_feat(post)
    losses = {}
    preds = {}
    for hname, head in wm.heads.items():
        pred = head(feat)
        if isinstance(pred, dict):
            preds.update(pred)
        else:
            preds[hname] = pred
    for k, pred in preds.items():
        losses[k] = -pred.log_prob(obs[k])
    model_loss = torch.mean(sum(losses.values()) + kl_loss)
    wm_params = dict(wm.named_parameters())
    grads = torch.autograd.grad(model_loss, list(wm_params.values()), allow_unused=True)

    sel = slice(0, 2)
    keep("embed", to_np(embed), sel)
    for k in ("stoch", "deter", "logit"):
        keep("post/" + k, to_np(post[k]), sel)
        keep("prior/" + k, to_np(prior[k]), sel)
    keep("action_after", to_np(action_in))  # obs_step zeroes prev_action at is_first rows in place
    if "image" in preds:
        recon = to_np(preds["image"].mode())
        keep("recon", recon, (slice(0, 1), slice(0, 2)))
    if s["encoder"] in ("mlp", "both"):
        for k, _ in common.PROPRIO_KEYS:
            keep("recon/" + k, to_np(preds[k]._mode))
    keep("reward_logits", to_np(preds["reward"].logits), sel)
    keep("cont_logit", to_np(preds["cont"]._dist.base_dist.logits), sel)
    for k, v in losses.items():
        out["loss/" + k] = to_np(v)
    out["kl_value"] = to_np(kl_value)
    out["dyn_loss"] = to_np(dyn_loss)
    out["rep_loss"] = to_np(rep_loss)
    out["model_loss"] = to_np(model_loss)
    out["prior_ent"] = to_np(wm.dynamics.get_dist(prior).entropy())
    out["post_ent"] = to_np(wm.dynamics.get_dist(post).entropy())
    gn = 0.0
    for (k, _), g in zip(wm_params.items(), grads):
        assert g is not None, k
        keep("grad/" + k, to_np(g))
        gn += float((g.double() ** 2).sum())
    out["model_grad_norm"] = np.float64(np.sqrt(gn))

    # ---- behaviour forward (reference modules, injected noise) --------------------------------
    start = {k: v.detach() for k, v in post.items()}
    objective = lambda f, st, a: wm.heads["reward"](wm.dynamics.get_feat(st)).mode()  # dreamer.py:196-198
    TAPE = Tape(common.imagine_tape(noise))
    feats, states, actions = beh._imagine(start, beh.actor, cfg.imag_horizon)
    assert TAPE.pos == len(TAPE.arrays)
    rows = (slice(None), slice(0, 8))
    keep("imag/feat", to_np(feats), rows)
    keep("imag/action", to_np(actions), rows)
    for k in ("stoch", "deter", "logit"):
        keep("imag/" + k, to_np(states[k]), rows)
    reward = objective(feats, states, actions)
    actor_ent = beh.actor(feats).entropy()
    ema0 = beh.ema_vals.clone()
    target, weights, base = beh._compute_target(feats, states, reward)
    actor_loss, mets = beh._compute_actor_loss(feats, actions, target, weights, base)
    actor_loss = actor_loss - cfg.actor["entropy"] * actor_ent[:-1, ..., None]
    actor_loss = torch.mean(actor_loss)
    value = beh.value(feats[:-1].detach())
    tgt = torch.stack(target, dim=1)
    value_loss = -value.log_prob(tgt.detach())
    slow = beh._slow_value(feats[:-1].detach())
    value_loss = value_loss - value.log_prob(slow.mode().detach())
    value_loss = torch.mean(weights[:-1] * value_loss[:, :, None])
    keep("imag/reward", to_np(reward), rows)
    keep("imag/actor_ent", to_np(actor_ent), rows)
    keep("imag/target", to_np(tgt), rows)
    keep("imag/weights", to_np(weights), rows)
    keep("imag/value", to_np(beh.value(feats).mode()), rows)
    out["ema_vals_after"] = to_np(beh.ema_vals)
    out["ema_vals_before"] = to_np(ema0)
    out["actor_loss"] = to_np(actor_loss)
    out["value_loss"] = to_np(value_loss)
    a_params = dict(beh.actor.named_parameters())
    v_params = dict(beh.value.named_parameters())
    ga = torch.autograd.grad(actor_loss, list(a_params.values()), retain_graph=True)
    gv = torch.autograd.grad(value_loss, list(v_params.values()))
    for (k, _), g in zip(a_params.items(), ga):
        keep("grad/actor." + k, to_np(g))
    for (k, _), g in zip(v_params.items(), gv):
        keep("grad/value." + k, to_np(g))
    out["actor_grad_norm"] = np.float64(np.sqrt(sum(float((g.double() ** 2).sum()) for g in ga)))
    out["value_grad_norm"] = np.float64(np.sqrt(sum(float((g.double() ** 2).sum()) for g in gv)))

    # ---- one full reference update through the reference's own _train (optimizer included) ------
    for prm in list(wm.parameters()) + list(beh.parameters()):
        prm.requires_grad_(False)
    beh.ema_vals.copy_(ema0)
    TAPE = Tape(common.observe_tape(noise))
    with quiet:
        post_t, context, mets_wm = wm._train({k: v.copy() for k, v in data.items()})
    assert TAPE.pos == len(TAPE.arrays)
    for k in ("model_loss", "model_grad_norm", "kl", "prior_ent", "post_ent"):
        out["train/" + k] = np.asarray(mets_wm[k], np.float64)
    # the observe inside _train must equal the piecewise one above (same weights, same noise)
    assert torch.equal(post_t["logit"], post["logit"].detach())
    TAPE = Tape(common.imagine_tape(noise))
    reward_fn = lambda f, st, a: wm.heads["reward"](wm.dynamics.get_feat(st)).mode()
    with quiet:
        mets_b = beh._train(post_t, reward_fn)[-1]
    assert TAPE.pos == len(TAPE.arrays)
    for k in ("actor_loss", "actor_grad_norm", "value_loss", "value_grad_norm", "actor_entropy",
              "EMA_005", "EMA_095", "target_mean", "target_std", "imag_reward_mean", "value_mean"):
        out["train/" + k] = np.asarray(mets_b[k], np.float64)
    # NOTE: beh._train ran AFTER the world model's Adam step, so its numbers correspond to the
    # updated world model (exactly the dreamer.py:194-200 order).  Post-update parameters:
    sd = {**{k: v for k, v in wm.state_dict().items()},
          **{k: v for k, v in beh.state_dict().items() if not k.startswith("_world_model.")}}
    for k, v in sd.items():
        keep("after/" + k, to_np(v))
    TAPE = None

    out["meta/name"] = np.array(name)
    out["meta/full"] = np.array(full)
    if full:
        for k, v in data.items():
            out["data/" + k] = v
        for k, v in noise.items():
            out["noise/" + k] = v
        for k, v in w.items():
            out["w/" + k] = v
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays; "
          f"model_loss={float(out['model_loss']):.6f} actor_loss={float(out['actor_loss']):.6f} "
          f"value_loss={float(out['value_loss']):.6f}")


def run_p2e(name, tools, networks, models):
    """One exploration update of the reference's exploration.Plan2Explore (exploration.py:86-135) after the world
    model's own update, exactly as dreamer.py:194-203 orders them: wm._train(data) -> p2e.train(start, context, data).
    Stored: the ensemble loss / gradients / post-Adam parameters, the intrinsic reward on the imagined states, and
    the exploration actor's and critic's losses, gradients and post-Adam parameters."""
    global TAPE
    sys.path.insert(0, REF)
    import exploration

    s = common.SHAPES[name]
    cfg, wm, beh, w = build_reference(name, tools, networks, models)
    extr = lambda f, st, a: wm.heads["reward"](f).mean()  # dreamer.py:80
    with contextlib.redirect_stdout(io.StringIO()):
        p2e = exploration.Plan2Explore(cfg, wm, extr)
    p2e.requires_grad_(False)
    pw = common.make_p2e_weights(name)
    sd = p2e.state_dict()
    # (`actor.*` are aliases of `_behavior.actor.*`: exploration.py:47 registers the same module twice)
    own = {k for k in sd if not k.startswith(("_behavior._world_model.", "actor.")) and k != "_behavior.ema_vals"}
    assert own == set(pw), own ^ set(pw)
    for k in own:
        assert tuple(sd[k].shape) == pw[k].shape, (k, sd[k].shape, pw[k].shape)
        sd[k] = torch.from_numpy(pw[k])
        if k.startswith("_behavior.actor."):
            sd[k[len("_behavior."):]] = sd[k]
    p2e.load_state_dict(sd)
    data = common.make_batch(name)
    noise = common.make_noise(name)
    noise_x = common.make_noise(name, seed=5)  # the exploration behaviour's own imagination draws
    out = {}
    quiet = contextlib.redirect_stdout(io.StringIO())

    # gradients as the optimizers see them (after clipping, which is inactive at these norms), captured at Adam.step
    grabbed = {}

    def grab(tag, named):
        def hook(opt, args, kwargs):
            for k, prm in named:
                grabbed[f"{tag}{k}"] = prm.grad.detach().clone()
        return hook

    p2e._expl_opt._opt.register_step_pre_hook(grab("_networks.", list(p2e._networks.named_parameters())))
    p2e._behavior._actor_opt._opt.register_step_pre_hook(grab("_behavior.actor.", list(p2e._behavior.actor.named_parameters())))
    p2e._behavior._value_opt._opt.register_step_pre_hook(grab("_behavior.value.", list(p2e._behavior.value.named_parameters())))
    seen = {}
    stock_reward = p2e._intrinsic_reward

    def spy(feat, state, action):
        r = stock_reward(feat, state, action)
        seen["reward"], seen["feat"], seen["action"] = r.detach().clone(), feat.detach().clone(), action.detach().clone()
        return r

    p2e._intrinsic_reward = spy

    TAPE = Tape(common.observe_tape(noise))
    with quiet:
        post, context, _ = wm._train({k: v.copy() for k, v in data.items()})
    assert TAPE.pos == len(TAPE.arrays)
    TAPE = Tape(common.imagine_tape(noise_x))
    with quiet:
        _, mets = p2e.train(post, context, {k: v.copy() for k, v in data.items()})
    assert TAPE.pos == len(TAPE.arrays)
    TAPE = None
    for k in ("explorer_loss", "explorer_grad_norm", "actor_loss", "actor_grad_norm", "value_loss", "value_grad_norm",
              "actor_entropy", "EMA_005", "EMA_095", "imag_reward_mean", "imag_reward_std", "target_mean",
              "value_mean"):
        out["train/" + k] = np.asarray(mets[k], np.float64)
    out["imag/reward"] = to_np(seen["reward"])
    out["imag/feat"] = to_np(seen["feat"])
    out["imag/action"] = to_np(seen["action"])
    for k, g in grabbed.items():
        out["grad/" + k] = to_np(g)
    for k, v in p2e.state_dict().items():
        if not k.startswith(("_behavior._world_model.", "actor.")):
            out["after/" + k] = to_np(v)
    out["post/stoch"], out["post/deter"] = to_np(post["stoch"]), to_np(post["deter"])
    out["meta/name"] = np.array(name)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays; explorer_loss="
          f"{float(out['train/explorer_loss']):.6f} actor_loss={float(out['train/actor_loss']):.6f} "
          f"reward mean {float(out['imag/reward'].mean()):.6f}")


def run_video(name, tools, networks, models):
    """WorldModel.video_pred of the reference (models.py:192-213) with injected noise -> {name}_video.npz
    (full output for the tiny config; checksum + the first sequence's rows 0/4/5/T-1 otherwise)."""
    global TAPE
    cfg, wm, beh, w = build_reference(name, tools, networks, models)
    data = common.make_batch(name)
    noise = common.make_video_noise(name)
    TAPE = Tape(common.video_tape(noise))
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        video = wm.video_pred({k: v.copy() for k, v in data.items()})
    assert TAPE.pos == len(TAPE.arrays)
    TAPE = None
    v = to_np(video)
    out = {"sum/video": common.checksum(v), "meta/shape": np.array(v.shape)}
    if name.startswith("tiny"):
        out["video"] = v
    else:
        T = v.shape[1]
        out["video_rows"] = v[0, [0, 4, 5, T - 1]]
    path = os.path.join(HERE, f"{name}_video.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, video {v.shape}, mean {v.mean():.6f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tools, networks, models = import_reference()
    install_noise_hooks(tools)
    plan = [("tiny", True), ("tiny_onehot", True), ("tiny_proprio", True), ("tiny_both", True), ("tiny_mixed", True),
            ("cfg2", False),
            ("cfg1", False), ("cfg3", False), ("cfg4_b4", False), ("cfg5_b4", False)]
    for name, full in plan:
        if args.only and name != args.only:
            continue
        run_config(name, tools, networks, models, full)
    for name in ("tiny", "cfg2"):
        if args.only in (None, name + "_video"):
            run_video(name, tools, networks, models)
    for name in ("tiny_p2e", "tiny_p2e_ac"):
        if args.only in (None, name):
            run_p2e(name, tools, networks, models)


if __name__ == "__main__":
    main()
