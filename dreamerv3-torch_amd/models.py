"""MI355X-native `models` module: RewardEMA, WorldModel, ImagBehavior with the reference's surface.

`WorldModel._train(data)` and `ImagBehavior._train(start, objective)` are the hot path
(models.py:108-171, 327-446 in the reference).  Here each is one explicit forward + hand-derived
backward over libdv3hip kernels (dv3hip.engine), gradients landing in flat per-optimizer buckets,
followed by one all-reduce (data parallel) and one fused clip+Adam launch per optimizer.  There
is no autograd graph, no host synchronisation and no allocation in steady state.

Row order: activations are time-major inside (row t*B+b); dict entries handed back to callers are
[B,T,...] views.  Metrics are device scalars wrapped so that float()/np.mean() work lazily.
"""
from __future__ import annotations

import copy

import numpy as np
import torch
from torch import nn

import networks
import tools
from dv3hip import _dev
from dv3hip import engine as E
from dv3hip import ops
from dv3hip.params import ParamBucket

to_np = lambda x: x.detach().cpu().numpy()
# development switches (live only with a `build.py --dev` library, dv3hip/_dev.py; see _imagine_fwd)
_FUSED_IMAG = _dev.flag("DV3_FUSED_IMAG", True)  # fused launches + one-hot gather in the rollout (off: one launch per op)
_STACK_DETER = _dev.flag("DV3_STACK_DETER", True)  # img_out + actor layer-0 dense half as one GEMM


# World model -> the behaviours built on it / the UpdateRunner its _train shares with them.  Kept out of the modules'
# __dict__ (weak references and hipGraphs do not pickle or deep-copy; a module must).
import weakref  # noqa: E402

_BEHAVIORS = weakref.WeakKeyDictionary()
_RUNNERS = weakref.WeakKeyDictionary()


def runner_of(world_model):
    """The dv3hip.graph.UpdateRunner that WorldModel._train / ImagBehavior._train of this world model go through."""
    return _RUNNERS.get(world_model)


def share_runner(world_model, runner):
    _RUNNERS[world_model] = runner


class DeviceScalar:
    """A metric that stays on the GPU until somebody looks at it (one D2H copy at log time instead of a
    sync per update)."""

    __slots__ = ("_t",)

    def __init__(self, t):
        self._t = t

    def __float__(self):
        return float(self._t.item())

    def __array__(self, dtype=None, copy=None):
        a = self._t.detach().cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def item(self):
        return self._t.item()

    def __repr__(self):
        return f"DeviceScalar({float(self):.6g})"


def _wrap(metrics):
    """Metric tensors live in workspaces that the next update overwrites: snapshot them -- all in ONE stacked copy
    (a clone per metric was ~30 copy launches per update) -- and hand out lazily-converted scalars."""
    keys = [k for k, v in metrics.items() if isinstance(v, torch.Tensor)]
    out = dict(metrics)
    if keys:
        snap = torch.stack([metrics[k].detach().reshape(()).to(torch.float32) for k in keys])
        for i, k in enumerate(keys):
            out[k] = DeviceScalar(snap[i])
    return out


class RewardEMA:
    """models.py:11-26: EMA of the 5% / 95% quantiles of the lambda-returns."""

    def __init__(self, device, alpha=1e-2):
        self.device, self.alpha = device, alpha
        self.range = torch.tensor([0.05, 0.95], device=device)

    def quantiles(self, x):
        """torch.quantile(x.flatten(), [0.05, 0.95]) (models.py:20-21) by exact radix selection on the device."""
        return ops.quantile2_ema(x.detach().contiguous(), 0.05, 0.95, out_q=torch.empty(2, device=x.device))

    def __call__(self, x, ema_vals):
        ops.quantile2_ema(x.detach().contiguous(), 0.05, 0.95, ema=ema_vals, alpha=self.alpha)
        scale = torch.clip(ema_vals[1] - ema_vals[0], min=1.0)
        return ema_vals[0].detach(), scale.detach()


class WorldModel(nn.Module):
    def __init__(self, obs_space, act_space, step, config):
        super().__init__()
        self._step = step
        if config.precision != 32:
            raise NotImplementedError("the path is fp32 (configs.yaml:18, parity at 1e-4)")
        self._config = config
        shapes = {k: tuple(v.shape) for k, v in obs_space.spaces.items()}
        self.encoder = networks.MultiEncoder(shapes, **config.encoder)
        self.embed_size = self.encoder.outdim
        self.dynamics = networks.RSSM(config.dyn_stoch, config.dyn_deter, config.dyn_hidden, config.dyn_rec_depth,
                                      config.dyn_discrete, config.act, config.norm, config.dyn_mean_act,
                                      config.dyn_std_act, config.dyn_min_std, config.unimix_ratio, config.initial,
                                      config.num_actions, self.embed_size, config.device)
        self.heads = nn.ModuleDict()
        feat_size = config.dyn_stoch * config.dyn_discrete + config.dyn_deter
        self.heads["decoder"] = networks.MultiDecoder(feat_size, shapes, **config.decoder)
        self.heads["reward"] = networks.MLP(
            feat_size, (255,) if config.reward_head["dist"] == "symlog_disc" else (), config.reward_head["layers"],
            config.units, config.act, config.norm, dist=config.reward_head["dist"],
            outscale=config.reward_head["outscale"], device=config.device, name="Reward")
        self.heads["cont"] = networks.MLP(
            feat_size, (), config.cont_head["layers"], config.units, config.act, config.norm, dist="binary",
            outscale=config.cont_head["outscale"], device=config.device, name="Cont")
        if config.reward_head["dist"] != "symlog_disc":
            raise NotImplementedError("reward head: symlog_disc only")
        for name in config.grad_heads:
            assert name in self.heads, name
        self._model_opt = tools.Optimizer("model", list(self.parameters()), config.model_lr, config.opt_eps,
                                          config.grad_clip, config.weight_decay, opt=config.opt, use_amp=False)
        self._scales = dict(reward=config.reward_head["loss_scale"], cont=config.cont_head["loss_scale"])

    # ------------------------------------------------------------------------------------------
    def preprocess(self, obs):
        """models.py:174-190 (public API: float32 tensors on the device, image/255, cont)."""
        dev = self._config.device
        obs = {k: torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(dev).to(torch.float32)
               for k, v in obs.items()}
        obs["image"] = obs["image"] / 255.0
        if "discount" in obs:
            obs["discount"] = (obs["discount"] * self._config.discount).unsqueeze(-1)
        assert "is_first" in obs and "is_terminal" in obs
        obs["cont"] = (1.0 - obs["is_terminal"]).unsqueeze(-1)
        return obs

    def _stage(self, data):
        """Host batch -> device, images kept as uint8 (12.6 MB instead of the reference's 50 MB of f32,
        models.py:176-180); normalisation happens inside the kernels."""
        dev = torch.device(self._config.device)
        out = {}
        for k, v in data.items():
            if isinstance(v, torch.Tensor):
                t = v.to(dev, non_blocking=True)
            else:
                t = torch.from_numpy(np.ascontiguousarray(v)).to(dev, non_blocking=True)
            if k == "image":
                if t.dtype != torch.uint8:
                    t = torch.clamp(torch.round(t.to(torch.float32)), 0, 255).to(torch.uint8)
                out[k] = t.contiguous()
            else:
                out[k] = t.to(torch.float32).contiguous()
        return out

    # ------------------------------------------------------------------------------------------
    @tools.on_config_device
    def _train(self, data, noise=None):
        """One world-model update (models.py:108-171) -> (post, context, metrics).  `noise` (tests): dict(q_prior,
        q_post) of [T,B,S,D] Exp(1) draws.  A driver written against the reference's classes (its dreamer.py:192-199)
        calls this and then ImagBehavior._train(post, reward): once a behaviour exists both go through one
        dv3hip.graph.UpdateRunner, which replays the update from hipGraphs after two eager warm-up calls (config key
        `hip_graph`, default on) -- the eager launch sequence is host-bound (19.5 ms per cfg-2 update against 16.2).
        Aliasing: `post` / `context` are views of the scan's buffers, which the NEXT world-model update rewrites in place
        (the behaviour of the same update reads them there); a caller that keeps them across a later update clones
        them.  The metrics are per-call snapshots."""
        r = self._auto_runner() if noise is None else None
        if r is None:
            return self._train_eager(data, noise)
        return r.train_wm(data)

    def _auto_runner(self):
        r = _RUNNERS.get(self)
        if r is None:
            cfg = self._config
            beh = next((b() for b in _BEHAVIORS.get(self, []) if b() is not None), None)
            if (beh is None or not bool(getattr(cfg, "hip_graph", True)) or not torch.cuda.is_available()
                    or torch.device(cfg.device).type != "cuda"):
                return None
            from dv3hip.graph import UpdateRunner

            r = UpdateRunner(self, beh, use_graph=True)
            _RUNNERS[self] = r
        return r

    def _train_eager(self, data, noise=None):
        """= train_fwd_bwd (forward, losses, explicit backward into the flat gradient bucket) + train_opt (all-reduce,
        clip, Adam); split so that a hipGraph can hold each half with the collective between them."""
        self.train_fwd_bwd(data, noise)
        return self.train_opt()

    def train_opt(self, allreduce=True):
        post, context, metrics, loss = self._pending
        metrics = dict(metrics)
        metrics.update(self._model_opt.finish(loss, allreduce))
        return post, context, _wrap(metrics)

    def train_fwd_bwd(self, data, noise=None):
        cfg = self._config
        st = self._stage(data)
        B, T = st["action"].shape[0], st["action"].shape[1]
        TB = B * T
        dev = st["action"].device
        dyn = self.dynamics
        rssm = dyn.engine
        ws = rssm.ws
        S, D, SD, De = dyn._stoch, dyn._discrete, dyn._stoch * dyn._discrete, dyn._deter
        nz = noise or {}
        rng = dyn._rng()
        self._model_opt.begin()

        # ---- inputs to time-major
        act_tm = ops.transpose01(st["action"], ws.get("wm.action", (T, B, st["action"].shape[2])))
        first_tm = ops.transpose01(st["is_first"], ws.get("wm.first", (T, B)))
        reward_tm = ops.transpose01(st["reward"], ws.get("wm.reward", (T, B)))
        cont_tm = ws.get("wm.cont", (T, B))
        ops.transpose01(st["is_terminal"], cont_tm)
        ops.axpby(ws.get("wm.ones", (T, B)).fill_(1.0), cont_tm, 1.0, -1.0)  # cont = 1 - is_terminal

        # ---- encoder (networks.MultiEncoder.forward, networks.py:348-356: the CNN on the image, the MLP on the symlog of
        # the vector keys, their outputs side by side)
        enc = self.encoder
        cnn_eng = mlp_eng = None
        parts = []
        if enc.cnn_shapes:
            cnn_eng = enc._cnn.engine
            parts.append(cnn_eng.forward(st["image"], (B, T)))
        if enc.mlp_shapes:
            keys = list(enc.mlp_shapes)
            widths = [int(np.prod(enc.mlp_shapes[k])) for k in keys]
            xin = ws.get("wm.mlp_in", (TB, sum(widths)))
            raw = ws.get("wm.mlp_raw", (TB, sum(widths)))
            off = 0
            for k, w in zip(keys, widths):
                tmp = ops.transpose01(st[k].reshape(B, T, w), ws.get(f"wm.key.{k}", (T, B, w)))
                raw[:, off:off + w] = tmp.view(TB, w)
                off += w
            ops.symlog(raw, xin)
            mlp_eng = enc._mlp.engine_for(".wm")
            parts.append(mlp_eng.forward(xin)[0])
            self._enc_in = xin
        if len(parts) == 2:
            # image + vector keys together (the `minecraft` block, configs.yaml:206-207): embed = [cnn | mlp]
            Ec, Em = parts[0].shape[1], parts[1].shape[1]
            embed = ws.get("wm.embed", (TB, Ec + Em))
            embed[:, :Ec].copy_(parts[0])
            embed[:, Ec:].copy_(parts[1])
        else:
            embed = parts[0]
        E_ = embed.shape[1]

        # ---- RSSM scan
        force = None
        if "force_post" in nz or "force_prior" in nz:  # parity tests: teacher-forced draws + flip counter
            force = dict(post=nz.get("force_post"), prior=nz.get("force_prior"), flips=nz.get("flips"))
        out = rssm.observe_fwd(embed.view(T, B, E_), act_tm, first_tm, q_prior=nz.get("q_prior"),
                               q_post=nz.get("q_post"), rng=rng, force=force)
        ps, dt = out["post_stoch"].view(TB, SD), out["deter"].view(TB, De)
        kl = ws.get("wm.kl", (T, B))
        ent_p, ent_q = ws.get("wm.ent_post", (T, B)), ws.get("wm.ent_prior", (T, B))
        ops.kl_fwd(out["post_logit"], out["prior_logit"], kl, ent_p, ent_q, unimix=dyn._unimix_ratio)
        # the reverse scan beside the deferred weight gradients on two CU-masked streams, where that pays
        lanes_pay = rssm.lanes_pay(heavy_side=bool(self.heads["decoder"].cnn_shapes))
        E.SideStream.host_sync_point(lanes_pay)  # (captured update: the lanes are launched once the GPU is here)

        # ---- heads, losses and their upstream gradients (loss = mean over B*T of the per-row sum)
        up = 1.0 / TB
        dec = self.heads["decoder"]
        # [model_loss, image, reward, cont, kl, ent_prior, ent_post, spare, one slot per vector-decoder key ...]
        acc = ws.zeros("wm.acc", (8 + len(dec.mlp_shapes),))
        gs, gd = ws.get("wm.gs", (T, B, SD)), ws.get("wm.gd", (T, B, De))
        wrote = False
        deferred = []  # weight-gradient launches that can run beside the reverse scan (E.SideStream)
        grad_heads = cfg.grad_heads
        if dec.cnn_shapes:
            dec_eng = dec._cnn.engine
            recon = dec_eng.forward(ps, dt)
            limg = ws.get("wm.loss_img", (TB,))
            drecon = ws.get("wm.drecon", recon.shape)
            ops.mse_image(recon, st["image"], limg, drecon, upstream=up, perm=(B, T))
            ops.dot_accumulate(limg, acc[1:2], scale=up)
            dec_eng.backward(drecon, gs.view(TB, SD), gd.view(TB, De), acc_dx=False, defer=deferred)
            # (pipelined update, lanes mode: the deferred launches behind this index -- the heads' weight gradients -- may
            # run on the world model's own lane, graph.UpdateRunner._capture_pipe)
            self._defer_heads_from = len(deferred)
            wrote = True
            if "decoder" not in grad_heads:
                gs.zero_(), gd.zero_()
        if dec.mlp_shapes:
            mdec = dec._mlp
            deng = mdec.engine_for(".wm")
            h, _, _ = deng.forward(ps, dt)
            dh = ws.zeros("wm.dec_dh", h.shape)
            for ki, (k, shp) in enumerate(dec.mlp_shapes.items()):
                lin = mdec.mean_layer[k]
                w = int(np.prod(shp))
                mode = ws.get(f"wm.dec.{k}", (TB, w))
                ops.gemm(h, lin.weight, mode, bias=lin.bias)
                tgt = ops.transpose01(st[k].reshape(B, T, w), ws.get(f"wm.key.{k}", (T, B, w))).view(TB, w)
                lk = ws.get(f"wm.loss.{k}", (TB,))
                dmode = ws.get(f"wm.dmode.{k}", (TB, w))
                ops.symlog_mse(mode, tgt, lk, dmode, upstream=up)
                ops.dot_accumulate(lk, acc[8 + ki:9 + ki], scale=up)
                ops.gemm(dmode, lin.weight, dh, transB=False, accumulate=True)
                E.lin_wgrad(lin.weight, dmode, h)
                ops.colsum(dmode, lin.bias.grad, accumulate=True)
            gh = "decoder" in grad_heads
            deng.backward(ps, dt, slice(0, TB), wgrad=True, dh=dh, dx1=gs.view(TB, SD) if gh else None,
                          dx2=gd.view(TB, De) if gh else None, acc_dx=wrote, defer=deferred)
            wrote = wrote or gh
        if not wrote:
            gs.zero_(), gd.zero_()
        # reward head: -log_prob(reward) under the 255-bucket two-hot head
        pidx = out["post_idx"].view(TB, S) if E._GATHER_OBS else None
        reng = self.heads["reward"].engine_for(".wm")
        if pidx is not None:
            reng.pack_onehot(SD)
        _, r_logits, _ = reng.forward(ps, dt, idx=pidx, D=D)
        lp_r = ws.get("wm.lp_r", (TB,))
        ops.disc_logprob_fwd(r_logits, reward_tm.view(TB), lp_r)
        ops.dot_accumulate(lp_r, acc[2:3], scale=-up)
        dr = ws.get("wm.dr_logits", r_logits.shape)
        up_r = ws.get("wm.up_r", (TB,)).fill_(-up * self._scales["reward"])
        ops.disc_logprob_bwd(r_logits, reward_tm.view(TB), up_r, dr)
        g_r = "reward" in grad_heads
        reng.backward(ps, dt, slice(0, TB), dout=dr, wgrad=True, dx1=gs.view(TB, SD) if g_r else None,
                      dx2=gd.view(TB, De) if g_r else None, acc_dx=True, defer=deferred)
        # continue head
        ceng = self.heads["cont"].engine_for(".wm")
        if pidx is not None:
            ceng.pack_onehot(SD)
        _, c_logit, _ = ceng.forward(ps, dt, idx=pidx, D=D)
        lp_c = ws.get("wm.lp_c", (TB,))
        ops.bernoulli_logprob_fwd(c_logit.view(TB), cont_tm.view(TB), lp_c)
        ops.dot_accumulate(lp_c, acc[3:4], scale=-up)
        dc = ws.get("wm.dc_logit", (TB, 1))
        up_c = ws.get("wm.up_c", (TB,)).fill_(-up * self._scales["cont"])
        ops.bernoulli_logprob_bwd(c_logit.view(TB), cont_tm.view(TB), up_c, dc.view(TB))
        g_c = "cont" in grad_heads
        ceng.backward(ps, dt, slice(0, TB), dout=dc, wgrad=True, dx1=gs.view(TB, SD) if g_c else None,
                      dx2=gd.view(TB, De) if g_c else None, acc_dx=True, defer=deferred)
        # KL
        dpl, dql = ws.get("wm.dpost_logit", (T, B, S, D)), ws.get("wm.dprior_logit", (T, B, S, D))
        ops.kl_bwd(out["post_logit"], out["prior_logit"], kl, dpl, dql, unimix=dyn._unimix_ratio, free=cfg.kl_free,
                   dyn_scale=cfg.dyn_scale, rep_scale=cfg.rep_scale, upstream=up)
        ops.dot_accumulate(kl.view(TB), acc[4:5], clip_min=cfg.kl_free, scale=up)  # mean of the clipped KL
        # ---- backward through the scan and the encoder
        dembed = ws.get("wm.dembed", (T, B, E_))
        side = rssm.observe_bwd(dpl, dql, gs, gd, dembed, extra_side=deferred, lanes_pay=lanes_pay)
        de = dembed.view(TB, E_)
        if cnn_eng is not None and mlp_eng is not None:
            Ec = parts[0].shape[1]
            d_cnn, d_mlp = ws.get("wm.dembed_cnn", (TB, Ec)), ws.get("wm.dembed_mlp", (TB, E_ - Ec))
            d_cnn.copy_(de[:, :Ec])
            d_mlp.copy_(de[:, Ec:])
        else:
            d_cnn = d_mlp = de
        with ops.gemm_group():  # (the encoder's dense weight gradients as one grid: their operands stay put until the join)
            if cnn_eng is not None:
                cnn_eng.backward(d_cnn)
            if mlp_eng is not None:
                mlp_eng.backward(self._enc_in, None, slice(0, TB), wgrad=True, dh=d_mlp)
        side.join()

        # ---- scalar loss + optimizer
        loss = ws.get("wm.loss", (1,))
        ops.dot_accumulate(kl.view(TB), ws.zeros("wm.kl_mean", (1,)), scale=up)
        ops.dot_accumulate(ent_q.view(TB), acc[5:6], scale=up)
        ops.dot_accumulate(ent_p.view(TB), acc[6:7], scale=up)
        # model_loss = image + vector + reward*scale + cont*scale + (dyn_scale + rep_scale) * clipped KL
        loss.copy_(acc[1:2] + acc[8:].sum(0, keepdim=True) + self._scales["reward"] * acc[2:3]
                   + self._scales["cont"] * acc[3:4] + (cfg.dyn_scale + cfg.rep_scale) * acc[4:5])
        metrics = {}
        if dec.cnn_shapes:
            metrics["image_loss"] = acc[1]
        for ki, k in enumerate(dec.mlp_shapes):
            metrics[f"{k}_loss"] = acc[8 + ki]  # each key's own -log_prob mean (models.py:150)
        metrics.update(reward_loss=acc[2], cont_loss=acc[3], kl_free=cfg.kl_free, dyn_scale=cfg.dyn_scale,
                       rep_scale=cfg.rep_scale, dyn_loss=acc[4], rep_loss=acc[4], kl=ws.get("wm.kl_mean", (1,))[0],
                       prior_ent=acc[5], post_ent=acc[6])
        bt = lambda x: x.transpose(0, 1)
        post = {"stoch": bt(out["post_stoch"]), "deter": bt(out["deter"]), "logit": bt(out["post_logit"])}
        self._last = dict(out=out, embed=embed.view(T, B, E_), kl=kl, ent_post=ent_p, action_tm=out["action"])
        context = _LazyContext(self, post)
        rng.commit()
        self._pending = (post, context, metrics, loss[0])

    @tools.on_config_device
    def video_pred(self, data, noise=None):
        """models.py:192-213 (forward-only open-loop prediction for logging).  noise (tests): dict of Exp(1)
        draws q_prior, q_post [5,Bv,S,D] and q_open [T-5,Bv,S,D]; default = the Philox stream."""
        data = self.preprocess(data)
        embed = self.encoder(data)
        nz = noise or {}
        states, _ = self.dynamics.observe(embed[:6, :5], data["action"][:6, :5], data["is_first"][:6, :5],
                                          noise=noise)
        recon = self.heads["decoder"](self.dynamics.get_feat(states))["image"].mode()[:6]
        init = {k: v[:, -1] for k, v in states.items()}
        prior = self.dynamics.imagine_with_action(data["action"][:6, 5:], init, noise=nz.get("q_open"))
        openl = self.heads["decoder"](self.dynamics.get_feat(prior))["image"].mode()
        model = torch.cat([recon[:, :5], openl], 1)
        truth = data["image"][:6]
        error = (model - truth + 1.0) / 2.0
        return torch.cat([truth, model, error], 2)


class _LazyContext(dict):
    """`context` of WorldModel._train (embed, feat, kl, postent), built only if somebody reads it."""

    def __init__(self, wm, post):
        super().__init__()
        self._wm, self._post = wm, post

    def __missing__(self, key):
        last = self._wm._last
        bt = lambda x: x.transpose(0, 1)
        if key == "embed":
            v = bt(last["embed"])
        elif key == "feat":
            v = self._wm.dynamics.get_feat(self._post)
        elif key == "kl":
            v = bt(last["kl"])
        elif key == "postent":
            v = bt(last["ent_post"])
        else:
            raise KeyError(key)
        self[key] = v
        return v


class ImagBehavior(nn.Module):
    def __init__(self, config, world_model, future_predictor=None):
        super().__init__()
        self._config = config
        self._world_model = world_model
        feat_size = config.dyn_stoch * config.dyn_discrete + config.dyn_deter
        self.actor = networks.MLP(
            feat_size, (config.num_actions,), config.actor["layers"], config.units, config.act, config.norm,
            config.actor["dist"], config.actor["std"], config.actor["min_std"], config.actor["max_std"], absmax=1.0,
            temp=config.actor["temp"], unimix_ratio=config.actor["unimix_ratio"], outscale=config.actor["outscale"],
            name="Actor")
        self.value = networks.MLP(
            feat_size, (255,) if config.critic["dist"] == "symlog_disc" else (), config.critic["layers"], config.units,
            config.act, config.norm, config.critic["dist"], outscale=config.critic["outscale"], device=config.device,
            name="Value")
        # (WorldModel._train finds the behaviour that follows it in an update through this list: see _auto_runner)
        _BEHAVIORS.setdefault(world_model, []).append(weakref.ref(self))
        if config.critic["dist"] != "symlog_disc" or config.actor["dist"] not in ("normal", "onehot"):
            raise NotImplementedError("critic symlog_disc; actor normal|onehot")
        if config.imag_gradient not in ("dynamics", "reinforce", "both"):
            raise NotImplementedError(config.imag_gradient)
        if config.critic["slow_target"]:
            self._slow_value = copy.deepcopy(self.value)
            self._updates = 0
        kw = dict(wd=config.weight_decay, opt=config.opt, use_amp=False)
        self._actor_opt = tools.Optimizer("actor", list(self.actor.parameters()), config.actor["lr"],
                                          config.actor["eps"], config.actor["grad_clip"], **kw)
        # (data parallel: the two return-normalisation EMA values ride behind the critic's gradient through ITS all-reduce)
        self._value_opt = tools.Optimizer("value", list(self.value.parameters()), config.critic["lr"],
                                          config.critic["eps"], config.critic["grad_clip"],
                                          extra=2 if config.reward_EMA else 0, **kw)
        if self._config.reward_EMA:
            self.register_buffer("ema_vals", torch.zeros((2,), device=self._config.device))
            self.reward_ema = RewardEMA(device=self._config.device)

    # ------------------------------------------------------------------------------------------
    def _update_slow_target(self):
        """models.py:683-689 on the flat buckets: slow = mix*value + (1-mix)*slow."""
        if self._config.critic["slow_target"]:
            if self._updates % self._config.critic["slow_target_update"] == 0:
                mix = self._config.critic["slow_target_fraction"]
                vb = self._value_opt.bucket.ensure()
                sb = self._slow_bucket()
                ops.axpby(vb.flat, sb.flat, mix, 1.0 - mix)
            self._updates += 1

    def _slow_bucket(self):
        b = getattr(self, "_slow_b", None)
        if b is None:
            b = ParamBucket("slow_value", list(self._slow_value.parameters()))
            object.__setattr__(self, "_slow_b", b)
        return b.ensure()

    def _flat_start(self, start):
        """start {[B,T,...]} -> ([N,SD], [N,De], [N,SD]) without a copy when it is a time-major view."""
        dyn = self._world_model.dynamics
        SD = dyn._stoch * dyn._discrete
        outs = []
        for k, w in (("stoch", SD), ("deter", dyn._deter), ("logit", SD)):
            v = start[k]
            tm = v.transpose(0, 1)
            src = tm if tm.is_contiguous() else v.contiguous()
            outs.append(src.reshape(-1, w))
        return outs

    def _imagine(self, start, policy, horizon, first_action=None, noise=None):
        """models.py:448-548 (forward only): returns feats [H,N,F], states {[H,N,...]}, actions [H,N,A].
        Row order follows the memory order of `start` (time-major when it comes from WorldModel._train)."""
        if policy is not None and policy is not self.actor:
            return self._imagine_generic(start, policy, horizon)
        self._imagine_fwd(start, horizon, noise)
        st = self._im
        S, D = self._world_model.dynamics._stoch, self._world_model.dynamics._discrete
        H, N = st["H"], st["N"]
        states = {"stoch": st["stoch"].view(H, N, S, D).clone(), "deter": st["deter"].clone(),
                  "logit": st["logit"].view(H, N, S, D).clone()}
        feats = torch.cat([st["stoch"], st["deter"]], -1)
        return feats, states, st["action"].clone()

    def _imagine_generic(self, start, policy, horizon):
        """models.py:448-548 for a foreign `policy` (any callable feat -> distribution with .sample()): the rollout
        step by step through the public RSSM methods, i.e. through dv3hip.autograd when gradients are wanted, so the
        returned states / actions carry a graph as the reference's do; the returned feats are DETACHED, as the
        reference's are (models.py:513-517: `feat = get_feat(state).detach()`).  Rows are b*T+t as there."""
        dyn = self._world_model.dynamics
        state = {k: v.reshape((-1,) + tuple(v.shape[2:])) for k, v in start.items()}
        feats, states, actions = [], [], []
        for t in range(horizon):
            feat = dyn.get_feat(state).detach()
            action = policy(feat).sample()
            feats.append(feat), states.append(state), actions.append(action)
            if t < horizon - 1:  # the H-th successor is discarded by the reference (models.py:546)
                state = dyn.img_step(state, action)
        return (torch.stack(feats, 0), {k: torch.stack([s_[k] for s_ in states], 0) for k in states[0]},
                torch.stack(actions, 0))

    def _imagine_fwd(self, start, horizon, noise=None, packed=False):
        cfg = self._config
        dyn = self._world_model.dynamics
        rssm = dyn.engine
        ws = rssm.ws
        S, D, SD, De, Hd, A = dyn._stoch, dyn._discrete, dyn._stoch * dyn._discrete, dyn._deter, dyn._hidden, \
            dyn._num_actions
        s0, d0, l0 = self._flat_start(start)
        N, H = s0.shape[0], horizon
        nz = noise or {}
        rng = dyn._rng()
        g = ws.get
        stoch, deter, logit = g("im.stoch", (H, N, SD)), g("im.deter", (H, N, De)), g("im.logit", (H, N, SD))
        stoch[0].copy_(s0), deter[0].copy_(d0), logit[0].copy_(l0)
        action, ent = g("im.action", (H, N, A)), g("im.ent", (H, N))
        eps = g("im.eps", (H, N, A))
        actor_eng = self.actor.engine_for(".imag")
        U0 = actor_eng.P.layers[0].W.shape[0]
        # img_out and the dense half of the actor's first layer both read the new deter: one stacked GEMM per step
        # writes [x2pre | actor pre0 of the NEXT step] (engine.RSSMEngine.img_step_fwd, wcat)
        stack = _FUSED_IMAG and _STACK_DETER
        cat = g("im.cat", (H, N, Hd + U0)) if stack else None
        step = dict(x1pre=g("im.x1pre", (H, N, Hd)), m1=g("im.m1", (H, N)), r1=g("im.r1", (H, N)),
                    x1=g("im.x1", (H, N, Hd)), gpre=g("im.gpre", (H, N, 3 * De)), mg=g("im.mg", (H, N)),
                    rg=g("im.rg", (H, N)), x2pre=cat[..., :Hd] if stack else g("im.x2pre", (H, N, Hd)),
                    m2=g("im.m2", (H, N)), r2=g("im.r2", (H, N)), x2=g("im.x2", (H, N, Hd)))
        if stack:
            step["cat"] = cat
            wcat = g("im.wcat", (Hd + U0, De))
            wcat[:Hd].copy_(rssm.P.img_out.W)
            wcat[Hd:].copy_(actor_eng.P.layers[0].W[:, SD:])
        else:
            wcat = None
        normal = cfg.actor["dist"] == "normal"
        q_img, act_noise = nz.get("q_img"), nz.get("act")
        f_img, f_act, flips = nz.get("force_img"), nz.get("force_act"), nz.get("flips")  # parity tests
        # The stochastic state is an exact one-hot (tools.py:452-460): carry its class indices and let every Linear
        # that reads it (actor layer 0, img_in) gather weight columns instead of multiplying zeros (engine.py).
        idx = g("im.idx", (H, N, S), torch.int32)
        ops.onehot_to_idx(stoch[0].view(N, S, D), idx[0].view(-1))  # class indices of the start states
        # (pipelined capture, graph.UpdateRunner.step_pipelined: everything up to here reads the posterior of the world
        # model's scan and runs BEFORE the next update's scan overwrites it; the rollout below runs beside that scan)
        E.Cuts.mark("bh.A")
        if _FUSED_IMAG and not packed:
            tr = []
            actor_eng.pack_onehot(SD, defer=tr)
            rssm.pack_img_in(defer=tr)
            ops.transpose2d_many(tr)

        def run_chain(rc):
            """The H-step rollout of the row range rc (rows are independent: models.py:450-451 flattens [B,T])."""
            n, r0 = rc.stop - rc.start, rc.start
            for t in range(H):
                if t:
                    E.Cuts.mark(f"bh.A@{t}")  # (optional cut: the schedule may move the rest of the rollout off the lane)
                nz_act = None if act_noise is None else act_noise[t][rc]
                if _FUSED_IMAG:
                    head = dict(action=action[t][rc], ent=ent[t][rc], rng=rng, onehot=not normal, flips=flips,
                                noise=nz_act)
                    if normal:
                        head.update(eps_out=eps[t][rc], min_std=cfg.actor["min_std"], max_std=cfg.actor["max_std"])
                    else:
                        head.update(unimix=cfg.actor["unimix_ratio"], forced=None if f_act is None else f_act[t][rc])
                    actor_eng.forward(stoch[t][rc], deter[t][rc], row0=t * N + r0, total=H * N, idx=idx[t][rc], D=D,
                                      head=head, base0=cat[t - 1][rc][:, Hd:] if (stack and t > 0) else None)
                else:  # one launch per op (development switch DV3_FUSED_IMAG=0: the r01 launch sequence, for A/B)
                    _, mean_raw, std_raw = actor_eng.forward(stoch[t][rc], deter[t][rc], row0=t * N + r0, total=H * N)
                    if normal:
                        if nz_act is not None:
                            eps[t][rc].copy_(nz_act)
                        else:
                            ops.fill_normal(eps[t][rc], rng)
                        ops.actor_normal_fwd(mean_raw, std_raw, eps[t][rc], action[t][rc], ent[t][rc],
                                             min_std=cfg.actor["min_std"], max_std=cfg.actor["max_std"])
                    else:
                        ops.onehot_sample(mean_raw, action[t][rc], noise=nz_act, rng=rng,
                                          unimix=cfg.actor["unimix_ratio"],
                                          forced=None if f_act is None else f_act[t][rc], flips=flips)
                        ops.onehot_ent_logp_fwd(mean_raw, None, ent[t][rc], None, unimix=cfg.actor["unimix_ratio"])
                if t < H - 1:
                    b = {k: v[t][rc] for k, v in step.items()}
                    b.update(deter=deter[t + 1][rc], logit=logit[t + 1][rc].view(n, S, D),
                             stoch=stoch[t + 1][rc].view(n, S, D))
                    rssm.img_step_fwd(stoch[t][rc], deter[t][rc], action[t][rc], b,
                                      noise=None if q_img is None else q_img[t][rc], rng=rng,
                                      forced=None if f_img is None else f_img[t][rc], flips=flips,
                                      idx=idx[t][rc] if _FUSED_IMAG else None, idx_out=idx[t + 1][rc], wcat=wcat)

        run_chain(slice(0, N))
        rng.commit()
        self._im = dict(H=H, N=N, stoch=stoch, deter=deter, logit=logit, action=action, ent=ent, eps=eps, step=step,
                        actor=actor_eng, idx=idx)

    # ------------------------------------------------------------------------------------------
    @tools.on_config_device
    def _train(self, start, objective=None, noise=None):
        """One actor + critic update (models.py:327-446).  `objective(feat, state, action) -> reward [H,N,1]`:
        * None, or the world model's reward head on the imagined states (what dreamer.py:196-199 passes; recognised on
          its first use by evaluating it once and comparing with the fused head, then remembered per code object):
          the fully fused path -- reward head forward and backward in hand-written kernels, hipGraph-capturable;
        * anything else (exploration.Plan2Explore._intrinsic_reward, exploration.py:108-121): the objective is
          evaluated under torch autograd on leaf views of the imagined states / actions and its input gradients are
          injected into the hand-written reverse rollout where the reward head's would enter (see train_fwd_bwd)."""
        r = _RUNNERS.get(self._world_model) if noise is None else None
        if r is not None and r.beh is self and r.owns(start) and self._is_known_reward_head(objective):
            return r.train_behavior()
        return self._train_eager(start, objective, noise)

    def _is_known_reward_head(self, objective):
        """None, or an objective an earlier (eager) call has recognised as the world model's reward head."""
        if objective is None:
            return True
        return self.__dict__.get("_objective_kinds", {}).get(self._objective_key(objective)) is True

    def _train_eager(self, start, objective=None, noise=None):
        self.train_fwd_bwd(start, noise, objective)
        return self.train_opt()

    @staticmethod
    def _objective_key(objective):
        fn = getattr(objective, "__func__", objective)
        return getattr(fn, "__code__", None) or id(objective)

    def _objective_is_head(self, objective, im, reward):
        """First use of an `objective`: is it the reward head on the imagined states (then the fused backward is its
        exact gradient)?  One forward evaluation through the public modules + one host comparison, cached."""
        kinds = self.__dict__.setdefault("_objective_kinds", {})
        key = self._objective_key(objective)
        if key not in kinds:
            with torch.no_grad():
                state = self._imag_state(im)
                got = objective(self._world_model.dynamics.get_feat(state), state, im["action"])
            same = (isinstance(got, torch.Tensor) and got.numel() == reward.numel()
                    and bool(torch.allclose(got.reshape(reward.shape).to(reward.dtype), reward, rtol=1e-4, atol=1e-4)))
            kinds[key] = same
        return kinds[key]

    def _imag_state(self, im):
        S, D = self._world_model.dynamics._stoch, self._world_model.dynamics._discrete
        H, N = im["H"], im["N"]
        return {"stoch": im["stoch"].view(H, N, S, D), "deter": im["deter"], "logit": im["logit"].view(H, N, S, D)}

    def _eval_objective(self, objective, im, need_grad):
        """reward = objective(feat, state, action) on leaf views of the imagined trajectory (no copy).  With
        need_grad the call records an autograd graph whose input gradients train_fwd_bwd later pulls with the
        upstream d loss / d reward of the lambda-return backward."""
        st = self._imag_state(im)
        H, N = im["H"], im["N"]
        leaves = dict(stoch=st["stoch"].detach().requires_grad_(need_grad),
                      deter=st["deter"].detach().requires_grad_(need_grad),
                      action=im["action"].detach().requires_grad_(need_grad))
        state = {"stoch": leaves["stoch"], "deter": leaves["deter"], "logit": st["logit"].detach()}
        with torch.enable_grad() if need_grad else torch.no_grad():
            # as in the reference, `feat` is detached (models.py:513-517 returns get_feat(state).detach()): gradients
            # reach the dynamics through `state` and `action` only
            feat = self._world_model.dynamics.get_feat(state).detach()
            r = objective(feat, state, leaves["action"])
        if not isinstance(r, torch.Tensor) or r.numel() != H * N:
            raise ValueError(f"objective must return one reward per imagined state [H={H}, N={N}, 1]; got "
                             f"{tuple(getattr(r, 'shape', ()))}")
        return r, leaves

    def sync_ema(self):
        """Data parallel: the return-normalisation EMA is computed from each rank's own imagined returns and averaged
        over the ranks so that the replicas normalise alike (the reference is single-process).  The two floats travel
        in the tail of the critic's gradient bucket (ema_to_wire / ema_from_wire around ITS all-reduce: three
        collectives per update, not four); kept as a public no-op for callers of the earlier interface."""

    def _ema_on_wire(self):
        return bool(self._config.reward_EMA) and ParamBucket.distributed()

    def ema_to_wire(self):
        """After the backward, before the critic's all-reduce: this rank's EMA values into the bucket's tail."""
        if self._ema_on_wire():
            self._value_opt.bucket.ensure().tail.copy_(self.ema_vals)

    def ema_from_wire(self):
        """After the critic's all-reduce: the mean over the ranks."""
        if self._ema_on_wire():
            import torch.distributed as dist

            torch.mul(self._value_opt.bucket.tail, 1.0 / dist.get_world_size(), out=self.ema_vals)

    def train_opt(self, allreduce=True):
        ret, metrics, losses = self._pending
        metrics = dict(metrics)
        metrics.update(self._actor_opt.finish(losses[0], allreduce))
        metrics.update(self._value_opt.finish(losses[1], allreduce))
        self.ema_from_wire()
        return ret + (_wrap(metrics),)

    def train_fwd_bwd(self, start, noise=None, objective=None):
        cfg = self._config
        wm = self._world_model
        dyn = wm.dynamics
        rssm = dyn.engine
        ws = rssm.ws
        S, D, SD, De, A = dyn._stoch, dyn._discrete, dyn._stoch * dyn._discrete, dyn._deter, dyn._num_actions
        H = cfg.imag_horizon
        self._update_slow_target()
        self._actor_opt.begin()
        self._value_opt.begin()
        reng = wm.heads["reward"].engine_for(".imag")
        ceng = wm.heads["cont"].engine_for(".imag")
        veng = self.value.engine_for(".imag")
        seng = networks.MLP.engine_for(self._slow_value, ".imag")
        self._slow_bucket()
        wt_bwd = None
        if _FUSED_IMAG:
            # every transposed weight copy of this update (the one-hot gathers of the actor, img_in and the four heads,
            # the reverse rollout's data-gradient operands) in ONE launch: the world model is frozen from here on
            # (models.py:335) and the actor / critic step only after the backward below
            tr = []
            self.actor.engine_for(".imag").pack_onehot(SD, defer=tr)
            for e in (reng, ceng, veng, seng):
                e.pack_onehot(SD, defer=tr)
            if cfg.imag_gradient in ("dynamics", "both"):
                wt_bwd = rssm.pack_bwd(defer=tr)
            else:
                rssm.pack_img_in(defer=tr)
            ops.transpose2d_many(tr)
        self._imagine_fwd(start, H, noise, packed=True)
        E.Cuts.mark("bh.B")
        im = self._im
        N = im["N"]
        HN, H1N = H * N, (H - 1) * N
        stoch, deter, action, ent = im["stoch"], im["deter"], im["action"], im["ent"]
        fs, fd = stoch.view(HN, SD), deter.view(HN, De)
        g = ws.get
        # ---- heads over all H*N imagined states
        # first layers read feat = [stoch | deter]: deter through the MFMA GEMM, the one-hot stoch as a gather
        fidx = im["idx"].view(HN, S) if _FUSED_IMAG else None
        use_dyn = cfg.imag_gradient in ("dynamics", "both")
        reward = g("bh.reward", (H, N))
        custom = objective is not None and self.__dict__.get("_objective_kinds", {}).get(
            self._objective_key(objective)) is False
        r_logits = obj_out = obj_leaves = None
        if not custom:
            _, r_logits, _ = reng.forward(fs, fd, idx=fidx, D=D)
            ops.disc_mode_fwd(r_logits, reward)
            if objective is not None and not self._objective_is_head(objective, im, reward):
                custom = True
        if custom:
            # a foreign objective: evaluated with torch autograd on leaf views of the trajectory (its own modules run
            # their kernels through dv3hip.autograd); gradients are pulled out below
            obj_out, obj_leaves = self._eval_objective(objective, im, need_grad=use_dyn)
            reward.copy_(obj_out.detach().reshape(H, N))
        _, c_logit, _ = ceng.forward(fs, fd, idx=fidx, D=D)
        _, v_logits, _ = veng.forward(fs, fd, idx=fidx, D=D)
        value = ops.disc_mode_fwd(v_logits, g("bh.value", (H, N)))
        _, s_logits, _ = seng.forward(fs[:H1N], fd[:H1N], idx=None if fidx is None else fidx[:H1N], D=D)
        slow = ops.disc_mode_fwd(s_logits, g("bh.slow", (H - 1, N)))
        target, weights, disc = g("bh.target", (H - 1, N)), g("bh.weights", (H, N)), g("bh.disc", (H, N))
        ops.lambda_return_fwd(reward, value, c_logit.view(H, N), target, weights, disc, gamma=cfg.discount,
                              lam=cfg.discount_lambda)
        # ---- actor loss (+ gradients w.r.t. target / entropy / log-prob)
        if cfg.reward_EMA:
            self.reward_ema(target, self.ema_vals)
            ema = self.ema_vals
        else:
            ema = g("bh.ema_fixed", (2,))
            ema[0], ema[1] = 0.0, 1.0
        acc = ws.zeros("bh.acc", (4,))  # [actor_loss, value_loss, entropy mean, spare]
        dent = g("bh.dent", (H, N))
        # imag_gradient (models.py:663-678): 'dynamics' back-propagates the normalised return through the imagined
        # states; 'reinforce' weights log pi(a) with the detached advantage; 'both' mixes the raw return into the latter
        use_logp = cfg.imag_gradient in ("reinforce", "both")
        reinforce = not use_dyn
        normal = cfg.actor["dist"] == "normal"
        a_mean, a_std = self._actor_heads(im)
        dlogp = dtarget = logp = None
        if use_logp:
            logp = g("bh.logp", (H, N))
            if normal:
                ops.actor_normal_logp(a_mean, a_std, action.view(HN, A), logp.view(HN), min_std=cfg.actor["min_std"],
                                      max_std=cfg.actor["max_std"])
            else:
                ops.onehot_ent_logp_fwd(a_mean, action.view(HN, A), None, logp.view(HN),
                                        unimix=cfg.actor["unimix_ratio"])
            dlogp = g("bh.dlogp", (H, N))
        if use_dyn:
            dtarget = g("bh.dtarget", (H - 1, N))
        ops.actor_loss(target, value, weights, ent, ema, acc[0:1], dent, dtarget=dtarget, logp=logp, dlogp=dlogp,
                       entropy_coef=cfg.actor["entropy"], mode={"dynamics": 0, "reinforce": 1, "both": 2}[cfg.imag_gradient],
                       mix=getattr(cfg, "imag_gradient_mix", 0.0))
        # ---- critic branch: -log_prob(target) - log_prob(slow.mode), weighted by the cumulative discount.
        # Independent of the actor's backward, so it runs on the side stream beside the dynamics scan.
        R = H1N

        def _critic():
            inv = 1.0 / H1N
            up_v = ops.scale_neg(weights.view(HN)[:R], g("bh.up_v", (R,)), inv)
            lp, lp2 = g("bh.lp_v", (R,)), g("bh.lp_v2", (R,))
            dvl = g("bh.dv_logits", (R, 255))
            ops.disc_logprob_fwd(v_logits[:R], target.view(R), lp)
            ops.dot_accumulate(lp, acc[1:2], w=up_v)
            ops.disc_logprob_bwd(v_logits[:R], target.view(R), up_v, dvl)
            if cfg.critic["slow_target"]:
                ops.disc_logprob_fwd(v_logits[:R], slow.view(R), lp2)
                ops.dot_accumulate(lp2, acc[1:2], w=up_v)
                ops.disc_logprob_bwd(v_logits[:R], slow.view(R), up_v, dvl, accumulate=True)
            veng.backward(fs[:R], fd[:R], slice(0, R), dout=dvl, wgrad=True)

        E.Cuts.mark("bh.B@critic")
        side = E.SideStream(fs.device)
        side.run([_critic], chain=False)  # (what follows fills the chip: in line unless the plain second stream is on)
        E.Cuts.mark("bh.B@dyn")
        # ---- dynamics backprop: target -> reward / cont heads -> imagined states -> actions
        daction = g("bh.daction", (H, N, A))
        if not reinforce:
            dreward, dcl = g("bh.dreward", (H, N)), g("bh.dcont", (H, N))
            ops.lambda_return_bwd(dtarget, value, c_logit.view(H, N), target, dreward, dcl, gamma=cfg.discount,
                                  lam=cfg.discount_lambda)
            gs, gd = g("bh.gs", (H, N, SD)), g("bh.gd", (H, N, De))
            rows = slice(N, HN)  # step 0 is the (detached) start state: no gradient there
            g_act = None
            if custom:
                # d loss / d reward -> the objective's own graph -> gradients on the imagined stoch / deter / action
                # (reward[0] never enters a return, so dreward[0] = 0 and row block 0 receives nothing)
                g_st = g_dt = None
                if obj_out.requires_grad:  # (an objective of the detached feat alone has no gradient at all)
                    g_st, g_dt, g_act = torch.autograd.grad(
                        obj_out, [obj_leaves["stoch"], obj_leaves["deter"], obj_leaves["action"]],
                        grad_outputs=dreward.view_as(obj_out), allow_unused=True)
                for buf, gr, w in ((gs, g_st, SD), (gd, g_dt, De)):
                    if gr is None:
                        buf.view(HN, w)[rows].zero_()
                    else:
                        buf.view(HN, w)[rows].copy_(gr.reshape(HN, w)[rows])
            else:
                drl = g("bh.dr_logits", (H1N, 255))
                ops.disc_mode_bwd(r_logits[rows], dreward.view(HN)[rows], drl)
                reng.backward(fs[rows], fd[rows], rows, dout=drl, wgrad=False, dx1=gs.view(HN, SD)[rows],
                              dx2=gd.view(HN, De)[rows])
            ceng.backward(fs[rows], fd[rows], rows, dout=dcl.view(HN, 1)[rows], wgrad=False,
                          dx1=gs.view(HN, SD)[rows], dx2=gd.view(HN, De)[rows], acc_dx=True)
            Hd = dyn._hidden
            scratch = dict(dlogit=g("bh.s.dlogit", (N, SD)), dx2=g("bh.s.dx2", (N, Hd)), dx2pre=g("bh.s.dx2pre", (N, Hd)),
                           dgpre=g("bh.s.dgpre", (N, 3 * De)), dx1=g("bh.s.dx1", (N, Hd)),
                           dx1pre=g("bh.s.dx1pre", (N, Hd)))
            E.Cuts.mark("bh.C")  # the reverse rollout: a chain of dependent 1024-row launches
            for t in range(H - 1, 0, -1):
                if t < H - 1:
                    E.Cuts.mark(f"bh.C@{t}")
                b = {k: v[t - 1] for k, v in im["step"].items()}
                b.update(logit=im["logit"][t].view(N, S, D))
                # state gradients flow straight into gs/gd[t-1] (which already hold the heads' gradient);
                # step 0 is the detached start state: its slot is scratch
                rssm.img_step_bwd(gs[t], gd[t], deter[t - 1], b, scratch, gs[t - 1], gd[t - 1], daction[t - 1],
                                  accumulate_prev=t > 1, wt=wt_bwd)
            E.Cuts.mark("bh.D")
            if g_act is not None:
                # an action-conditioned objective: reward_t depends on action_t directly, including the LAST step's
                # (r_{H-1} enters the return of step H-2), so the actor's backward below covers all H steps
                daction[H - 1].zero_()
                ops.axpby(g_act.reshape(H, N, A).contiguous(), daction, 1.0, 1.0)
                R = HN
        # ---- actor backward over steps 0..H-2 (the last step's action feeds nothing that is used, unless the
        # objective reads it: R = HN above; dent / dlogp are zero on the last step)
        dmean, dstd = g("bh.dmean", (R, A)), g("bh.dstd", (R, A))
        if normal:
            ops.actor_normal_bwd(a_mean[:R], a_std[:R], dmean, dstd, eps=im["eps"].view(HN, A)[:R],
                                 action=action.view(HN, A)[:R], daction=None if reinforce else daction.view(HN, A)[:R],
                                 dent=dent.view(HN)[:R], dlogp=dlogp.view(HN)[:R] if use_logp else None,
                                 min_std=cfg.actor["min_std"], max_std=cfg.actor["max_std"], logp_of_sample=use_logp)
        else:
            if not reinforce:
                # sampled one-hot action: straight-through gradient of the sample, plus the entropy term
                ops.onehot_st_bwd(a_mean[:R], daction.view(HN, A)[:R], dmean, unimix=cfg.actor["unimix_ratio"])
            ops.onehot_ent_logp_bwd(a_mean[:R], action.view(HN, A)[:R], dent.view(HN)[:R],
                                    dlogp.view(HN)[:R] if use_logp else None, dmean,
                                    unimix=cfg.actor["unimix_ratio"], accumulate=not reinforce)
            dstd = None
        im["actor"].backward(fs[:R], fd[:R], slice(0, R), dout=dmean, dout2=dstd, wgrad=True)
        side.join()
        # ---- metrics + optimizers
        ops.dot_accumulate(ent.view(HN), acc[2:3], scale=1.0 / HN)
        metrics = {}
        # the five statistics groups of models.py:431-445 in one launch (value: the critic on feat[:-1], models.py:419)
        groups = [(value[:-1], "value", None, None), (target, "target", None, None), (reward, "imag_reward", None, None),
                  (action if normal else torch.argmax(action, dim=-1).float(), "imag_action", None, None)]
        if cfg.imag_gradient == "both":
            metrics["imag_gradient_mix"] = cfg.imag_gradient_mix
        if cfg.reward_EMA:
            scale = torch.clip(ema[1:2] - ema[0:1], min=1.0)
            groups.append((target, "normed_target", ema[0:1], scale))
            metrics["EMA_005"], metrics["EMA_095"] = ema[0], ema[1]
        metrics.update(tools.tensorstats_many(groups))
        metrics["actor_entropy"] = acc[2]
        self.ema_to_wire()
        self._last = dict(reward=reward, value=value, target=target, weights=weights, disc=disc, slow=slow)
        imag_state = {"stoch": stoch.view(H, N, S, D), "deter": deter, "logit": im["logit"].view(H, N, S, D)}
        self._pending = ((None, imag_state, action, weights.view(H, N, 1)), metrics, (acc[0], acc[1]))

    def _actor_heads(self, im):
        eng = im["actor"]
        _, out, out2 = eng._bufs(eng.total)
        return out, out2
