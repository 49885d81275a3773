"""MI355X-native `networks` module: the reference's class surface (RSSM, GRUCell, MLP, ConvEncoder,
ConvDecoder, MultiEncoder, MultiDecoder, Conv2dSamePad, ImgChLayerNorm) over libdv3hip.

Modules are parameter containers with the reference's names, shapes and state_dict keys
 plus the reference's methods.  Every method computes with the HIP kernels
(dv3hip.ops / dv3hip.engine); there is no ATen math path and no CPU path.  Two ways in:

* the fused training path -- models.WorldModel._train / models.ImagBehavior._train drive dv3hip.engine's explicit
  forward + hand-derived backward over whole updates (no autograd graph);
* the public methods below.  Without gradients (acting, logging, open-loop prediction) they launch the forward
  kernels only.  When autograd is recording and an input or a parameter requires a gradient they go through
  dv3hip.autograd's Functions -- the same kernels forward and backward -- so that callers written against the
  reference's surface with `loss.backward()` (exploration.Plan2Explore, the causal world models)
  run unchanged.

Continuous latents (dyn_discrete: 0) are not implemented: no shipped config uses them.
"""
from __future__ import annotations

import math
import re

import numpy as np
import torch
from torch import nn

import tools
from dv3hip import autograd as AG
from dv3hip import engine as E
from dv3hip import ops


def _workspace(mod: nn.Module, device) -> E.Workspace:
    ws = getattr(mod, "_ws", None)
    if ws is None or ws.device != torch.device(device):
        ws = E.Workspace(torch.device(device))
        object.__setattr__(mod, "_ws", ws)
    return ws


def _dense_ln(seq) -> E.PDenseLN:
    lin, norm = seq[0], seq[1]
    return E.PDenseLN(lin.weight, norm.weight, norm.bias)


class GRUCell(nn.Module):
    """networks.py:742-768: Linear(inp+size -> 3*size, no bias) + LayerNorm(3*size), update bias -1."""

    def __init__(self, inp_size, size, norm=True, act=torch.tanh, update_bias=-1):
        super().__init__()
        if not norm or update_bias != -1:
            raise NotImplementedError("the fused GRU kernel implements norm=True, update_bias=-1")
        self._inp_size, self._size = inp_size, size
        self.layers = nn.Sequential()
        self.layers.add_module("GRU_linear", nn.Linear(inp_size + size, 3 * size, bias=False))
        self.layers.add_module("GRU_norm", nn.LayerNorm(3 * size, eps=1e-03))

    @property
    def state_size(self):
        return self._size

    def params(self) -> E.PDenseLN:
        return E.PDenseLN(self.layers.GRU_linear.weight, self.layers.GRU_norm.weight, self.layers.GRU_norm.bias)

    def forward(self, inputs, state):
        p = self.params()
        if AG.wants_grad(inputs, state[0], p.W, p.g, p.b):
            out = AG.GRUFn.apply(inputs, state[0], p.W, p.g, p.b)
            return out, [out]
        h = state[0].contiguous()
        M = h.shape[0]
        pre = torch.empty(M, 3 * self._size, device=h.device)
        ops.gemm(inputs.contiguous(), p.W, pre, A2=h)
        out = torch.empty_like(h)
        mean, rstd = torch.empty(M, device=h.device), torch.empty(M, device=h.device)
        ops.gru_fwd(pre, p.g, p.b, h, out, mean, rstd)
        return out, [out]


class DenseLNBlock(nn.Sequential):
    """[Linear(no bias), LayerNorm(eps 1e-3), SiLU] as the reference builds its `_img_in_layers`, `_img_out_layers`
    and `_obs_out_layers` (networks.py:44-78; same child indices, so the same state_dict keys `0.weight`,
    `1.weight`, `1.bias`), callable like the nn.Sequential it is there (scm_world_model.py:138, 160, 164): one GEMM +
    one fused LayerNorm/SiLU launch, through dv3hip.autograd.TrunkFn when gradients are wanted."""

    def __init__(self, inp, hidden):
        super().__init__(nn.Linear(inp, hidden, bias=False), nn.LayerNorm(hidden, eps=1e-03), nn.SiLU())

    def forward(self, x):
        W, g, b = self[0].weight, self[1].weight, self[1].bias
        if AG.wants_grad(x, W, g, b):
            return AG.TrunkFn.apply(x, W, g, b)
        x2 = x.reshape(-1, x.shape[-1]).to(torch.float32).contiguous()
        pre = torch.empty(x2.shape[0], W.shape[0], device=x.device)
        y = torch.empty_like(pre)
        E.dense_ln_fwd(E.PDenseLN(W, g, b), x2, None, pre, None, None, y)
        return y.view(tuple(x.shape[:-1]) + (W.shape[0],))


class RSSM(nn.Module):
    def __init__(self, stoch=30, deter=200, hidden=200, rec_depth=1, discrete=False, act="SiLU", norm=True,
                 mean_act="none", std_act="softplus", min_std=0.1, unimix_ratio=0.01, initial="learned",
                 num_actions=None, embed=None, device=None):
        super().__init__()
        if not discrete:
            raise NotImplementedError("continuous latents (dyn_discrete: 0) are not implemented")
        if act != "SiLU" or not norm or rec_depth != 1 or initial != "learned":
            raise NotImplementedError("kernels implement act=SiLU, norm=True, rec_depth=1, initial=learned")
        self._stoch, self._deter, self._hidden, self._discrete = stoch, deter, hidden, discrete
        self._unimix_ratio, self._num_actions, self._embed, self._device = unimix_ratio, num_actions, embed, device
        self._min_std, self._rec_depth, self._initial = min_std, rec_depth, initial

        def block(inp):
            seq = DenseLNBlock(inp, hidden)
            seq.apply(tools.weight_init)
            return seq

        self._img_in_layers = block(stoch * discrete + num_actions)
        self._cell = GRUCell(hidden, deter, norm=norm)
        self._cell.apply(tools.weight_init)
        self._img_out_layers = block(deter)
        self._obs_out_layers = block(deter + embed)
        self._imgs_stat_layer = nn.Linear(hidden, stoch * discrete)
        self._imgs_stat_layer.apply(tools.uniform_weight_init(1.0))
        self._obs_stat_layer = nn.Linear(hidden, stoch * discrete)
        self._obs_stat_layer.apply(tools.uniform_weight_init(1.0))
        self.W = nn.Parameter(torch.zeros((1, deter), device=torch.device(device) if device else None),
                              requires_grad=True)

    # ---- engine plumbing -----------------------------------------------------------------------
    def params(self) -> E.PRSSM:
        return E.PRSSM(self.W, _dense_ln(self._img_in_layers), self._cell.params(), _dense_ln(self._img_out_layers),
                       _dense_ln(self._obs_out_layers),
                       E.PLin(self._imgs_stat_layer.weight, self._imgs_stat_layer.bias),
                       E.PLin(self._obs_stat_layer.weight, self._obs_stat_layer.bias))

    @property
    def engine(self) -> E.RSSMEngine:
        ws = _workspace(self, self.W.device)
        eng = getattr(self, "_eng", None)
        if eng is None or eng.ws is not ws:
            eng = E.RSSMEngine(self.params(), ws, stoch=self._stoch, discrete=self._discrete, deter=self._deter,
                               hidden=self._hidden, num_actions=self._num_actions, embed=self._embed,
                               unimix=self._unimix_ratio)
            object.__setattr__(self, "_eng", eng)
        eng.P = self.params()
        return eng

    def _rng(self):
        return tools.default_rng(self.W.device)

    # ---- reference API (forward only) ------------------------------------------------------------
    def _all_params(self):
        return AG.rssm_param_list(self.params())

    def initial(self, batch_size):
        S, D = self._stoch, self._discrete
        if AG.wants_grad(*self._all_params()):  # networks.py:99-123: deter = tanh(W), stoch = mode of the prior head
            deter = AG.TanhFn.apply(self.W).repeat(batch_size, 1)
            return dict(logit=torch.zeros(batch_size, S, D, device=deter.device), stoch=self.get_stoch(deter),
                        deter=deter)
        s0, d0 = self.engine.init_state_fwd()
        return dict(logit=torch.zeros(batch_size, S, D, device=s0.device),
                    stoch=s0.view(1, S, D).repeat(batch_size, 1, 1), deter=d0.repeat(batch_size, 1))

    def get_feat(self, state):
        st = state["stoch"]
        return torch.cat([st.reshape(list(st.shape[:-2]) + [self._stoch * self._discrete]), state["deter"]], -1)

    def get_dist(self, state, dtype=None):
        return tools.IndependentOneHot(tools.OneHotDist(state["logit"], unimix_ratio=self._unimix_ratio,
                                                        rng=self._rng()))

    def get_stoch(self, deter):
        p = self.params()
        if AG.wants_grad(deter, *self._all_params()):
            x = self._img_out_layers(deter)
            logit = self._suff_stats_layer("ims", x)["logit"]
            return self.get_dist({"logit": logit}).mode()
        M = deter.shape[0]
        dev = deter.device
        pre, x = torch.empty(M, self._hidden, device=dev), torch.empty(M, self._hidden, device=dev)
        E.dense_ln_fwd(p.img_out, deter.contiguous(), None, pre, None, None, x)
        logit = torch.empty(M, self._stoch, self._discrete, device=dev)
        ops.gemm(x, p.ims.W, logit.view(M, -1), bias=p.ims.b)
        return tools.OneHotDist(logit, unimix_ratio=self._unimix_ratio).mode()

    def _suff_stats_layer(self, name, x):
        """networks.py:241-250 (discrete latents): x [..., hidden] -> {"logit": [..., stoch, discrete]}."""
        lin = {"ims": self._imgs_stat_layer, "obs": self._obs_stat_layer}.get(name)
        if lin is None:
            raise NotImplementedError(name)
        lead = x.shape[:-1]
        if AG.wants_grad(x, lin.weight, lin.bias):
            out = AG.LinearFn.apply(x, lin.weight, lin.bias)
            return {"logit": out.reshape(tuple(lead) + (self._stoch, self._discrete))}
        x2 = x.to(torch.float32).reshape(-1, x.shape[-1]).contiguous()
        out = torch.empty(x2.shape[0], self._stoch * self._discrete, device=x.device)
        ops.gemm(x2, lin.weight, out, bias=lin.bias)
        return {"logit": out.reshape(tuple(lead) + (self._stoch, self._discrete))}

    def _step_bufs(self, M, dev):
        S, D, De, Hd = self._stoch, self._discrete, self._deter, self._hidden
        mk = lambda *s: torch.empty(*s, device=dev)
        return dict(x1pre=mk(M, Hd), m1=mk(M), r1=mk(M), x1=mk(M, Hd), gpre=mk(M, 3 * De), mg=mk(M), rg=mk(M),
                    deter=mk(M, De), x2pre=mk(M, Hd), m2=mk(M), r2=mk(M), x2=mk(M, Hd), logit=mk(M, S, D),
                    stoch=mk(M, S, D))

    def _img_step_grad(self, prev_state, prev_action, sample, noise):
        """img_step as a chain of autograd nodes (each one a kernel pair): img_in -> GRU -> img_out -> stats -> sample."""
        st = prev_state["stoch"]
        x = torch.cat([st.reshape(tuple(st.shape[:-2]) + (self._stoch * self._discrete,)),
                       prev_action.to(torch.float32)], -1)
        x = self._img_in_layers(x)
        deter, _ = self._cell(x, [prev_state["deter"]])
        logit = self._suff_stats_layer("ims", self._img_out_layers(deter))["logit"]
        dist = tools.OneHotDist(logit, unimix_ratio=self._unimix_ratio, rng=self._rng())
        stoch = dist.sample(noise=noise) if sample else dist.mode()
        return {"stoch": stoch, "deter": deter, "logit": logit}

    def img_step(self, prev_state, prev_action, sample=True, noise=None):
        """networks.py:208-233."""
        if AG.wants_grad(prev_state["stoch"], prev_state["deter"], prev_action, *self._all_params()):
            return self._img_step_grad(prev_state, prev_action, sample, noise)
        st = prev_state["stoch"].contiguous()
        M = st.shape[0]
        b = self._step_bufs(M, st.device)
        self.engine.img_step_fwd(st.view(M, -1), prev_state["deter"].contiguous(),
                                 prev_action.to(torch.float32).contiguous(), b, noise=noise,
                                 rng=self._rng(), sample=sample)
        self._rng().commit()
        return {"stoch": b["stoch"], "deter": b["deter"], "logit": b["logit"]}

    def obs_step(self, prev_state, prev_action, embed, is_first, sample=True, noise=None, prior=True):
        """networks.py:174-206 (branch-free reset; no host sync on is_first).

        Returns (post, prior).  Unlike the reference this does not write through `prev_action` /
        `prev_state`: the zeroed action is used internally.  prior=False (the acting step, which
        drops the prior: dreamer.py:131-134) skips the prior head and returns (post, None)."""
        p = self.params()
        B = embed.shape[0]
        dev = embed.device
        S, D, SD, De, Hd, A = self._stoch, self._discrete, self._stoch * self._discrete, self._deter, \
            self._hidden, self._num_actions
        ps = prev_state or {}
        if AG.wants_grad(embed, prev_action, ps.get("stoch"), ps.get("deter"), *self._all_params()):
            # networks.py:174-206 with the reset as a branch-free blend (m = 0: identity; m = 1: initial state)
            nz = noise or {}
            m = is_first.to(torch.float32).reshape(B, 1)
            init = self.initial(B)
            if prev_state is None:
                prev_state, prev_action = init, torch.zeros(B, A, device=dev)
            else:
                prev_action = prev_action.to(torch.float32) * (1.0 - m)
                mm = lambda v: m.reshape((B,) + (1,) * (v.dim() - 1))
                prev_state = {k: v * (1.0 - mm(v)) + init[k] * mm(v) for k, v in prev_state.items()}
            pri = self._img_step_grad(prev_state, prev_action, sample, nz.get("prior"))
            x = self._obs_out_layers(torch.cat([pri["deter"], embed.to(torch.float32)], -1))
            logit = self._suff_stats_layer("obs", x)["logit"]
            dist = tools.OneHotDist(logit, unimix_ratio=self._unimix_ratio, rng=self._rng())
            stoch = dist.sample(noise=nz.get("post")) if sample else dist.mode()
            return {"stoch": stoch, "deter": pri["deter"], "logit": logit}, (pri if prior else None)
        s0, d0 = self.engine.init_state_fwd()
        first = is_first.to(torch.float32).reshape(B).contiguous()
        sin, din, ain = torch.empty(B, SD, device=dev), torch.empty(B, De, device=dev), torch.empty(B, A, device=dev)
        if prev_state is None:
            first = torch.ones(B, device=dev)
            ops.reset_blend(None, s0.view(SD), first, sin)
            ops.reset_blend(None, d0.view(De), first, din)
            ain.zero_()
        else:
            ops.reset_blend(prev_state["stoch"].reshape(B, SD).contiguous(), s0.view(SD), first, sin)
            ops.reset_blend(prev_state["deter"].contiguous(), d0.view(De), first, din)
            ops.reset_blend(prev_action.to(torch.float32).contiguous(), None, first, ain)
        nz = noise or {}
        b = self._step_bufs(B, dev)
        self.engine.img_step_fwd(sin, din, ain, b, noise=nz.get("prior"), rng=self._rng(), sample=sample, head=prior)
        prior = {"stoch": b["stoch"], "deter": b["deter"], "logit": b["logit"]} if prior else None
        x3pre, x3 = torch.empty(B, Hd, device=dev), torch.empty(B, Hd, device=dev)
        E.dense_ln_fwd(p.obs_out, b["deter"], embed.contiguous(), x3pre, None, None, x3)
        logit = torch.empty(B, S, D, device=dev)
        ops.gemm(x3, p.obs.W, logit.view(B, SD), bias=p.obs.b)
        stoch = torch.empty(B, S, D, device=dev)
        ops.onehot_sample(logit, stoch, noise=nz.get("post"), rng=self._rng(), unimix=self._unimix_ratio,
                          mode=not sample)
        self._rng().commit()
        post = {"stoch": stoch, "deter": b["deter"], "logit": logit}
        return post, prior

    def observe(self, embed, action, is_first, state=None, noise=None):
        """networks.py:127-143: embed [B,T,E], action [B,T,A], is_first [B,T] -> (post, prior) of [B,T,...]."""
        B, T = embed.shape[0], embed.shape[1]
        dev = embed.device
        nz = noise or {}
        if state is None and AG.wants_grad(embed, *self._all_params()):
            # the whole scan as ONE autograd node over engine.RSSMEngine.observe_fwd / observe_bwd
            tmg = lambda x: x.to(torch.float32).transpose(0, 1).contiguous()
            dims = dict(stoch=self._stoch, discrete=self._discrete, deter=self._deter, hidden=self._hidden,
                        num_actions=self._num_actions, embed=self._embed, unimix=self._unimix_ratio)
            ps, pl, dt, qs, ql = AG.ObserveFn.apply(tmg(embed), tmg(action), tmg(is_first), nz.get("q_prior"),
                                                   nz.get("q_post"), self._rng(), dims, *self._all_params())
            self._rng().commit()
            bt = lambda x: x.transpose(0, 1)
            return ({"stoch": bt(ps), "deter": bt(dt), "logit": bt(pl)},
                    {"stoch": bt(qs), "deter": bt(dt), "logit": bt(ql)})
        if state is not None and AG.wants_grad(embed, *self._all_params()):
            # carried state (rare with gradients): step by step through obs_step's autograd chain
            swap = lambda x: x.transpose(0, 1)
            outs = tools.static_scan(lambda prev, a, e, f: self.obs_step(prev[0], a, e, f),
                                     (swap(action), swap(embed), swap(is_first)), (state, state))
            return tuple({k: swap(v) for k, v in o.items()} for o in outs)
        tm = lambda x: ops.transpose01(x.to(torch.float32).contiguous(),
                                       torch.empty((T, B) + tuple(x.shape[2:]), device=dev))
        state0 = None
        if state is not None:  # carried state {stoch [B,S,D], deter [B,De], logit}: step 0 does not reset
            state0 = (state["stoch"].to(torch.float32).reshape(B, -1).contiguous(),
                      state["deter"].to(torch.float32).contiguous())
        out = self.engine.observe_fwd(tm(embed), tm(action), tm(is_first), q_prior=nz.get("q_prior"),
                                      q_post=nz.get("q_post"), rng=self._rng(), state0=state0)
        self._rng().commit()
        bt = lambda x: x.transpose(0, 1).clone()
        post = {"stoch": bt(out["post_stoch"]), "deter": bt(out["deter"]), "logit": bt(out["post_logit"])}
        prior = {"stoch": bt(out["prior_stoch"]), "deter": bt(out["deter"]), "logit": bt(out["prior_logit"])}
        return post, prior

    def imagine_with_action(self, action, state, noise=None):
        """networks.py:145-152: open-loop rollout of given actions [B,T,A] from `state` {[B,...]}.
        noise (tests): Exp(1) draws [T,B,S,D] for the prior samples; default = the Philox stream."""
        assert isinstance(state, dict), state
        outs = []
        cur = state
        for t in range(action.shape[1]):
            cur = self.img_step(cur, action[:, t], noise=None if noise is None else noise[t])
            outs.append(cur)
        return {k: torch.stack([o[k] for o in outs], 1) for k in outs[0]}

    def kl_loss(self, post, prior, free, dyn_scale, rep_scale):
        """networks.py:272-290 forward values: (loss, value, dyn_loss, rep_loss), each [B,T]."""
        if AG.wants_grad(post["logit"], prior["logit"]):
            return AG.KLLossFn.apply(post["logit"], prior["logit"], free, dyn_scale, rep_scale, self._unimix_ratio)
        pl, ql = post["logit"].contiguous(), prior["logit"].contiguous()
        shape = pl.shape[:-2]
        kl = torch.empty(shape, device=pl.device)
        ops.kl_fwd(pl, ql, kl, unimix=self._unimix_ratio)
        clipped = torch.clip(kl, min=free)
        return dyn_scale * clipped + rep_scale * clipped, kl, clipped, clipped.clone()


# ---------------------------------------------------------------------------------------------
class MLP(nn.Module):
    """networks.py:588-739 for the dists the shipped configs use: normal (actor), onehot (actor),
    symlog_disc (reward/value), binary (cont), symlog_mse (proprio decoder), None (proprio encoder)."""

    def __init__(self, inp_dim, shape, layers, units, act="SiLU", norm=True, dist="normal", std=1.0, min_std=0.1,
                 max_std=1.0, absmax=None, temp=0.1, unimix_ratio=0.01, outscale=1.0, symlog_inputs=False,
                 device="cuda", name="NoName"):
        super().__init__()
        if act != "SiLU" or not norm:
            raise NotImplementedError("kernels implement act=SiLU, norm=True")
        self._shape = (shape,) if isinstance(shape, int) else shape
        if self._shape is not None and not isinstance(self._shape, dict) and len(self._shape) == 0:
            self._shape = (1,)
        self._dist, self._std, self._min_std, self._max_std = dist, std, min_std, max_std
        self._absmax, self._unimix_ratio, self._symlog_inputs, self._device = absmax, unimix_ratio, symlog_inputs, device
        self._name = name
        self.layers = nn.Sequential()
        for i in range(layers):
            self.layers.add_module(f"{name}_linear{i}", nn.Linear(inp_dim, units, bias=False))
            self.layers.add_module(f"{name}_norm{i}", nn.LayerNorm(units, eps=1e-03))
            self.layers.add_module(f"{name}_act{i}", nn.SiLU())
            inp_dim = units
        self.layers.apply(tools.weight_init)
        self._n_layers = layers
        if isinstance(self._shape, dict):
            self.mean_layer = nn.ModuleDict({k: nn.Linear(inp_dim, int(np.prod(s))) for k, s in self._shape.items()})
            self.mean_layer.apply(tools.uniform_weight_init(outscale))
            if self._std == "learned":
                raise NotImplementedError("learned std with dict outputs")
        elif self._shape is not None:
            self.mean_layer = nn.Linear(inp_dim, int(np.prod(self._shape)))
            self.mean_layer.apply(tools.uniform_weight_init(outscale))
            if self._std == "learned":
                assert dist in ("tanh_normal", "normal", "trunc_normal", "huber"), dist
                self.std_layer = nn.Linear(units, int(np.prod(self._shape)))
                self.std_layer.apply(tools.uniform_weight_init(outscale))

    def trunk_params(self):
        nm = self._name
        return [E.PDenseLN(getattr(self.layers, f"{nm}_linear{i}").weight, getattr(self.layers, f"{nm}_norm{i}").weight,
                           getattr(self.layers, f"{nm}_norm{i}").bias) for i in range(self._n_layers)]

    def params(self, key=None) -> E.PMLP:
        out = out2 = None
        if isinstance(self._shape, dict):
            if key is not None:
                out = E.PLin(self.mean_layer[key].weight, self.mean_layer[key].bias)
        elif self._shape is not None:
            out = E.PLin(self.mean_layer.weight, self.mean_layer.bias)
            if hasattr(self, "std_layer"):
                out2 = E.PLin(self.std_layer.weight, self.std_layer.bias)
        return E.PMLP(self.trunk_params(), out, out2)

    def engine_for(self, tag="", key=None) -> E.MLPEngine:
        dev = next(self.parameters()).device
        ws = _workspace(self, dev)
        return E.MLPEngine(f"{self._name}{tag}", self.params(key), ws)

    def make_dist(self, mean, std=None):
        d = self._dist
        if d == "normal":
            if std is None:  # fixed std (networks.py:614, 683-684): the constant goes through the same squashing
                std = torch.full_like(mean, float(self._std))
            return tools.ContDist(mean, std, self._min_std, self._max_std, absmax=self._absmax)
        if d == "onehot":
            return tools.OneHotDist(mean, unimix_ratio=self._unimix_ratio)
        if d == "symlog_disc":
            return tools.DiscDist(logits=mean, device=self._device)
        if d == "binary":
            return tools.Bernoulli(mean)
        if d == "symlog_mse":
            return tools.SymlogDist(mean)
        raise NotImplementedError(d)

    def _forward_grad(self, features):
        """networks.py:657-681 as autograd nodes: the trunk is one node (engine.MLPEngine forward / backward), every
        head Linear another; the returned distributions differentiate through dv3hip.autograd's head Functions."""
        x = features.to(torch.float32)
        if self._symlog_inputs:
            x = tools.symlog(x)  # inputs are data for every caller: no gradient through the symlog
        flat = []
        for P in self.trunk_params():
            flat += [P.W, P.g, P.b]
        h = AG.TrunkFn.apply(x, *flat) if flat else x
        if self._shape is None:
            return h
        if isinstance(self._shape, dict):
            return {k: self.make_dist(AG.LinearFn.apply(h, self.mean_layer[k].weight, self.mean_layer[k].bias)
                                      .reshape(tuple(h.shape[:-1]) + tuple(shp)))
                    for k, shp in self._shape.items()}
        mean = AG.LinearFn.apply(h, self.mean_layer.weight, self.mean_layer.bias)
        std = AG.LinearFn.apply(h, self.std_layer.weight, self.std_layer.bias) if hasattr(self, "std_layer") else None
        return self.make_dist(mean, std)

    def forward(self, features, dtype=None):
        if AG.wants_grad(features, *self.parameters()):
            return self._forward_grad(features)
        lead = features.shape[:-1]
        x = features.reshape(-1, features.shape[-1]).to(torch.float32).contiguous()
        if self._symlog_inputs:
            x = tools.symlog(x)
        if isinstance(self._shape, dict):
            eng = self.engine_for(".pub")
            h, _, _ = eng.forward(x)
            dists = {}
            for k, shp in self._shape.items():
                lin = self.mean_layer[k]
                o = torch.empty(x.shape[0], lin.weight.shape[0], device=x.device)
                ops.gemm(h, lin.weight, o, bias=lin.bias)
                dists[k] = self.make_dist(o.reshape(tuple(lead) + tuple(shp)))
            return dists
        eng = self.engine_for(".pub")
        h, o, o2 = eng.forward(x)
        if self._shape is None:
            return h.reshape(tuple(lead) + (h.shape[-1],)).clone()
        mean = o.reshape(tuple(lead) + (o.shape[-1],)).clone()
        std = o2.reshape(tuple(lead) + (o2.shape[-1],)).clone() if o2 is not None else None
        return self.make_dist(mean, std)


# ---------------------------------------------------------------------------------------------
class Conv2dSamePad(nn.Conv2d):
    """networks.py:771-798.  Parameter container for the k4 s2 'same' conv; forward = NHWC implicit GEMM."""

    def forward(self, x_nhwc):
        Co, Ci = self.weight.shape[0], self.weight.shape[1]
        wp = torch.empty(Co, 16 * Ci, device=x_nhwc.device)
        ops.pack_conv_weight(self.weight, wp, transposed=False)
        N, H, W, _ = x_nhwc.shape
        y = torch.empty(N, H // 2, W // 2, Co, device=x_nhwc.device)
        return ops.conv_s2_fwd(x_nhwc.contiguous(), wp, y, Ci=Ci, Co=Co)


class ImgChLayerNorm(nn.Module):
    """networks.py:801-810 (LayerNorm over channels per pixel; our activations are NHWC already)."""

    def __init__(self, ch, eps=1e-03):
        super().__init__()
        self.norm = nn.LayerNorm(ch, eps=eps)


class ConvEncoder(nn.Module):
    def __init__(self, input_shape, depth=32, act="SiLU", norm=True, kernel_size=4, minres=4):
        super().__init__()
        if act != "SiLU" or not norm or kernel_size != 4:
            raise NotImplementedError("kernels implement act=SiLU, norm=True, kernel_size=4")
        h, w, input_ch = input_shape
        stages = int(np.log2(h) - np.log2(minres))
        in_dim, out_dim = input_ch, depth
        mods = []
        for _ in range(stages):
            mods += [Conv2dSamePad(in_channels=in_dim, out_channels=out_dim, kernel_size=4, stride=2, bias=False),
                     ImgChLayerNorm(out_dim), nn.SiLU()]
            in_dim, out_dim = out_dim, out_dim * 2
            h, w = h // 2, w // 2
        self.outdim = out_dim // 2 * h * w
        self._size, self._stages = input_shape[0], stages
        self.layers = nn.Sequential(*mods)
        self.layers.apply(tools.weight_init)

    def conv_params(self):
        return [E.PConvLayer(self.layers[3 * i].weight, self.layers[3 * i + 1].norm.weight,
                             self.layers[3 * i + 1].norm.bias) for i in range(self._stages)]

    @property
    def engine(self) -> E.ConvEncoderEngine:
        ws = _workspace(self, self.layers[0].weight.device)
        return E.ConvEncoderEngine(self.conv_params(), ws, size=self._size)

    def forward(self, obs):
        """obs f32 in [0,1], [..., H, W, C] (networks.py:486-496) -> [..., outdim]."""
        lead = obs.shape[:-3]
        x = (obs.to(torch.float32) - 0.5).reshape((-1,) + tuple(obs.shape[-3:])).contiguous()
        if AG.wants_grad(*self.parameters()):
            flat = []
            for L in self.conv_params():
                flat += [L.W, L.g, L.b]
            emb = AG.ConvEncoderFn.apply(x, self._size, *flat)
            return emb.reshape(tuple(lead) + (emb.shape[-1],))
        emb = self.engine.forward(x_f32=x, keep=False)
        return emb.reshape(tuple(lead) + (emb.shape[-1],)).clone()


class ConvDecoder(nn.Module):
    def __init__(self, feat_size, shape=(3, 64, 64), depth=32, act="SiLU", norm=True, kernel_size=4, minres=4,
                 outscale=1.0, cnn_sigmoid=False):
        super().__init__()
        if act != "SiLU" or not norm or kernel_size != 4 or cnn_sigmoid:
            raise NotImplementedError("kernels implement act=SiLU, norm=True, kernel_size=4, cnn_sigmoid=False")
        self._shape, self._minres = shape, minres
        layer_num = int(np.log2(shape[1]) - np.log2(minres))
        out_ch = minres ** 2 * depth * 2 ** (layer_num - 1)
        self._embed_size = out_ch
        self._linear_layer = nn.Linear(feat_size, out_ch)
        self._linear_layer.apply(tools.uniform_weight_init(outscale))
        in_dim = out_ch // (minres ** 2)
        mods = []
        for i in range(layer_num):
            last = i == layer_num - 1
            out_dim = shape[0] if last else in_dim // 2
            mods.append(nn.ConvTranspose2d(in_dim, out_dim, 4, 2, padding=(1, 1), output_padding=(0, 0), bias=last))
            if not last:
                mods += [ImgChLayerNorm(out_dim), nn.SiLU()]
            in_dim = out_dim
        for m in mods[:-1]:
            m.apply(tools.weight_init)
        mods[-1].apply(tools.uniform_weight_init(outscale))
        self.layers = nn.Sequential(*mods)
        self._layer_num = layer_num

    def conv_params(self):
        out = []
        for i in range(self._layer_num - 1):
            out.append(E.PConvLayer(self.layers[3 * i].weight, self.layers[3 * i + 1].norm.weight,
                                    self.layers[3 * i + 1].norm.bias))
        lastm = self.layers[3 * (self._layer_num - 1)]
        out.append(E.PConvLayer(lastm.weight, None, None, lastm.bias))
        return out

    @property
    def engine(self) -> E.ConvDecoderEngine:
        ws = _workspace(self, self._linear_layer.weight.device)
        return E.ConvDecoderEngine(E.PLin(self._linear_layer.weight, self._linear_layer.bias), self.conv_params(), ws,
                                   minres=self._minres)

    def forward(self, features, dtype=None):
        """networks.py:568-585: features [..., F] -> mean image [..., H, W, C] (+0.5)."""
        lead = features.shape[:-1]
        if AG.wants_grad(features, *self.parameters()):
            L = self.conv_params()
            flat = [self._linear_layer.weight, self._linear_layer.bias]
            for l in L[:-1]:
                flat += [l.W, l.g, l.b]
            flat += [L[-1].W, L[-1].bias]
            rec = AG.ConvDecoderFn.apply(features.reshape(-1, features.shape[-1]), self._minres, *flat)
            return rec.reshape(tuple(lead) + tuple(rec.shape[1:]))
        x = features.reshape(-1, features.shape[-1]).to(torch.float32).contiguous()
        rec = self.engine.forward(x, None)
        return rec.reshape(tuple(lead) + tuple(rec.shape[1:])).clone()


class MultiEncoder(nn.Module):
    """networks.py:293-357: routes obs keys to the CNN / MLP encoders by regex."""

    def __init__(self, shapes, mlp_keys, cnn_keys, act, norm, cnn_depth, kernel_size, minres, mlp_layers, mlp_units,
                 symlog_inputs):
        super().__init__()
        excluded = ("is_first", "is_last", "is_terminal", "reward")
        shapes = {k: v for k, v in shapes.items() if k not in excluded and not k.startswith("log_")}
        self.cnn_shapes = {k: v for k, v in shapes.items() if len(v) == 3 and re.match(cnn_keys, k)}
        self.mlp_shapes = {k: v for k, v in shapes.items() if len(v) in (1, 2) and re.match(mlp_keys, k)}
        self.outdim = 0
        if self.cnn_shapes:
            input_ch = sum(v[-1] for v in self.cnn_shapes.values())
            input_shape = tuple(self.cnn_shapes.values())[0][:2] + (input_ch,)
            self._cnn = ConvEncoder(input_shape, cnn_depth, act, norm, kernel_size, minres)
            self.outdim += self._cnn.outdim
        if self.mlp_shapes:
            input_size = sum(sum(v) for v in self.mlp_shapes.values())
            self._mlp = MLP(input_size, None, mlp_layers, mlp_units, act, norm, symlog_inputs=symlog_inputs,
                            name="Encoder")
            self.outdim += mlp_units

    def forward(self, obs):
        outs = []
        if self.cnn_shapes:
            outs.append(self._cnn(torch.cat([obs[k] for k in self.cnn_shapes], -1)))
        if self.mlp_shapes:
            outs.append(self._mlp(torch.cat([obs[k] for k in self.mlp_shapes], -1)))
        return torch.cat(outs, -1)


class MultiDecoder(nn.Module):
    """networks.py:360-445."""

    def __init__(self, feat_size, shapes, mlp_keys, cnn_keys, act, norm, cnn_depth, kernel_size, minres, mlp_layers,
                 mlp_units, cnn_sigmoid, image_dist, vector_dist, outscale):
        super().__init__()
        excluded = ("is_first", "is_last", "is_terminal")
        shapes = {k: v for k, v in shapes.items() if k not in excluded}
        self.cnn_shapes = {k: v for k, v in shapes.items() if len(v) == 3 and re.match(cnn_keys, k)}
        self.mlp_shapes = {k: v for k, v in shapes.items() if len(v) in (1, 2) and re.match(mlp_keys, k)}
        if self.cnn_shapes:
            some = list(self.cnn_shapes.values())[0]
            shape = (sum(x[-1] for x in self.cnn_shapes.values()),) + some[:-1]
            self._cnn = ConvDecoder(feat_size, shape, cnn_depth, act, norm, kernel_size, minres, outscale=outscale,
                                    cnn_sigmoid=cnn_sigmoid)
        if self.mlp_shapes:
            self._mlp = MLP(feat_size, self.mlp_shapes, mlp_layers, mlp_units, act, norm, vector_dist,
                            outscale=outscale, name="Decoder")
        if image_dist != "mse":
            raise NotImplementedError(image_dist)
        self._image_dist = image_dist

    def forward(self, features):
        dists = {}
        if self.cnn_shapes:
            out = self._cnn(features)
            sizes = [v[-1] for v in self.cnn_shapes.values()]
            for key, o in zip(self.cnn_shapes.keys(), torch.split(out, sizes, -1)):
                dists[key] = tools.MSEDist(o)
        if self.mlp_shapes:
            dists.update(self._mlp(features))
        return dists
