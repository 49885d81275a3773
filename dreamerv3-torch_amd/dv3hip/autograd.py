"""torch.autograd bridge over the libdv3hip kernel pairs.

The hot path (models.WorldModel._train / ImagBehavior._train) runs an explicit, hand-derived backward and builds no
autograd graph.  Callers OUTSIDE that path -- the reference's exploration.Plan2Explore (exploration.py:40-135), its
causal world models (scm_world_model.py:129-165, 500-560; causal_VAE.py:1045-1120), anything that writes
`loss = f(head(feat).log_prob(x)); opt(loss, params)` against the public classes -- need gradients through the public
methods.  This module gives every public method a `torch.autograd.Function` whose forward AND backward are the same HIP
kernels the hot path uses (dv3hip.engine / dv3hip.ops): no ATen math re-implementation of a layer, no CPU path.

How: the engines accumulate parameter gradients into `p.grad` of the tensors in their parameter containers.  A
Function call builds a private engine over *shadow* parameters -- detached aliases of the caller's parameters that
carry a fresh zero `.grad` -- on a private Workspace (the activations must survive until backward, and the same module
may be called several times before it), runs the engine's forward, and in backward runs the engine's backward and
hands the shadows' `.grad` to autograd, which accumulates them into the real `.grad` (for parameters of a
tools.Optimizer those are views of its flat gradient bucket, so `Optimizer.__call__(loss, params)` -- tools.py:760-776
-- clips and steps them like the hot path's).

networks.py / tools.py / models.py dispatch here only when gradients are wanted (`wants_grad`); acting, logging and
the fused training path never come through this module.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import engine as E
from . import ops

F32 = torch.float32


def wants_grad(*tensors) -> bool:
    """True when autograd is recording and any of the tensors (inputs or parameters) requires a gradient."""
    if not torch.is_grad_enabled():
        return False
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.requires_grad:
            return True
    return False


def _shadow(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Detached alias of a parameter with a fresh zero .grad for the engine's backward to accumulate into."""
    if p is None:
        return None
    s = p.detach()
    s.grad = torch.zeros_like(s, memory_format=torch.contiguous_format)
    return s


def _c(t: Optional[torch.Tensor], like: Optional[torch.Tensor] = None, shape=None) -> torch.Tensor:
    """Upstream gradient as a fresh contiguous fp32 buffer the kernels may overwrite (zeros when autograd passed None)."""
    if t is None:
        return torch.zeros(shape if shape is not None else like.shape, dtype=F32, device=like.device)
    return t.to(F32).contiguous().clone()


def _rows(x: torch.Tensor) -> torch.Tensor:
    return x.reshape(-1, x.shape[-1]).to(F32).contiguous()


# ---------------------------------------------------------------------------------------------
# Linear / [Linear -> LayerNorm -> SiLU] stacks / GRU cell
# ---------------------------------------------------------------------------------------------
class LinearFn(Function):
    """y = x W^T + b (nn.Linear; networks.py:241-250 `_suff_stats_layer`, every head's mean / std layer)."""

    @staticmethod
    def forward(ctx, x, W, b):
        x2 = _rows(x)
        y = torch.empty(x2.shape[0], W.shape[0], device=x.device, dtype=F32)
        ops.gemm(x2, W.detach(), y, bias=None if b is None else b.detach())
        ctx.save_for_backward(x2, W)
        ctx.has_bias = b is not None
        ctx.lead = tuple(x.shape[:-1])
        return y.view(ctx.lead + (W.shape[0],))

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        dy2 = _rows(dy)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            ops.gemm(dy2, W.detach(), dx, transB=False)
            dx = dx.view(ctx.lead + (x2.shape[1],))
        if ctx.needs_input_grad[1]:
            dW = torch.zeros_like(W, memory_format=torch.contiguous_format)
            ops.gemm(dy2, x2, dW, transA=True, transB=False, accumulate=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(W.shape[0], device=W.device, dtype=F32)
            ops.colsum(dy2, db)
        return dx, dW, db


class TrunkFn(Function):
    """n x [Linear(no bias) -> LayerNorm(eps 1e-3) -> SiLU] (networks.MLP.layers, networks.py:624-635; the RSSM's
    `_img_in_layers` / `_img_out_layers` / `_obs_out_layers`, networks.py:44-78) through engine.MLPEngine.
    params = (W0, g0, b0, W1, g1, b1, ...)."""

    @staticmethod
    def forward(ctx, x, *params):
        n = len(params) // 3
        x2 = _rows(x)
        layers = [E.PDenseLN(_shadow(params[3 * i]), _shadow(params[3 * i + 1]), _shadow(params[3 * i + 2]))
                  for i in range(n)]
        eng = E.MLPEngine("ag", E.PMLP(layers), E.Workspace(x.device))
        h, _, _ = eng.forward(x2)
        ctx.eng, ctx.x2, ctx.layers = eng, x2, layers
        ctx.lead = tuple(x.shape[:-1])
        return h.view(ctx.lead + (h.shape[-1],))

    @staticmethod
    @once_differentiable
    def backward(ctx, dh):
        eng, x2 = ctx.eng, ctx.x2
        R = x2.shape[0]
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        eng.backward(x2, None, slice(0, R), dh=_c(dh.reshape(R, -1)), wgrad=any(ctx.needs_input_grad[1:]), dx1=dx)
        grads: List[Optional[torch.Tensor]] = []
        for i, L in enumerate(ctx.layers):
            for j, t in enumerate((L.W, L.g, L.b)):
                grads.append(t.grad if ctx.needs_input_grad[1 + 3 * i + j] else None)
        ctx.eng = None
        return (None if dx is None else dx.view(ctx.lead + (x2.shape[1],)),) + tuple(grads)


class GRUFn(Function):
    """networks.GRUCell.forward (networks.py:760-768): LN(Linear(cat[x, h])) -> gates -> h'."""

    @staticmethod
    def forward(ctx, x, h, W, g, b):
        x2, h2 = _rows(x), _rows(h)
        M, De = h2.shape
        pre = torch.empty(M, 3 * De, device=x.device, dtype=F32)
        ops.gemm(x2, W.detach(), pre, A2=h2)
        out = torch.empty_like(h2)
        mean, rstd = torch.empty(M, device=x.device), torch.empty(M, device=x.device)
        ops.gru_fwd(pre, g.detach().contiguous(), b.detach().contiguous(), h2, out, mean, rstd)
        ctx.save_for_backward(x2, h2, W, g, b, pre, mean, rstd)
        return out.view(h.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x2, h2, W, g, b, pre, mean, rstd = ctx.saved_tensors
        M, De = h2.shape
        Hd = x2.shape[1]
        wg = any(ctx.needs_input_grad[2:])
        dpre = torch.empty_like(pre)
        dh = torch.empty_like(h2)
        dg = torch.zeros_like(g) if wg else None
        db = torch.zeros_like(b) if wg else None
        ops.gru_bwd(_c(dout.reshape(M, De)), pre, g.detach().contiguous(), b.detach().contiguous(), h2, mean, rstd, dpre,
                    dh, dg, db)
        Wd = W.detach()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            ops.gemm(dpre, Wd[:, :Hd], dx, transB=False)
        if ctx.needs_input_grad[1]:
            ops.gemm(dpre, Wd[:, Hd:], dh, transB=False, accumulate=True)
        dW = None
        if ctx.needs_input_grad[2]:
            Ws = _shadow(W)
            E.lin_wgrad(Ws, dpre, x2, h2)
            dW = Ws.grad
        return (dx, dh if ctx.needs_input_grad[1] else None, dW, dg if ctx.needs_input_grad[3] else None,
                db if ctx.needs_input_grad[4] else None)


# ---------------------------------------------------------------------------------------------
# categorical latents
# ---------------------------------------------------------------------------------------------
class OneHotSampleFn(Function):
    """tools.OneHotDist.sample / mode (tools.py:444-460): an exact one-hot forward, the straight-through gradient
    (`+ probs - probs.detach()`, resp. `+ logits - logits.detach()` for the mode) backward."""

    @staticmethod
    def forward(ctx, logit, noise, rng, unimix, mode):
        lg = logit.detach().to(F32).contiguous()
        out = torch.empty_like(lg)
        ops.onehot_sample(lg, out, noise=None if noise is None else noise.to(F32).contiguous(), rng=rng,
                          unimix=unimix, mode=mode)
        ctx.save_for_backward(lg)
        ctx.unimix, ctx.mode = unimix, mode
        ctx.mark_non_differentiable()
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (lg,) = ctx.saved_tensors
        dl = torch.empty_like(lg)
        ops.onehot_st_bwd(lg, _c(dout.reshape(lg.shape)), dl, unimix=ctx.unimix, mode=ctx.mode)
        return dl, None, None, None, None


class OneHotEntLogpFn(Function):
    """tools.OneHotDist.entropy / log_prob per categorical group (tools.py:436-442 probabilities)."""

    @staticmethod
    def forward(ctx, logit, x, unimix, want_ent):
        lg = logit.detach().to(F32).contiguous()
        out = torch.empty(lg.shape[:-1], device=lg.device, dtype=F32)
        xs = None if x is None else x.detach().to(F32).contiguous()
        if want_ent:
            ops.onehot_ent_logp_fwd(lg, None, out, None, unimix=unimix)
        else:
            ops.onehot_ent_logp_fwd(lg, xs, None, out, unimix=unimix)
        ctx.save_for_backward(lg, xs if xs is not None else lg)
        ctx.unimix, ctx.want_ent = unimix, want_ent
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        lg, xs = ctx.saved_tensors
        dl = torch.empty_like(lg)
        up = _c(dout.reshape(lg.shape[:-1]))
        if ctx.want_ent:
            ops.onehot_ent_logp_bwd(lg, None, up, None, dl, unimix=ctx.unimix)
        else:
            ops.onehot_ent_logp_bwd(lg, xs, None, up, dl, unimix=ctx.unimix)
        return dl, None, None, None


class KLLossFn(Function):
    """networks.RSSM.kl_loss (networks.py:272-290) -> loss = dyn_scale * max(dyn, free) + rep_scale * max(rep, free)
    per row; gradient through `loss` only (value / dyn / rep are logged, never optimised, by every caller)."""

    @staticmethod
    def forward(ctx, post_logit, prior_logit, free, dyn_scale, rep_scale, unimix):
        pl, ql = post_logit.detach().to(F32).contiguous(), prior_logit.detach().to(F32).contiguous()
        kl = torch.empty(pl.shape[:-2], device=pl.device, dtype=F32)
        ops.kl_fwd(pl, ql, kl, unimix=unimix)
        ctx.save_for_backward(pl, ql, kl)
        ctx.cfg = (float(free), float(dyn_scale), float(rep_scale), float(unimix))
        clipped = torch.clip(kl, min=free)
        return (dyn_scale + rep_scale) * clipped, kl.clone(), clipped, clipped.clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss, dvalue, ddyn, drep):
        pl, ql, kl = ctx.saved_tensors
        free, dyn_scale, rep_scale, unimix = ctx.cfg
        dp, dq = torch.empty_like(pl), torch.empty_like(ql)
        ops.kl_bwd(pl, ql, kl, dp, dq, unimix=unimix, free=free, dyn_scale=dyn_scale, rep_scale=rep_scale, upstream=1.0)
        up = dloss.to(F32).reshape(kl.shape + (1, 1))
        return dp * up, dq * up, None, None, None, None


# ---------------------------------------------------------------------------------------------
# RSSM.observe: the whole scan as one node (engine.RSSMEngine.observe_fwd / observe_bwd)
# ---------------------------------------------------------------------------------------------
def rssm_param_list(P: E.PRSSM) -> List[torch.Tensor]:
    return [P.W0, P.img_in.W, P.img_in.g, P.img_in.b, P.gru.W, P.gru.g, P.gru.b, P.img_out.W, P.img_out.g,
            P.img_out.b, P.obs_out.W, P.obs_out.g, P.obs_out.b, P.ims.W, P.ims.b, P.obs.W, P.obs.b]


def _rssm_shadow(params: Sequence[torch.Tensor]) -> E.PRSSM:
    s = [_shadow(p) for p in params]
    return E.PRSSM(s[0], E.PDenseLN(*s[1:4]), E.PDenseLN(*s[4:7]), E.PDenseLN(*s[7:10]), E.PDenseLN(*s[10:13]),
                   E.PLin(*s[13:15]), E.PLin(*s[15:17]))


class ObserveFn(Function):
    """networks.RSSM.observe (networks.py:127-143) on time-major inputs.  Outputs (post_stoch, post_logit, deter,
    prior_stoch, prior_logit), each [T,B,...]; gradients flow to `embed` and the 17 RSSM parameters (the action
    and the reset flags are data)."""

    @staticmethod
    def forward(ctx, embed_tm, action_tm, first_tm, q_prior, q_post, rng, dims, *params):
        P = _rssm_shadow(params)
        eng = E.RSSMEngine(P, E.Workspace(embed_tm.device), **dims)
        emb = embed_tm.detach().to(F32).contiguous()
        out = eng.observe_fwd(emb, action_tm.detach().to(F32).contiguous(), first_tm.detach().to(F32).contiguous(),
                              q_prior=q_prior, q_post=q_post, rng=rng)
        ctx.eng, ctx.P, ctx.out = eng, P, out
        ctx.set_materialize_grads(False)
        return (out["post_stoch"].clone(), out["post_logit"].clone(), out["deter"].clone(), out["prior_stoch"].clone(),
                out["prior_logit"].clone())

    @staticmethod
    @once_differentiable
    def backward(ctx, d_ps, d_pl, d_dt, d_qs, d_ql):
        eng, P, out = ctx.eng, ctx.P, ctx.out
        T, B, S, D, De = eng.T, eng.B, eng.S, eng.D, eng.De
        dev = out["deter"].device
        dpl = _c(d_pl, shape=(T, B, S, D), like=out["deter"])
        dql = _c(d_ql, shape=(T, B, S, D), like=out["deter"])
        if d_qs is not None:  # the prior sample's straight-through path into its logits
            ops.onehot_st_bwd(out["prior_logit"], _c(d_qs.reshape(T, B, S, D)), dql, unimix=eng.unimix, accumulate=True)
        gs = _c(None if d_ps is None else d_ps.reshape(T, B, S * D), shape=(T, B, S * D), like=out["deter"])
        gd = _c(d_dt, shape=(T, B, De), like=out["deter"])
        dembed = torch.empty(T, B, eng.E, device=dev, dtype=F32)
        side = eng.observe_bwd(dpl, dql, gs, gd, dembed)
        side.join()
        grads = [p.grad if ctx.needs_input_grad[7 + i] else None for i, p in enumerate(rssm_param_list(P))]
        ctx.eng = ctx.out = None
        return (dembed if ctx.needs_input_grad[0] else None, None, None, None, None, None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------
# conv stacks
# ---------------------------------------------------------------------------------------------
class ConvEncoderFn(Function):
    """networks.ConvEncoder.forward (networks.py:486-496) on x = image - 0.5 [N,H,W,C]; params = (W, g, b) per layer.
    The image itself gets no gradient (it is data for every caller)."""

    @staticmethod
    def forward(ctx, x, size, *params):
        layers = [E.PConvLayer(_shadow(params[3 * i]), _shadow(params[3 * i + 1]), _shadow(params[3 * i + 2]))
                  for i in range(len(params) // 3)]
        eng = E.ConvEncoderEngine(layers, E.Workspace(x.device), size=size)
        emb = eng.forward(x_f32=x.detach().to(F32).contiguous(), keep=True)
        ctx.eng, ctx.layers = eng, layers
        return emb.clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, demb):
        ctx.eng.backward(_c(demb))
        grads = []
        for i, L in enumerate(ctx.layers):
            for j, t in enumerate((L.W, L.g, L.b)):
                grads.append(t.grad if ctx.needs_input_grad[2 + 3 * i + j] else None)
        ctx.eng = None
        return (None, None) + tuple(grads)


class ConvDecoderFn(Function):
    """networks.ConvDecoder.forward (networks.py:568-585): feat [R,F] -> mean image [R,64,64,3] (+0.5).
    params = (lin.W, lin.b, W0, g0, b0, ..., W_last, bias_last)."""

    @staticmethod
    def forward(ctx, feat, minres, *params):
        lin = E.PLin(_shadow(params[0]), _shadow(params[1]))
        rest = params[2:]
        n_ln = (len(rest) - 2) // 3
        layers = [E.PConvLayer(_shadow(rest[3 * i]), _shadow(rest[3 * i + 1]), _shadow(rest[3 * i + 2]))
                  for i in range(n_ln)]
        layers.append(E.PConvLayer(_shadow(rest[-2]), None, None, _shadow(rest[-1])))
        eng = E.ConvDecoderEngine(lin, layers, E.Workspace(feat.device), minres=minres)
        x = _rows(feat)
        F_ = x.shape[1]
        k1 = max(32, (F_ // 2) // 32 * 32) if F_ >= 64 else F_ // 2  # two K segments, as get_feat's [stoch | deter]
        x1, x2 = x[:, :k1].contiguous(), x[:, k1:].contiguous()
        rec = eng.forward(x1, x2)
        ctx.eng, ctx.lin, ctx.layers, ctx.k1, ctx.F = eng, lin, layers, k1, F_
        return rec.clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, drec):
        R = drec.shape[0]
        dx1 = torch.empty(R, ctx.k1, device=drec.device, dtype=F32)
        dx2 = torch.empty(R, ctx.F - ctx.k1, device=drec.device, dtype=F32)
        ctx.eng.backward(_c(drec), dx1, dx2)
        flat = [ctx.lin.W, ctx.lin.b]
        for L in ctx.layers[:-1]:
            flat += [L.W, L.g, L.b]
        flat += [ctx.layers[-1].W, ctx.layers[-1].bias]
        grads = [t.grad if ctx.needs_input_grad[2 + i] else None for i, t in enumerate(flat)]
        ctx.eng = None
        return (torch.cat([dx1, dx2], 1) if ctx.needs_input_grad[0] else None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------
# distribution heads
# ---------------------------------------------------------------------------------------------
class DiscModeFn(Function):
    """tools.DiscDist.mean / mode (tools.py:475-486): symexp(sum softmax(l) * buckets)."""

    @staticmethod
    def forward(ctx, logits):
        lg = logits.detach().to(F32).contiguous()
        out = torch.empty(lg.shape[:-1] + (1,), device=lg.device, dtype=F32)
        ops.disc_mode_fwd(lg, out)
        ctx.save_for_backward(lg)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (lg,) = ctx.saved_tensors
        dl = torch.empty_like(lg)
        ops.disc_mode_bwd(lg, _c(dout.reshape(lg.shape[:-1])), dl)
        return dl


class DiscLogProbFn(Function):
    """tools.DiscDist.log_prob (tools.py:488-517): two-hot cross entropy against symlog(x)."""

    @staticmethod
    def forward(ctx, logits, x):
        lg = logits.detach().to(F32).contiguous()
        xs = x.detach().to(F32).contiguous()
        out = torch.empty(lg.shape[:-1], device=lg.device, dtype=F32)
        ops.disc_logprob_fwd(lg, xs, out)
        ctx.save_for_backward(lg, xs)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        lg, xs = ctx.saved_tensors
        dl = torch.empty_like(lg)
        ops.disc_logprob_bwd(lg, xs, _c(dout.reshape(lg.shape[:-1])), dl)
        return dl, None


class BernoulliLogProbFn(Function):
    """tools.Bernoulli.log_prob (tools.py:620-628) elementwise on logits."""

    @staticmethod
    def forward(ctx, logit, x):
        lg = logit.detach().to(F32).contiguous()
        xs = x.detach().to(F32).contiguous().reshape(lg.shape)
        out = torch.empty_like(lg)
        ops.bernoulli_logprob_fwd(lg, xs, out)
        ctx.save_for_backward(lg, xs)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        lg, xs = ctx.saved_tensors
        dl = torch.empty_like(lg)
        ops.bernoulli_logprob_bwd(lg, xs, _c(dout.reshape(lg.shape)), dl)
        return dl, None


class SymlogMSEFn(Function):
    """tools.SymlogDist.log_prob (tools.py:556-572, mse / sum): -sum (mode - symlog(x))^2 with the 1e-8 cut."""

    @staticmethod
    def forward(ctx, mode, x):
        m = mode.detach().to(F32).contiguous()
        xs = x.detach().to(F32).contiguous()
        loss = torch.empty(m.shape[:-1], device=m.device, dtype=F32)
        dmode = torch.empty_like(m)
        ops.symlog_mse(m, xs, loss, dmode, upstream=1.0)
        ctx.save_for_backward(dmode)
        return -loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (dmode,) = ctx.saved_tensors
        return dmode * (-dout.to(F32)).unsqueeze(-1), None


class NormalFn(Function):
    """The continuous actor's Normal(tanh(mean), (max-min) sigmoid(std + 2) + min) (networks.py:693-700) under
    tools.ContDist (tools.py:575-601).  kind: "sample" (rsample from eps, absmax 1 rescale detached), "entropy",
    "logp" (of a given action, treated as a constant)."""

    @staticmethod
    def forward(ctx, mean_raw, std_raw, aux, kind, min_std, max_std):
        mr, sr = mean_raw.detach().to(F32).contiguous(), std_raw.detach().to(F32).contiguous()
        ax = None if aux is None else aux.detach().to(F32).contiguous()
        lead = mr.shape[:-1]
        if kind == "sample":
            out = torch.empty_like(mr)
            ops.actor_normal_fwd(mr, sr, ax, out, None, min_std=min_std, max_std=max_std)
        elif kind == "entropy":
            out = torch.empty(lead, device=mr.device, dtype=F32)
            ops.actor_normal_fwd(mr, sr, None, None, out, min_std=min_std, max_std=max_std)
        else:
            out = torch.empty(lead, device=mr.device, dtype=F32)
            ops.actor_normal_logp(mr, sr, ax, out, min_std=min_std, max_std=max_std)
        ctx.save_for_backward(mr, sr, ax if ax is not None else mr, out)
        ctx.kind, ctx.stds = kind, (min_std, max_std)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        mr, sr, ax, out = ctx.saved_tensors
        dm, ds = torch.empty_like(mr), torch.empty_like(sr)
        lo, hi = ctx.stds
        up = _c(dout.reshape(out.shape))
        if ctx.kind == "sample":
            ops.actor_normal_bwd(mr, sr, dm, ds, eps=ax, action=out, daction=up, min_std=lo, max_std=hi)
        elif ctx.kind == "entropy":
            ops.actor_normal_bwd(mr, sr, dm, ds, dent=up, min_std=lo, max_std=hi)
        else:
            ops.actor_normal_bwd(mr, sr, dm, ds, action=ax, dlogp=up, min_std=lo, max_std=hi)
        return dm, ds, None, None, None, None


class TanhFn(Function):
    @staticmethod
    def forward(ctx, x):
        xs = x.detach().to(F32).contiguous()
        y = torch.empty_like(xs)
        ops.tanh_fwd(xs, y)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        ops.tanh_bwd(y, _c(dy.reshape(y.shape)), dx)
        return dx
