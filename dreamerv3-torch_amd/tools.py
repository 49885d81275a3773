"""Hot-path subset of the reference's `tools` module, re-implemented over libdv3hip.

Same names, argument meaning and return shapes as the reference (tools.py) for everything the
world-model training path touches: the distribution wrappers returned by heads / RSSM.get_dist,
`lambda_return`, `Optimizer`, `RequiresGrad`, `static_scan`, initialisers and schedules.  The
reference's host-side I/O helpers (Logger, simulate, load_episodes, save_episodes, enable_deterministic_run,
...) are out of scope and are not re-implemented: names this module does not define resolve,
lazily, to the integrator's OWN reference `tools.py` (module-level __getattr__ below; its path comes from the
environment variable DV3_REFERENCE_TOOLS or from a `tools.py` found further down sys.path), so the reference's
`dreamer.py` runs against this module with no import edits.

Distribution objects are views over kernel outputs.  The fused training path never differentiates through them
(models.WorldModel._train / ImagBehavior._train run the hand-derived backward in dv3hip.engine) and acting / logging
only need values; when a caller builds a loss with autograd on top of them (exploration.Plan2Explore, the causal
world models: SURVEY.md 8(f) N4) and their parameters carry a graph, every method goes through the matching
dv3hip.autograd Function -- the same kernels forward and backward.
"""
from __future__ import annotations

import importlib.util
import math
import os
import re
import sys

import numpy as np
import torch
from torch import nn

from dv3hip import autograd as AG
from dv3hip import ops
from dv3hip.params import ParamBucket

to_np = lambda x: x.detach().cpu().numpy()


def _dev_f32(x, device=None):
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    if device is not None and x.device != torch.device(device):
        x = x.to(device)
    return x.to(torch.float32).contiguous()


def symlog(x):  # tools.py:22-23
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    return ops.symlog(x.contiguous(), out)


def symexp(x):  # tools.py:26-27 (host-side convenience; the kernels fuse it)
    return torch.sign(x) * (torch.exp(torch.abs(x)) - 1.0)


class RequiresGrad:  # tools.py:30-38
    def __init__(self, model):
        self._model = model

    def __enter__(self):
        self._model.requires_grad_(requires_grad=True)

    def __exit__(self, *args):
        self._model.requires_grad_(requires_grad=False)


# ---------------------------------------------------------------------------------------------
# distributions
# ---------------------------------------------------------------------------------------------
class OneHotDist:
    """tools.py:436-460: one-hot categorical over the last dim with `unimix_ratio` uniform mixing."""

    def __init__(self, logits=None, probs=None, unimix_ratio=0.0, rng=None):
        if logits is None:
            raise NotImplementedError("OneHotDist needs logits")
        self._logits = logits.contiguous()
        self._unimix = float(unimix_ratio)
        self._rng = rng

    @property
    def logits(self):
        return self._logits

    def mode(self):
        if AG.wants_grad(self._logits):
            return AG.OneHotSampleFn.apply(self._logits, None, None, self._unimix, True)
        out = torch.empty_like(self._logits)
        return ops.onehot_sample(self._logits, out, unimix=self._unimix, mode=True)

    def sample(self, sample_shape=(), seed=None, noise=None):
        if seed is not None:
            raise ValueError("need to check")
        if tuple(sample_shape) != ():
            raise NotImplementedError("sample_shape")
        out = torch.empty_like(self._logits)
        rng = None
        if noise is None:
            rng = self._rng if self._rng is not None else default_rng(self._logits.device)
        if AG.wants_grad(self._logits):  # exact one-hot forward, straight-through gradient (tools.py:452-460)
            out = AG.OneHotSampleFn.apply(self._logits, noise, rng, self._unimix, False)
            if rng is not None:
                rng.commit()
            return out
        ops.onehot_sample(self._logits, out, noise=noise, rng=rng, unimix=self._unimix)
        if rng is not None:
            rng.commit()
        return out

    def entropy(self):
        """Entropy per categorical group (callers wrap in Independent to sum over groups)."""
        if AG.wants_grad(self._logits):
            return AG.OneHotEntLogpFn.apply(self._logits, None, self._unimix, True)
        ent = torch.empty(self._logits.shape[:-1], device=self._logits.device)
        ops.onehot_ent_logp_fwd(self._logits, None, ent, None, unimix=self._unimix)
        return ent

    def log_prob(self, x):
        if AG.wants_grad(self._logits):
            return AG.OneHotEntLogpFn.apply(self._logits, x, self._unimix, False)
        lp = torch.empty(self._logits.shape[:-1], device=self._logits.device)
        ops.onehot_ent_logp_fwd(self._logits, x.contiguous(), None, lp, unimix=self._unimix)
        return lp


class IndependentOneHot:
    """Independent(OneHotDist, 1), what RSSM.get_dist returns (networks.py:161-166)."""

    def __init__(self, base: OneHotDist):
        self.base_dist = base

    def sample(self, sample_shape=(), noise=None):
        return self.base_dist.sample(sample_shape, noise=noise)

    def mode(self):
        return self.base_dist.mode()

    def entropy(self):
        return self.base_dist.entropy().sum(-1)

    def log_prob(self, x):
        return self.base_dist.log_prob(x).sum(-1)


class DiscDist:
    """tools.py:463-517: 255-bucket symlog two-hot head."""

    def __init__(self, logits, low=-20.0, high=20.0, transfwd=None, transbwd=None, device=None):
        if logits.shape[-1] != 255 or (low, high) != (-20.0, 20.0):
            raise NotImplementedError("DiscDist kernels are specialised for linspace(-20, 20, 255)")
        self.logits = logits.contiguous()

    def mean(self):
        if AG.wants_grad(self.logits):
            return AG.DiscModeFn.apply(self.logits)
        out = torch.empty(self.logits.shape[:-1] + (1,), device=self.logits.device)
        return ops.disc_mode_fwd(self.logits, out)

    mode = mean

    def log_prob(self, x):
        x = x.to(torch.float32)
        if x.dim() == self.logits.dim() and x.shape[-1] == 1:
            x = x[..., 0]
        if AG.wants_grad(self.logits):
            return AG.DiscLogProbFn.apply(self.logits, x)
        out = torch.empty(self.logits.shape[:-1], device=self.logits.device)
        return ops.disc_logprob_fwd(self.logits, x.contiguous(), out)


class MSEDist:
    """tools.py:520-540 (agg='sum')."""

    def __init__(self, mode, agg="sum"):
        if agg != "sum":
            raise NotImplementedError(agg)
        self._mode = mode

    def mode(self):
        return self._mode

    def mean(self):
        return self._mode

    def log_prob(self, value):
        assert self._mode.shape == value.shape, (self._mode.shape, value.shape)
        d = (self._mode - value) ** 2
        return -d.sum(list(range(d.dim()))[2:])


class SymlogDist:
    """tools.py:543-572 (dist='mse', agg='sum')."""

    def __init__(self, mode, dist="mse", agg="sum", tol=1e-8):
        if dist != "mse" or agg != "sum":
            raise NotImplementedError((dist, agg))
        self._mode = mode.contiguous()

    def mode(self):
        return symexp(self._mode)

    def mean(self):
        return symexp(self._mode)

    def log_prob(self, value):
        assert self._mode.shape == value.shape
        if AG.wants_grad(self._mode):
            return AG.SymlogMSEFn.apply(self._mode, value)
        loss = torch.empty(self._mode.shape[:-1], device=self._mode.device)
        ops.symlog_mse(self._mode, value.to(torch.float32).contiguous(), loss)
        return -loss


class ContDist:
    """tools.py:575-601 around Normal(tanh(mean), std) with absmax -- the continuous actor
    (networks.py:693-700).  Holds the two raw head outputs."""

    def __init__(self, mean_raw, std_raw, min_std, max_std, absmax=None, rng=None):
        if absmax not in (None, 1.0):
            raise NotImplementedError("absmax other than 1.0")
        self._mr, self._sr = mean_raw.contiguous(), std_raw.contiguous()
        self._min, self._max = float(min_std), float(max_std)
        self.absmax = absmax
        self._rng = rng

    def _grad(self):
        return AG.wants_grad(self._mr, self._sr)

    @property
    def mean(self):
        return AG.TanhFn.apply(self._mr) if self._grad() else torch.tanh(self._mr)

    def mode(self):
        out = self.mean
        if self.absmax is not None:  # tools.py:584-587: the rescale factor is detached
            out = out * (self.absmax / torch.clip(torch.abs(out.detach()), min=self.absmax))
        return out

    def sample(self, sample_shape=(), noise=None):
        if tuple(sample_shape) != ():
            raise NotImplementedError("sample_shape")
        if noise is None:
            rng = self._rng if self._rng is not None else default_rng(self._mr.device)
            noise = torch.empty_like(self._mr)
            ops.fill_normal(noise, rng)
            rng.commit()
        if self.absmax is None:
            raise NotImplementedError("ContDist.sample without absmax (the sampling kernel applies the absmax-1 rescale)")
        if self._grad():
            return AG.NormalFn.apply(self._mr, self._sr, noise, "sample", self._min, self._max)
        action = torch.empty_like(self._mr)
        ops.actor_normal_fwd(self._mr, self._sr, noise.contiguous(), action, None, min_std=self._min,
                             max_std=self._max)
        return action

    def entropy(self):
        if self._grad():
            return AG.NormalFn.apply(self._mr, self._sr, None, "entropy", self._min, self._max)
        ent = torch.empty(self._mr.shape[:-1], device=self._mr.device)
        ops.actor_normal_fwd(self._mr, self._sr, None, None, ent, min_std=self._min, max_std=self._max)
        return ent

    def log_prob(self, x):
        if self._grad():
            return AG.NormalFn.apply(self._mr, self._sr, x, "logp", self._min, self._max)
        lp = torch.empty(self._mr.shape[:-1], device=self._mr.device)
        ops.actor_normal_logp(self._mr, self._sr, x.to(torch.float32).contiguous(), lp, min_std=self._min,
                              max_std=self._max)
        return lp


class Bernoulli:
    """tools.py:604-628 over logits [..., 1]."""

    def __init__(self, logits):
        self._logits = logits.contiguous()

    @property
    def mean(self):
        return torch.sigmoid(self._logits)

    def mode(self):
        return torch.round(torch.sigmoid(self._logits))

    def log_prob(self, x):
        if AG.wants_grad(self._logits):
            return AG.BernoulliLogProbFn.apply(self._logits, x).sum(-1)
        out = torch.empty_like(self._logits)
        ops.bernoulli_logprob_fwd(self._logits, x.to(torch.float32).contiguous(), out)
        return out.sum(-1)


_DEFAULT_RNG = {}
_RNG_OVERRIDE = {}


def _rng_key(device) -> str:
    """One key per physical device: "cuda" (config.device without an index) and the "cuda:0" a tensor reports are the
    same Philox stream."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


class rng_override:
    """`with tools.rng_override(device, stream):` -- samplers that ask for the default stream of `device` get `stream`
    (dv3hip.graph.UpdateRunner captures each phase of the pipelined update against a Philox state of its own)."""

    def __init__(self, device, stream):
        self._key, self._stream = _rng_key(device), stream

    def __enter__(self):
        self._prev = _RNG_OVERRIDE.get(self._key)
        _RNG_OVERRIDE[self._key] = self._stream
        return self._stream

    def __exit__(self, *exc):
        if self._prev is None:
            _RNG_OVERRIDE.pop(self._key, None)
        else:
            _RNG_OVERRIDE[self._key] = self._prev
        return False


def on_config_device(fn):
    """Method decorator: run with `self._config.device` as the CURRENT device.  The kernels are launched on
    torch.cuda.current_stream() -- the current device's -- so an agent built on "cuda:1" in a process whose current device
    is 0 (the reference's `--device cuda:1`; torch's own ops switch per call) needs the switch at its entry points."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        dev = torch.device(self._config.device)
        if dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)

    return wrapper


def default_rng(device, seed=None):
    """Device-resident Philox stream shared by every sampler that gets no injected noise."""
    key = _rng_key(device)
    if seed is None and key in _RNG_OVERRIDE:
        return _RNG_OVERRIDE[key]
    if key not in _DEFAULT_RNG:
        _DEFAULT_RNG[key] = ops.RngStream(torch.device(device), seed or 0)
    elif seed is not None:
        _DEFAULT_RNG[key].reseed(seed)
    return _DEFAULT_RNG[key]


def set_seed_everywhere(seed):  # tools.py:961-966
    torch.manual_seed(seed)
    np.random.seed(seed)
    for k in list(_DEFAULT_RNG):
        _DEFAULT_RNG[k].reseed(seed)


# ---------------------------------------------------------------------------------------------
# returns
# ---------------------------------------------------------------------------------------------
def lambda_return(reward, value, pcont, bootstrap, lambda_, axis):
    """tools.py:702-728 for the layout ImagBehavior uses (axis 0, tensors [H-1, N, 1]).

    reward = r[1:], value = v[:-1], pcont = disc[1:], bootstrap = v[-1]; returns [H-1, N, 1]."""
    if axis != 0:
        raise NotImplementedError("axis != 0")
    h1 = reward.shape[0]
    n = reward[0].numel()
    dev = reward.device
    full_r = torch.zeros(h1 + 1, n, device=dev)
    full_r[1:] = reward.reshape(h1, n)
    full_v = torch.empty(h1 + 1, n, device=dev)
    full_v[:-1] = value.reshape(h1, n)
    full_v[-1] = bootstrap.reshape(n)
    # the kernel takes the continue LOGIT; disc = 1 * sigmoid(logit) => logit = log(d / (1 - d))
    d = pcont.reshape(h1, n).clamp(1e-7, 1 - 1e-7)
    full_c = torch.zeros(h1 + 1, n, device=dev)
    full_c[1:] = torch.log(d) - torch.log1p(-d)
    target = torch.empty(h1, n, device=dev)
    weights = torch.empty(h1 + 1, n, device=dev)
    ops.lambda_return_fwd(full_r, full_v, full_c, target, weights, None, gamma=1.0, lam=float(lambda_))
    return target.reshape(reward.shape)


# ---------------------------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------------------------
class _BucketAdam(torch.optim.Optimizer):
    """The `_opt` member of the reference's Optimizer (tools.py:752: torch.optim.Adam) as far as checkpoints see
    it: tools.recursively_collect_optim_state_dict (tools.py:975-1002) looks for torch.optim.Optimizer instances
    and saves their state_dict() under the attribute path (`_wm._model_opt._opt`, ...).  This one serialises the
    flat bucket in torch.optim.Adam's format and loads that format back, so `optims_state_dict` of a reference
    checkpoint and of this build are interchangeable.  The update itself is ParamBucket.step (dv3_adam_step)."""

    def __init__(self, bucket, lr, eps, weight_decay):
        self._bucket = bucket
        super().__init__(bucket.params, dict(lr=lr, betas=(0.9, 0.999), eps=eps, weight_decay=weight_decay,
                                             amsgrad=False, maximize=False, foreach=None, capturable=False,
                                             differentiable=False, fused=None))

    def step(self, closure=None):
        raise RuntimeError("the update runs in tools.Optimizer.finish() (flat-bucket clip + Adam kernel)")

    def state_dict(self):
        b = self._bucket.ensure()
        steps = float(b.state[0])
        state = {}
        if steps > 0:
            for i, (p, (off, n)) in enumerate(zip(b.params, b._layout)):
                state[i] = {"step": torch.tensor(steps, dtype=torch.float32),
                            "exp_avg": b.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": b.exp_avg_sq[off:off + n].view(p.shape).clone()}
        groups = [dict({k: v for k, v in g.items() if k != "params"}, params=list(range(len(g["params"]))))
                  for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        b = self._bucket.ensure()
        st = state_dict["state"]
        if len(st) not in (0, len(b.params)):
            raise ValueError(f"optimizer state has {len(st)} entries for {len(b.params)} parameters")
        b.exp_avg.zero_(), b.exp_avg_sq.zero_()
        steps = 0.0
        for i, (p, (off, n)) in enumerate(zip(b.params, b._layout)):
            e = st.get(i, st.get(str(i)))
            if e is None:
                continue
            if tuple(e["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {i}: shape {tuple(e['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
            b.exp_avg[off:off + n].copy_(e["exp_avg"].reshape(-1))
            b.exp_avg_sq[off:off + n].copy_(e["exp_avg_sq"].reshape(-1))
            steps = float(e["step"])
        b.state.zero_()
        b.state[0] = steps
        g = state_dict["param_groups"][0]
        for k in ("lr", "eps", "weight_decay"):
            if k in g:
                self.param_groups[0][k] = g[k]


class Optimizer:
    """tools.py:731-783 surface over a flat bucket: zero_grad / backward into .grad / all-reduce / clip / Adam.
    The hot path's backward is explicit, so it uses the two halves `begin()` ... `finish(loss)`; the reference's
    `__call__(loss, params)` is kept for autograd-built losses."""

    def __init__(self, name, parameters, lr, eps=1e-4, clip=None, wd=None, wd_pattern=r".*", opt="adam",
                 use_amp=False, extra=0):
        assert 0 <= (wd or 0) < 1
        assert not clip or 1 <= clip
        if opt != "adam":
            raise NotImplementedError(f"{opt} is not implemented")
        if use_amp:
            raise NotImplementedError("precision 16 is not supported: the path is fp32 (configs.yaml:18)")
        if wd_pattern != r".*":
            raise NotImplementedError
        self._name, self._clip = name, clip
        self.bucket = ParamBucket(name, parameters, extra=extra)  # extra: see ParamBucket (floats riding the all-reduce)
        self._opt = _BucketAdam(self.bucket, float(lr), float(eps), wd or 0.0)  # hyper-parameters live in its group

    @property
    def _lr(self):
        return self._opt.param_groups[0]["lr"]

    @property
    def _eps(self):
        return self._opt.param_groups[0]["eps"]

    @property
    def _wd(self):
        return self._opt.param_groups[0]["weight_decay"]

    def begin(self):
        self.bucket.ensure().zero_grad()

    def __call__(self, loss, params, retain_graph=True):
        """tools.py:760-776 for callers that built `loss` with torch autograd over parameters of this optimizer
        (exploration.py:104): zero_grad -> loss.backward() into the bucket's .grad views -> all-reduce -> clip ->
        Adam.  The hot path (WorldModel / ImagBehavior._train) does not come through here: its backward is explicit
        (begin() ... finish())."""
        if not isinstance(loss, torch.Tensor) or not loss.requires_grad:
            raise RuntimeError(f"{self._name}_opt(loss, params): loss carries no autograd graph; the accelerated "
                               "modules run an explicit backward -- use begin() / finish(loss)")
        assert len(loss.shape) == 0, loss.shape
        mine = {id(p) for p in self.bucket.params}
        if any(id(p) not in mine for p in params):
            raise ValueError("params are not the parameters this Optimizer was built over")
        self.begin()
        loss.backward(retain_graph=retain_graph)
        # snapshots: the norm lives in the bucket's state vector, which the next step overwrites
        return {k: v.detach().clone() for k, v in self.finish(loss.detach()).items()}

    def finish(self, loss, allreduce=True):
        """loss: 0-d device tensor (metric only).  Returns the reference's metric dict with device scalars.
        allreduce=False: the caller already all-reduced the gradient bucket (hipGraph replay keeps the
        collective outside the captured segment); only the 1/world scale is applied."""
        if allreduce:
            scale = self.bucket.allreduce()
        else:
            import torch.distributed as dist

            scale = 1.0 / dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1.0
        self.bucket.step(lr=self._lr, eps=self._eps, clip=self._clip, weight_decay=self._wd, grad_scale=scale)
        return {f"{self._name}_loss": loss, f"{self._name}_grad_norm": self.bucket.grad_norm}

    def state_dict(self):
        """torch.optim.Adam format (see _BucketAdam)."""
        return self._opt.state_dict()

    def load_state_dict(self, sd):
        self._opt.load_state_dict(sd)


def recursively_collect_optim_state_dict(obj, path="", optimizers_state_dicts=None, visited=None):
    """tools.py:975-1002: {attribute path: state_dict} of every torch.optim.Optimizer reachable from obj."""
    out = {} if optimizers_state_dicts is None else optimizers_state_dicts
    seen = set() if visited is None else visited
    if id(obj) in seen:
        return out
    seen.add(id(obj))
    attrs = dict(getattr(obj, "__dict__", {}))
    if isinstance(obj, torch.nn.Module):
        attrs.update({k: m for k, m in obj.named_modules() if k and "." not in k})
    for name, attr in attrs.items():
        here = f"{path}.{name}" if path else name
        if isinstance(attr, torch.optim.Optimizer):
            out[here] = attr.state_dict()
        elif hasattr(attr, "__dict__"):
            recursively_collect_optim_state_dict(attr, here, out, seen)
    return out


def recursively_load_optim_state_dict(obj, optimizers_state_dicts):
    """tools.py:1005-1011."""
    for path, sd in optimizers_state_dicts.items():
        cur = obj
        for key in path.split("."):
            cur = getattr(cur, key)
        cur.load_state_dict(sd)


# ---------------------------------------------------------------------------------------------
# scan, schedules, init
# ---------------------------------------------------------------------------------------------
def static_scan(fn, inputs, start):
    """tools.py:806-850 semantics (fold + stack on a new dim 0) without the quadratic re-concatenation."""
    last = start
    outs = []
    for i in range(inputs[0].shape[0]):
        last = fn(last, *[inp[i] for inp in inputs])
        outs.append(last)
    first = outs[0]
    if isinstance(first, dict):
        return [{k: torch.stack([o[k] for o in outs], 0) for k in first}]
    res = []
    for j in range(len(first)):
        if isinstance(first[j], dict):
            res.append({k: torch.stack([o[j][k] for o in outs], 0) for k in first[j]})
        else:
            res.append(torch.stack([o[j] for o in outs], 0))
    return res


# ---------------------------------------------------------------------------------------------
# Replay sampling: the two generators between the episode store and WorldModel._train.
# Same draws, same batches as the reference for the same seed (tests/golden/replay.npz), without its
# per-piece np.append / per-key Python list stacking: every sequence is written once into a preallocated
# [length, ...] array and every batch into preallocated [B, T, ...] arrays.
# ---------------------------------------------------------------------------------------------
def sample_episodes(episodes, length, seed=0):
    """tools.py:324-371: endless generator of dicts of [length, ...] arrays cut from `episodes` (dict
    name -> dict of per-step arrays).  A sequence starts at a uniform offset of an episode drawn with
    probability proportional to its length and continues from the START of further draws until `length`
    steps are collected; is_first is forced at the first step and at every join; keys containing "log_"
    are dropped; episodes shorter than 2 steps are redrawn.  Data-parallel ranks pass seed + rank."""
    rng = np.random.RandomState(seed)
    while True:
        eps = list(episodes.values())
        lens = np.array([len(next(iter(ep.values()))) for ep in eps])
        p = lens / np.sum(lens)
        out, size = None, 0
        while size < length:
            ep = eps[int(rng.choice(len(eps), p=p))]
            total = len(next(iter(ep.values())))
            if total < 2:
                continue
            if out is None:
                index = int(rng.randint(0, total - 1))
                keys = [k for k in ep if "log_" not in k]
                out = {k: np.empty((length,) + ep[k].shape[1:], ep[k].dtype) for k in keys}
            else:
                index = 0
            n = min(length - size, total - index)
            for k in out:
                out[k][size:size + n] = ep[k][index:index + n]
            if "is_first" in out:
                out["is_first"][size] = True
            size += n
        yield out


def from_generator(generator, batch_size):
    """tools.py:310-321: stack `batch_size` draws of `generator` on a new leading axis."""
    while True:
        first = next(generator)
        data = {k: np.empty((batch_size,) + v.shape, v.dtype) for k, v in first.items()}
        for i in range(batch_size):
            item = first if i == 0 else next(generator)
            for k, v in item.items():
                data[k][i] = v
        yield data


class TimeRecording:  # tools.py:41-53
    """`with TimeRecording("label"):` -- HIP-event timer around a region of the current stream; prints the label
    and the elapsed seconds on exit (and keeps them in .seconds)."""

    def __init__(self, comment):
        self._comment = comment
        self.seconds = None

    def __enter__(self):
        self._start = torch.cuda.Event(enable_timing=True)
        self._stop = torch.cuda.Event(enable_timing=True)
        self._start.record()
        return self

    def __exit__(self, *exc):
        self._stop.record()
        self._stop.synchronize()
        self.seconds = self._start.elapsed_time(self._stop) / 1000
        print(self._comment, self.seconds)


class Every:  # tools.py:853-868
    def __init__(self, every):
        self._every, self._last = every, None

    def __call__(self, step):
        if not self._every:
            return 0
        if self._last is None:
            self._last = step
            return 1
        count = int((step - self._last) / self._every)
        self._last += self._every * count
        return count


class Once:  # tools.py:871-879
    def __init__(self):
        self._once = True

    def __call__(self):
        if self._once:
            self._once = False
            return True
        return False


class Until:  # tools.py:882-887
    def __init__(self, until):
        self._until = until

    def __call__(self, step):
        return True if not self._until else step < self._until


def _fans(m):
    if isinstance(m, nn.Linear):
        return m.in_features, m.out_features
    space = m.kernel_size[0] * m.kernel_size[1]
    return space * m.in_channels, space * m.out_channels


def weight_init(m):
    """tools.py:890-917: truncated normal, std = sqrt(1/avg_fan)/0.8796, cut at 2 std; LN -> (1, 0)."""
    if isinstance(m, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)):
        fin, fout = _fans(m)
        std = math.sqrt(1.0 / ((fin + fout) / 2.0)) / 0.87962566103423978
        nn.init.trunc_normal_(m.weight.data, mean=0.0, std=std, a=-2.0 * std, b=2.0 * std)
        if getattr(m, "bias", None) is not None:
            m.bias.data.fill_(0.0)
    elif isinstance(m, nn.LayerNorm):
        m.weight.data.fill_(1.0)
        m.bias.data.fill_(0.0)


def uniform_weight_init(given_scale):
    """tools.py:920-946: uniform(+-sqrt(3*scale/avg_fan))."""

    def f(m):
        if isinstance(m, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)):
            fin, fout = _fans(m)
            limit = math.sqrt(3 * given_scale / ((fin + fout) / 2.0))
            nn.init.uniform_(m.weight.data, a=-limit, b=limit)
            if getattr(m, "bias", None) is not None:
                m.bias.data.fill_(0.0)
        elif isinstance(m, nn.LayerNorm):
            m.weight.data.fill_(1.0)
            m.bias.data.fill_(0.0)

    return f


def tensorstats(tensor, prefix=None, *, shift=None, scale=None):
    """tools.py:949-958: mean / std / min / max as device scalars (converted lazily).  One launch for the four
    reductions on device float tensors; shift / scale (1-element device tensors): statistics of (tensor - shift) /
    scale without materialising it."""
    if tensor.is_cuda and tensor.dtype == torch.float32 and tensor.numel() > 0:
        out = torch.empty(4, device=tensor.device)
        ops.tensorstats(tensor.detach().contiguous(), out, shift=shift, scale=scale)
        metrics = {"mean": out[0], "std": out[1], "min": out[2], "max": out[3]}
    else:
        if shift is not None:
            tensor = (tensor - shift) / (1.0 if scale is None else scale)
        metrics = {"mean": torch.mean(tensor), "std": torch.std(tensor), "min": torch.min(tensor),
                   "max": torch.max(tensor)}
    return {f"{prefix}_{k}": v for k, v in metrics.items()} if prefix else metrics


def tensorstats_many(groups):
    """tools.tensorstats for several tensors in ONE launch: groups = [(tensor, prefix, shift, scale), ...] (at most 6,
    device float32) -> the merged metric dict."""
    out = torch.empty(len(groups), 4, device=groups[0][0].device)
    ops.tensorstats_multi([(t.detach().contiguous(), sh, sc) for t, _, sh, sc in groups], out)
    metrics = {}
    for i, (_, prefix, _, _) in enumerate(groups):
        for j, k in enumerate(("mean", "std", "min", "max")):
            metrics[f"{prefix}_{k}"] = out[i, j]
    return metrics


def args_type(default):  # tools.py:786-803
    def parse_string(x):
        if default is None:
            return x
        if isinstance(default, bool):
            return bool(["False", "True"].index(x))
        if isinstance(default, int):
            return float(x) if ("e" in x or "." in x) else int(x)
        if isinstance(default, (list, tuple)):
            return tuple(args_type(default[0])(y) for y in x.split(","))
        return type(default)(x)

    def parse_object(x):
        return tuple(x) if isinstance(default, (list, tuple)) else x

    return lambda x: parse_string(x) if isinstance(x, str) else parse_object(x)


def load_config(path, blocks):
    """configs.yaml loader (dreamer.py:578-596 semantics): `defaults` then each named block, merged
    recursively; PyYAML reads 1e6-style numbers as strings, so coerce them."""
    import yaml

    raw = yaml.safe_load(open(path))

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
    return cfg


# ---------------------------------------------------------------------------------------------
# delegation of the host-side helpers to the integrator's reference tools.py
# ---------------------------------------------------------------------------------------------
# Names of the reference's tools.py that the drivers (dreamer.py:356-567, main_with_causal.py, eval scripts) use and
# that are NOT part of the accelerated path: resolved from the reference file at the integrator's site.  Nothing of
# that file is copied here.
PASSTHROUGH = ("Logger", "simulate", "load_episodes", "save_episodes", "add_to_cache", "erase_over_episodes",
               "convert", "enable_deterministic_run", "schedule", "static_scan_for_lambda_return", "SampleDist",
               "SafeTruncatedNormal", "TanhBijector", "UnnormalizedHuber")
_REFERENCE_TOOLS = None


def _find_reference_tools():
    cand = os.environ.get("DV3_REFERENCE_TOOLS")
    if cand:
        if os.path.isdir(cand):
            cand = os.path.join(cand, "tools.py")
        if not os.path.isfile(cand):
            raise ImportError(f"DV3_REFERENCE_TOOLS={cand}: no such file")
        return cand
    here = os.path.abspath(__file__)
    for d in sys.path:  # DV3_REFERENCE_TOOLS unset: the reference checkout further down sys.path (its own dreamer.py's dir)
        f = os.path.abspath(os.path.join(d or ".", "tools.py"))
        if f != here and os.path.isfile(f):
            try:
                text = open(f, errors="ignore").read()
            except OSError:
                continue
            if "def simulate(" in text and "class Logger" in text:  # looks like the reference's: only then execute it
                return f
    return None


def _reference_tools():
    global _REFERENCE_TOOLS
    if _REFERENCE_TOOLS is None:
        path = _find_reference_tools()
        if path is None:
            raise ImportError("the reference's tools.py was not found: set DV3_REFERENCE_TOOLS=/path/to/reference "
                              "(or keep the reference directory on sys.path behind this package)")
        spec = importlib.util.spec_from_file_location("_dv3_reference_tools", path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["_dv3_reference_tools"] = mod
        try:
            spec.loader.exec_module(mod)
        except BaseException:
            sys.modules.pop("_dv3_reference_tools", None)  # never leave a half-initialised module behind
            raise
        if not (hasattr(mod, "simulate") and hasattr(mod, "Logger")):
            sys.modules.pop("_dv3_reference_tools", None)
            raise ImportError(f"{path} is not the reference's tools.py (no simulate / Logger)")
        _REFERENCE_TOOLS = mod
    return _REFERENCE_TOOLS


def __getattr__(name):
    """PEP 562: attributes this module does not define come from the integrator's reference tools.py."""
    if name.startswith("__") or name not in PASSTHROUGH:
        # only the listed host-side helpers are delegated: a typo or a hasattr() probe never executes a foreign file
        raise AttributeError(f"module 'tools' (MI355X drop-in) has no attribute {name!r}")
    try:
        ref = _reference_tools()
    except ImportError as e:
        raise AttributeError(f"tools.{name} is a host-side helper of the reference that this drop-in does not "
                             f"re-implement, and {e}") from None
    try:
        return getattr(ref, name)
    except AttributeError:
        raise AttributeError(f"neither the MI355X tools module nor {ref.__file__} defines {name!r}") from None
