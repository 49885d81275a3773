"""CPU oracle for the dreamerv3-torch world-model training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the `dreamerv3-torch_amd/`
package, its C-ABI library, `networks.py`/`models.py`/`tools.py`) may import this
module.  Allowed importers: `tests/`, `__graft_entry__.smoke()`, and the
`cpu_baseline` leg of `bench.py` (as the checker / the timed CPU port, never as
the thing shipped).

What it is: a functional, plain-torch (CPU, fp32) restatement of the reference's
algorithm for the hot path, written from the math, with
every random draw made an explicit input so that the HIP path can be fed the
same noise.  Each function cites the reference file:line it follows
(paths are relative to the reference checkout).

Parity pin: the reference has no tests or golden vectors of its own.  This
oracle is pinned against outputs of the reference itself, produced in the build
container by `tests/golden/make_golden.py` (which imports the reference's
`networks`/`models`/`tools` unmodified) and committed as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against them.

Parameters are passed as a flat dict keyed by the reference's `state_dict`
names, e.g. ``p["dynamics._cell.layers.GRU_linear.weight"]``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-3  # networks.py:55,66,75,631,754,802


# --------------------------------------------------------------------------------------
# configuration of the path (the subset of configs.yaml the hot path reads)
# --------------------------------------------------------------------------------------
@dataclass
class PathConfig:
    stoch: int = 32  # dyn_stoch           configs.yaml:67
    discrete: int = 32  # dyn_discrete     configs.yaml:68
    deter: int = 512  # dyn_deter          configs.yaml:66
    hidden: int = 512  # dyn_hidden        configs.yaml:65
    units: int = 512  # units              configs.yaml:74
    num_actions: int = 6
    unimix: float = 0.01  # configs.yaml:92
    cnn_depth: int = 32  # encoder/decoder cnn_depth
    encoder: str = "cnn"  # "cnn" (dmc_vision), "mlp" (dmc_proprio) or "both" (image + vector keys: minecraft)
    mlp_keys: Tuple[Tuple[str, int], ...] = ()  # proprio keys and their widths, in obs_space order
    enc_mlp_layers: int = 5
    enc_mlp_units: int = 1024
    actor_layers: int = 2
    actor_dist: str = "normal"  # "normal" | "onehot"
    actor_min_std: float = 0.1
    actor_max_std: float = 1.0
    actor_entropy: float = 3e-4
    critic_layers: int = 2
    reward_layers: int = 2
    cont_layers: int = 2
    imag_gradient_mix: float = 0.0  # configs.yaml:112
    imag_gradient: str = "dynamics"  # "dynamics" | "reinforce"
    horizon: int = 15
    discount: float = 0.997
    discount_lambda: float = 0.95
    kl_free: float = 1.0
    dyn_scale: float = 0.5
    rep_scale: float = 0.1
    ema_alpha: float = 1e-2  # models.py:14
    slow_target_fraction: float = 0.02

    @property
    def sd(self) -> int:
        return self.stoch * self.discrete

    @property
    def feat(self) -> int:
        return self.sd + self.deter

    @property
    def embed(self) -> int:
        e = self.cnn_depth * 8 * 16 if self.encoder in ("cnn", "both") else 0
        return e + (self.enc_mlp_units if self.encoder in ("mlp", "both") else 0)


# --------------------------------------------------------------------------------------
# scalar maps                                                          tools.py:22-27
# --------------------------------------------------------------------------------------
def symlog(x: Tensor) -> Tensor:
    return torch.sign(x) * torch.log(torch.abs(x) + 1.0)


def symexp(x: Tensor) -> Tensor:
    return torch.sign(x) * (torch.exp(torch.abs(x)) - 1.0)


def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def dense_ln_silu(x: Tensor, W: Tensor, g: Tensor, b: Tensor) -> Tensor:
    """Linear(no bias) -> LayerNorm(eps 1e-3) -> SiLU   (networks.py:48-58, 625-636)."""
    return F.silu(layer_norm(x @ W.t(), g, b))


# --------------------------------------------------------------------------------------
# one-hot categorical with unimix                                     tools.py:436-460
# --------------------------------------------------------------------------------------
def unimix_logits(logit: Tensor, unimix: float) -> Tensor:
    """Normalised log-probs the reference's OneHotDist ends up holding.

    tools.py:439-442 builds probs = softmax*(1-u)+u/D, logits = log(probs); the torch base
    class (Categorical.__init__) then re-normalises logits -= logsumexp(logits).
    """
    d = logit.shape[-1]
    probs = F.softmax(logit, dim=-1) * (1.0 - unimix) + unimix / d
    lg = torch.log(probs)
    return lg - lg.logsumexp(dim=-1, keepdim=True)


def onehot_sample(logit: Tensor, q: Tensor, unimix: float) -> Tensor:
    """tools.py:452-460.  `q` ~ Exp(1), same shape as logit.

    OneHotCategorical.sample -> torch.multinomial(probs, 1) whose single-draw path is
    argmax(probs / q) with q ~ Exp(1) (verified bit-exact against torch 2.10 in the build
    container; see tests/golden/make_golden.py).  Forward value is an exact one-hot; the
    gradient is that of `probs` (straight-through).
    """
    lg = unimix_logits(logit, unimix)
    probs = F.softmax(lg, dim=-1)  # == Categorical.probs (lazy, from normalised logits)
    idx = torch.argmax(probs.detach() / q, dim=-1)
    sample = F.one_hot(idx, logit.shape[-1]).to(logit.dtype)
    return sample + (probs - probs.detach())


def onehot_mode(logit: Tensor, unimix: float) -> Tensor:
    """tools.py:446-450: one_hot(argmax logits) + logits - logits.detach()."""
    lg = unimix_logits(logit, unimix)
    idx = torch.argmax(lg, dim=-1)
    mode = F.one_hot(idx, logit.shape[-1]).to(logit.dtype)
    return mode.detach() + lg - lg.detach()


def onehot_entropy(logit: Tensor, unimix: float) -> Tensor:
    """Independent(OneHotDist, 1).entropy(): -sum_d p log p, summed over the S groups."""
    lg = unimix_logits(logit, unimix)
    probs = F.softmax(lg, dim=-1)
    lg = torch.clamp(lg, min=torch.finfo(lg.dtype).min)
    return -(lg * probs).sum(-1).sum(-1)


def onehot_kl(logit_p: Tensor, logit_q: Tensor, unimix: float) -> Tensor:
    """torch.distributions.kl._kl_categorical_categorical, summed over the S groups."""
    lp = unimix_logits(logit_p, unimix)
    lq = unimix_logits(logit_q, unimix)
    pp = F.softmax(lp, dim=-1)
    t = pp * (lp - lq)
    return t.sum(-1).sum(-1)


def onehot_logprob(logit: Tensor, x: Tensor, unimix: float) -> Tensor:
    """OneHotCategorical.log_prob on one-hot `x` (used by the onehot actor)."""
    lg = unimix_logits(logit, unimix)
    idx = x.max(-1)[1]
    return lg.gather(-1, idx[..., None]).squeeze(-1)


# --------------------------------------------------------------------------------------
# RSSM                                                                networks.py:13-290
# --------------------------------------------------------------------------------------
def gru_cell(p: Dict[str, Tensor], x: Tensor, h: Tensor, pre: str = "dynamics._cell.") -> Tensor:
    """networks.py:760-768 (LayerNorm over all 3*deter jointly, update bias -1)."""
    parts = layer_norm(
        torch.cat([x, h], -1) @ p[pre + "layers.GRU_linear.weight"].t(),
        p[pre + "layers.GRU_norm.weight"],
        p[pre + "layers.GRU_norm.bias"],
    )
    de = h.shape[-1]
    reset, cand, update = parts[..., :de], parts[..., de : 2 * de], parts[..., 2 * de :]
    reset = torch.sigmoid(reset)
    cand = torch.tanh(reset * cand)
    update = torch.sigmoid(update - 1.0)
    return update * cand + (1.0 - update) * h


def prior_logit(cfg: PathConfig, p: Dict[str, Tensor], deter: Tensor, pre: str = "dynamics.") -> Tensor:
    """img_out layers + 'ims' stat layer   (networks.py:225-227, 241-250)."""
    x = dense_ln_silu(
        deter,
        p[pre + "_img_out_layers.0.weight"],
        p[pre + "_img_out_layers.1.weight"],
        p[pre + "_img_out_layers.1.bias"],
    )
    lg = x @ p[pre + "_imgs_stat_layer.weight"].t() + p[pre + "_imgs_stat_layer.bias"]
    return lg.reshape(list(lg.shape[:-1]) + [cfg.stoch, cfg.discrete])


def img_step(
    cfg: PathConfig,
    p: Dict[str, Tensor],
    stoch: Tensor,
    deter: Tensor,
    action: Tensor,
    q_prior: Optional[Tensor],
    sample: bool = True,
    pre: str = "dynamics.",
) -> Dict[str, Tensor]:
    """networks.py:208-233.  stoch [M,S,D], deter [M,De], action [M,A], q_prior [M,S,D]."""
    x = torch.cat([stoch.reshape(list(stoch.shape[:-2]) + [cfg.sd]), action], -1)
    x = dense_ln_silu(
        x,
        p[pre + "_img_in_layers.0.weight"],
        p[pre + "_img_in_layers.1.weight"],
        p[pre + "_img_in_layers.1.bias"],
    )
    deter = gru_cell(p, x, deter, pre + "_cell.")
    logit = prior_logit(cfg, p, deter, pre)
    if sample:
        st = onehot_sample(logit, q_prior, cfg.unimix)
    else:
        st = onehot_mode(logit, cfg.unimix)
    return {"stoch": st, "deter": deter, "logit": logit}


def initial(cfg: PathConfig, p: Dict[str, Tensor], batch: int, pre: str = "dynamics.") -> Dict[str, Tensor]:
    """networks.py:99-123 with initial='learned' (+ get_stoch 235-239)."""
    deter = torch.tanh(p[pre + "W"]).repeat(batch, 1)
    stoch = onehot_mode(prior_logit(cfg, p, deter, pre), cfg.unimix)
    logit = torch.zeros(batch, cfg.stoch, cfg.discrete, dtype=deter.dtype)
    return {"logit": logit, "stoch": stoch, "deter": deter}


def obs_step(
    cfg: PathConfig,
    p: Dict[str, Tensor],
    prev: Optional[Dict[str, Tensor]],
    prev_action: Optional[Tensor],
    embed: Tensor,
    is_first: Tensor,
    q_prior: Optional[Tensor],
    q_post: Optional[Tensor],
    sample: bool = True,
    pre: str = "dynamics.",
) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """networks.py:174-206, in the branch-free form of SURVEY.md Appendix A.

    The reference branches on sum(is_first) (all / some / none); blending every row with
    m = is_first is value-identical to all three branches (m=0: identity; m=1: initial
    state and zero action), and is what the GPU path does (no host sync).
    """
    m = is_first.to(embed.dtype)[:, None]
    b = embed.shape[0]
    init = initial(cfg, p, b, pre)
    if prev is None:
        prev = init
        prev_action = torch.zeros(b, cfg.num_actions, dtype=embed.dtype)
    else:
        prev_action = prev_action * (1.0 - m)
        blended = {}
        for k, v in prev.items():
            mr = m.reshape([b] + [1] * (v.dim() - 1))
            blended[k] = v * (1.0 - mr) + init[k] * mr
        prev = blended
    prior = img_step(cfg, p, prev["stoch"], prev["deter"], prev_action, q_prior, sample, pre)
    x = torch.cat([prior["deter"], embed], -1)
    x = dense_ln_silu(
        x,
        p[pre + "_obs_out_layers.0.weight"],
        p[pre + "_obs_out_layers.1.weight"],
        p[pre + "_obs_out_layers.1.bias"],
    )
    lg = x @ p[pre + "_obs_stat_layer.weight"].t() + p[pre + "_obs_stat_layer.bias"]
    lg = lg.reshape(b, cfg.stoch, cfg.discrete)
    st = onehot_sample(lg, q_post, cfg.unimix) if sample else onehot_mode(lg, cfg.unimix)
    post = {"stoch": st, "deter": prior["deter"], "logit": lg}
    return post, prior


def observe(
    cfg: PathConfig,
    p: Dict[str, Tensor],
    embed: Tensor,
    action: Tensor,
    is_first: Tensor,
    q_prior: Tensor,
    q_post: Tensor,
    state: Optional[Dict[str, Tensor]] = None,
    teacher: Optional[Dict[str, Tensor]] = None,
) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """networks.py:127-143 + tools.static_scan (tools.py:806-850) over T.

    embed [B,T,E], action [B,T,A], is_first [B,T], q_* [T,B,S,D].  Outputs are [B,T,...].
    `teacher` (optional): a dict of [B,T,...] posterior states; when given, step t is fed
    teacher state t-1 instead of its own output (teacher-forced parity).
    """
    t_len = embed.shape[1]
    posts, priors = [], []
    prev = state
    for t in range(t_len):
        if teacher is not None and t > 0:
            prev = {k: v[:, t - 1] for k, v in teacher.items()}
        post, prior = obs_step(
            cfg, p, prev, action[:, t], embed[:, t], is_first[:, t], q_prior[t], q_post[t]
        )
        posts.append(post)
        priors.append(prior)
        prev = post
    stack = lambda lst: {k: torch.stack([d[k] for d in lst], 1) for k in lst[0]}
    return stack(posts), stack(priors)


def get_feat(cfg: PathConfig, state: Dict[str, Tensor]) -> Tensor:
    """networks.py:154-159."""
    st = state["stoch"]
    return torch.cat([st.reshape(list(st.shape[:-2]) + [cfg.sd]), state["deter"]], -1)


def kl_loss(cfg: PathConfig, post_logit: Tensor, prior_logit_: Tensor):
    """networks.py:272-290 -> (loss, value, dyn_loss, rep_loss), each [B,T]."""
    rep = value = onehot_kl(post_logit, prior_logit_.detach(), cfg.unimix)
    dyn = onehot_kl(post_logit.detach(), prior_logit_, cfg.unimix)
    rep_c = torch.clip(rep, min=cfg.kl_free)
    dyn_c = torch.clip(dyn, min=cfg.kl_free)
    return cfg.dyn_scale * dyn_c + cfg.rep_scale * rep_c, value, dyn_c, rep_c


# --------------------------------------------------------------------------------------
# MLP trunk and heads                                                 networks.py:588-739
# --------------------------------------------------------------------------------------
def mlp_trunk(p: Dict[str, Tensor], pre: str, name: str, layers: int, x: Tensor) -> Tensor:
    """networks.py:624-636, 661: `layers` x [Linear(no bias); LN; SiLU]."""
    for i in range(layers):
        x = dense_ln_silu(
            x,
            p[f"{pre}layers.{name}_linear{i}.weight"],
            p[f"{pre}layers.{name}_norm{i}.weight"],
            p[f"{pre}layers.{name}_norm{i}.bias"],
        )
    return x


def head_logits(p, pre: str, name: str, layers: int, feat: Tensor) -> Tensor:
    h = mlp_trunk(p, pre, name, layers, feat)
    return h @ p[pre + "mean_layer.weight"].t() + p[pre + "mean_layer.bias"]


BUCKETS = 255


def disc_buckets(dtype=torch.float32) -> Tensor:
    return torch.linspace(-20.0, 20.0, steps=BUCKETS, dtype=dtype)  # tools.py:476


def disc_mode(logits: Tensor) -> Tensor:
    """DiscDist.mode == mean: symexp(sum softmax*buckets), keepdim   (tools.py:481-487)."""
    probs = torch.softmax(logits, -1)
    return symexp(torch.sum(probs * disc_buckets(logits.dtype), dim=-1, keepdim=True))


def disc_logprob(logits: Tensor, x: Tensor) -> Tensor:
    """DiscDist.log_prob (tools.py:490-513); x has the shape of logits[..., 0]."""
    buckets = disc_buckets(logits.dtype)
    x = symlog(x)
    below = torch.sum((buckets <= x[..., None]).to(torch.int32), dim=-1) - 1
    above = BUCKETS - torch.sum((buckets > x[..., None]).to(torch.int32), dim=-1)
    below = torch.clip(below, 0, BUCKETS - 1)
    above = torch.clip(above, 0, BUCKETS - 1)
    equal = below == above
    dist_to_below = torch.where(equal, 1, torch.abs(buckets[below] - x))
    dist_to_above = torch.where(equal, 1, torch.abs(buckets[above] - x))
    total = dist_to_below + dist_to_above
    w_below = dist_to_above / total
    w_above = dist_to_below / total
    target = (
        F.one_hot(below, BUCKETS) * w_below[..., None] + F.one_hot(above, BUCKETS) * w_above[..., None]
    )
    log_pred = logits - torch.logsumexp(logits, -1, keepdim=True)
    return (target * log_pred).sum(-1)


def bernoulli_logprob(logit: Tensor, x: Tensor) -> Tensor:
    """tools.Bernoulli.log_prob (tools.py:622-627); logit, x [...,1] -> [...]."""
    return torch.sum(-F.softplus(logit) * (1 - x) + -F.softplus(-logit) * x, -1)


def actor_stats(cfg: PathConfig, p, feat: Tensor, pre: str = "actor."):
    """networks.py:657-681 + dist 'normal' (693-700) / 'onehot' (713-714)."""
    h = mlp_trunk(p, pre, "Actor", cfg.actor_layers, feat)
    mean = h @ p[pre + "mean_layer.weight"].t() + p[pre + "mean_layer.bias"]
    if cfg.actor_dist == "normal":
        std = h @ p[pre + "std_layer.weight"].t() + p[pre + "std_layer.bias"]
        std = (cfg.actor_max_std - cfg.actor_min_std) * torch.sigmoid(std + 2.0) + cfg.actor_min_std
        return torch.tanh(mean), std
    return mean, None


def actor_sample(cfg: PathConfig, p, feat: Tensor, noise: Tensor, pre: str = "actor.") -> Tensor:
    """policy(feat).sample(): ContDist.sample with absmax=1 (tools.py:594-598) or OneHotDist.sample.

    noise: N(0,1) [M,A] for 'normal'; Exp(1) [M,A] for 'onehot'.
    """
    mean, std = actor_stats(cfg, p, feat, pre)
    if cfg.actor_dist == "normal":
        out = mean + std * noise
        return out * (1.0 / torch.clip(torch.abs(out), min=1.0)).detach()
    return onehot_sample(mean, noise, cfg.unimix)


def actor_entropy(cfg: PathConfig, p, feat: Tensor, pre: str = "actor.") -> Tensor:
    mean, std = actor_stats(cfg, p, feat, pre)
    if cfg.actor_dist == "normal":
        return (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)).sum(-1)
    lg = unimix_logits(mean, cfg.unimix)
    pr = F.softmax(lg, -1)
    return -(torch.clamp(lg, min=torch.finfo(lg.dtype).min) * pr).sum(-1)


def actor_logprob(cfg: PathConfig, p, feat: Tensor, action: Tensor, pre: str = "actor.") -> Tensor:
    mean, std = actor_stats(cfg, p, feat, pre)
    if cfg.actor_dist == "normal":
        var = std**2
        lp = -((action - mean) ** 2) / (2 * var) - torch.log(std) - math.log(math.sqrt(2 * math.pi))
        return lp.sum(-1)
    return onehot_logprob(mean, action, cfg.unimix)


# --------------------------------------------------------------------------------------
# encoders / decoders                                                 networks.py:293-585
# --------------------------------------------------------------------------------------
def ch_layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """ImgChLayerNorm (networks.py:801-810): LN over C per pixel on NCHW."""
    return layer_norm(x.permute(0, 2, 3, 1), w, b).permute(0, 3, 1, 2)


def conv_encoder(cfg: PathConfig, p, image: Tensor, pre: str = "encoder._cnn.") -> Tensor:
    """networks.py:486-496 (+771-798).  image f32 in [0,1], [B,T,64,64,3] -> [B,T,E]."""
    obs = image - 0.5
    x = obs.reshape((-1,) + tuple(obs.shape[-3:])).permute(0, 3, 1, 2)
    for i in range(4):
        x = F.pad(x, [1, 1, 1, 1])  # Conv2dSamePad: k4 s2 on even sizes -> pad 1 each side
        x = F.conv2d(x, p[f"{pre}layers.{3 * i}.weight"], None, 2)
        x = ch_layer_norm(x, p[f"{pre}layers.{3 * i + 1}.norm.weight"], p[f"{pre}layers.{3 * i + 1}.norm.bias"])
        x = F.silu(x)
    x = x.reshape(x.shape[0], -1)  # (C,H,W) flatten order
    return x.reshape(list(obs.shape[:-3]) + [x.shape[-1]])


def conv_decoder(cfg: PathConfig, p, feat: Tensor, pre: str = "heads.decoder._cnn.") -> Tensor:
    """networks.py:568-585.  feat [B,T,F] -> mean image [B,T,64,64,3]."""
    x = feat @ p[pre + "_linear_layer.weight"].t() + p[pre + "_linear_layer.bias"]
    c = x.shape[-1] // 16
    x = x.reshape(-1, 4, 4, c).permute(0, 3, 1, 2)  # un-flatten is (H,W,C)
    for i in range(3):
        x = F.conv_transpose2d(x, p[f"{pre}layers.{3 * i}.weight"], None, 2, padding=1)
        x = ch_layer_norm(x, p[f"{pre}layers.{3 * i + 1}.norm.weight"], p[f"{pre}layers.{3 * i + 1}.norm.bias"])
        x = F.silu(x)
    x = F.conv_transpose2d(x, p[pre + "layers.9.weight"], p[pre + "layers.9.bias"], 2, padding=1)
    mean = x.reshape(tuple(feat.shape[:-1]) + (3, 64, 64)).permute(0, 1, 3, 4, 2)
    return mean + 0.5


def mlp_encoder(cfg: PathConfig, p, obs: Dict[str, Tensor], pre: str = "encoder._mlp.") -> Tensor:
    """networks.py:353-355, 657-664: symlog(cat keys) -> trunk, no head."""
    x = torch.cat([obs[k] for k, _ in cfg.mlp_keys], -1)
    return mlp_trunk(p, pre, "Encoder", cfg.enc_mlp_layers, symlog(x))


def mlp_decoder_modes(cfg: PathConfig, p, feat: Tensor, pre: str = "heads.decoder._mlp.") -> Dict[str, Tensor]:
    """networks.py:665-674 with dist symlog_mse: per-key mean layers (raw, symlog space)."""
    h = mlp_trunk(p, pre, "Decoder", cfg.enc_mlp_layers, feat)
    return {
        k: h @ p[f"{pre}mean_layer.{k}.weight"].t() + p[f"{pre}mean_layer.{k}.bias"] for k, _ in cfg.mlp_keys
    }


def symlog_mse_logprob(mode: Tensor, value: Tensor) -> Tensor:
    """tools.SymlogDist.log_prob, dist 'mse', agg 'sum' (tools.py:558-572)."""
    d = (mode - symlog(value)) ** 2.0
    d = torch.where(d < 1e-8, 0, d)
    return -d.sum(list(range(d.dim()))[2:])


# --------------------------------------------------------------------------------------
# world-model loss                                                     models.py:108-171
# --------------------------------------------------------------------------------------
def preprocess(cfg: PathConfig, data: Dict) -> Dict[str, Tensor]:
    """models.py:174-190 (numpy/torch dict -> f32; image/255; cont = 1 - is_terminal)."""
    obs = {k: torch.as_tensor(v).to(torch.float32) for k, v in data.items()}
    obs["image"] = obs["image"] / 255.0
    if "discount" in obs:
        obs["discount"] = (obs["discount"] * cfg.discount).unsqueeze(-1)
    obs["cont"] = (1.0 - obs["is_terminal"]).unsqueeze(-1)
    return obs


def wm_forward(cfg: PathConfig, p, data: Dict, q_prior: Tensor, q_post: Tensor) -> Dict[str, Tensor]:
    """WorldModel._train up to the scalar loss (models.py:113-147).  Returns every tensor
    the parity tests look at."""
    obs = preprocess(cfg, data)
    # MultiEncoder.forward (networks.py:348-356): the CNN's and the MLP's outputs side by side
    parts = []
    if cfg.encoder in ("cnn", "both"):
        parts.append(conv_encoder(cfg, p, obs["image"]))
    if cfg.encoder in ("mlp", "both"):
        parts.append(mlp_encoder(cfg, p, obs))
    embed = parts[0] if len(parts) == 1 else torch.cat(parts, -1)
    action = obs["action"].clone()  # obs_step zeroes prev_action at is_first rows in place
    post, prior = observe(cfg, p, embed, action, obs["is_first"], q_prior, q_post)
    kl, kl_value, dyn, rep = kl_loss(cfg, post["logit"], prior["logit"])
    feat = get_feat(cfg, post)
    losses = {}
    out = {}
    # MultiDecoder.forward (networks.py:424-438): image head and every vector key on the same feat
    if cfg.encoder in ("cnn", "both"):
        recon = conv_decoder(cfg, p, feat)
        out["recon"] = recon
        losses["image"] = ((recon - obs["image"]) ** 2).sum([2, 3, 4])  # -MSEDist.log_prob, tools.py:531-540
    if cfg.encoder in ("mlp", "both"):
        modes = mlp_decoder_modes(cfg, p, feat)
        out["recon_modes"] = modes
        for k, _ in cfg.mlp_keys:
            losses[k] = -symlog_mse_logprob(modes[k], obs[k])
    r_logits = head_logits(p, "heads.reward.", "Reward", cfg.reward_layers, feat)
    losses["reward"] = -disc_logprob(r_logits, obs["reward"])
    c_logit = head_logits(p, "heads.cont.", "Cont", cfg.cont_layers, feat)
    losses["cont"] = -bernoulli_logprob(c_logit, obs["cont"])
    model_loss = sum(losses.values()) + kl
    out.update(
        embed=embed,
        post=post,
        prior=prior,
        kl=kl_value,
        dyn_loss=dyn,
        rep_loss=rep,
        feat=feat,
        reward_logits=r_logits,
        cont_logit=c_logit,
        losses=losses,
        model_loss=torch.mean(model_loss),
        prior_ent=onehot_entropy(prior["logit"], cfg.unimix),
        post_ent=onehot_entropy(post["logit"], cfg.unimix),
    )
    return out


# --------------------------------------------------------------------------------------
# imagination and behaviour losses                                     models.py:327-689
# --------------------------------------------------------------------------------------
def imagine(cfg: PathConfig, p, start: Dict[str, Tensor], act_noise: Tensor, q_prior: Tensor,
            teacher: Optional[Dict[str, Tensor]] = None):
    """ImagBehavior._imagine (models.py:448-548).

    start {[B,T,...]} (detached posterior); act_noise [H,N,A]; q_prior [H,N,S,D].
    Returns feats [H,N,F] (detached), states {[H,N,...]}, actions [H,N,A].
    """
    state = {k: v.reshape([-1] + list(v.shape[2:])) for k, v in start.items()}
    feats, succs, actions = [], [], []
    first = state
    for t in range(cfg.horizon):
        if teacher is not None and t > 0:
            state = {k: v[t] for k, v in teacher.items()}
        feat = get_feat(cfg, state).detach()
        a = actor_sample(cfg, p, feat, act_noise[t])
        succ = img_step(cfg, p, state["stoch"], state["deter"], a, q_prior[t])
        feats.append(feat)
        actions.append(a)
        succs.append(succ)
        state = succ
    states = {k: torch.stack([first[k]] + [s[k] for s in succs[:-1]], 0) for k in first}
    return torch.stack(feats, 0), states, torch.stack(actions, 0)


def video_pred(cfg: PathConfig, p, data: Dict, q_prior: Tensor, q_post: Tensor, q_open: Tensor) -> Tensor:
    """WorldModel.video_pred (models.py:192-213): posterior reconstruction of the first 5 steps of the first 6
    sequences, open-loop prior rollout (RSSM.imagine_with_action, networks.py:145-152) of the rest, stacked
    with the truth and the error along the image height.  q_prior/q_post [5,Bv,S,D], q_open [T-5,Bv,S,D].
    Returns [Bv, T, 3*64, 64, 3]."""
    obs = preprocess(cfg, data)
    embed = conv_encoder(cfg, p, obs["image"])
    states, _ = observe(cfg, p, embed[:6, :5], obs["action"][:6, :5], obs["is_first"][:6, :5], q_prior, q_post)
    recon = conv_decoder(cfg, p, get_feat(cfg, states))[:6]
    cur = {k: v[:, -1] for k, v in states.items()}
    act = obs["action"][:6, 5:]
    outs = []
    for t in range(act.shape[1]):
        cur = img_step(cfg, p, cur["stoch"], cur["deter"], act[:, t], q_open[t])
        outs.append(cur)
    prior = {k: torch.stack([o[k] for o in outs], 1) for k in outs[0]}
    openl = conv_decoder(cfg, p, get_feat(cfg, prior))
    model = torch.cat([recon[:, :5], openl], 1)
    truth = obs["image"][:6]
    error = (model - truth + 1.0) / 2.0
    return torch.cat([truth, model, error], 2)


def lambda_return(reward: Tensor, value: Tensor, disc: Tensor, lam: float) -> Tensor:
    """models.py:627-634 + tools.lambda_return (tools.py:702-728).

    reward, value, disc [H,N,1] (full horizon).  R_t = r_{t+1} + d_{t+1}((1-lam) v_{t+1} + lam R_{t+1}),
    R_{H-1} := v_{H-1}; returns [H-1,N,1].
    """
    h = reward.shape[0]
    inputs = reward[1:] + disc[1:] * value[1:] * (1 - lam)
    agg = value[-1]
    outs = [None] * (h - 1)
    for t in reversed(range(h - 1)):
        agg = inputs[t] + disc[1:][t] * lam * agg
        outs[t] = agg
    return torch.stack(outs, 0)


def quantile_05_95(x: Tensor) -> Tensor:
    return torch.quantile(torch.flatten(x.detach()), torch.tensor([0.05, 0.95], dtype=x.dtype))


def behavior_forward(cfg: PathConfig, p, start: Dict[str, Tensor], act_noise: Tensor, q_prior: Tensor,
                     ema_vals: Tensor, reward_fn=None) -> Dict[str, Tensor]:
    """ImagBehavior._train up to the two scalar losses (models.py:337-429, 620-681).

    `ema_vals` [2] is updated in place (models.py:23).  Slow-critic params live under
    `_slow_value.`; the caller applies the EMA update (models.py:683-689) beforehand.
    reward_fn(feat, states, actions) -> [H,N,1]: the `objective` argument of models.py:327-331 (default: the world
    model's reward head on the imagined states, dreamer.py:196-199).  As in the reference, `feat` is DETACHED
    (models.py:513-517 returns get_feat(state).detach()); gradients reach the dynamics through `states` / `actions` only.
    """
    feats, states, actions = imagine(cfg, p, start, act_noise, q_prior)
    sfeat = get_feat(cfg, states)
    if reward_fn is not None:
        reward = reward_fn(feats, states, actions)
    else:
        reward = disc_mode(head_logits(p, "heads.reward.", "Reward", cfg.reward_layers, sfeat))
    ent = actor_entropy(cfg, p, feats)
    disc = cfg.discount * torch.sigmoid(head_logits(p, "heads.cont.", "Cont", cfg.cont_layers, sfeat))
    v_logits = head_logits(p, "value.", "Value", cfg.critic_layers, feats)
    value = disc_mode(v_logits)
    target = lambda_return(reward, value, disc, cfg.discount_lambda)
    weights = torch.cumprod(torch.cat([torch.ones_like(disc[:1]), disc[:-1]], 0), 0).detach()
    base = value[:-1]
    # models.py:654-661 RewardEMA
    qv = quantile_05_95(target)
    ema_vals[:] = cfg.ema_alpha * qv + (1 - cfg.ema_alpha) * ema_vals
    scale = torch.clip(ema_vals[1] - ema_vals[0], min=1.0).detach()
    offset = ema_vals[0].detach()
    normed_target = (target - offset) / scale
    normed_base = (base - offset) / scale
    adv = normed_target - normed_base
    if cfg.imag_gradient == "dynamics":
        actor_target = adv
    elif cfg.imag_gradient == "reinforce":
        actor_target = actor_logprob(cfg, p, feats, actions)[:-1][:, :, None] * (target - value[:-1]).detach()
    elif cfg.imag_gradient == "both":  # models.py:670-676 (the mixed-in target is the raw, un-normalised return)
        actor_target = actor_logprob(cfg, p, feats, actions)[:-1][:, :, None] * (target - value[:-1]).detach()
        actor_target = cfg.imag_gradient_mix * target + (1 - cfg.imag_gradient_mix) * actor_target
    else:
        raise NotImplementedError(cfg.imag_gradient)
    actor_loss = -weights[:-1] * actor_target
    actor_loss = actor_loss - cfg.actor_entropy * ent[:-1, ..., None]
    actor_loss = torch.mean(actor_loss)
    # critic (models.py:419-429)
    vl = head_logits(p, "value.", "Value", cfg.critic_layers, feats[:-1].detach())
    value_loss = -disc_logprob(vl, target.detach().squeeze(-1))
    slow = disc_mode(head_logits(p, "_slow_value.", "Value", cfg.critic_layers, feats[:-1].detach()))
    value_loss = value_loss - disc_logprob(vl, slow.detach().squeeze(-1))
    value_loss = torch.mean(weights[:-1] * value_loss[:, :, None])
    return dict(
        feats=feats, states=states, actions=actions, reward=reward, actor_ent=ent, discount=disc,
        value=value, target=target, weights=weights, normed_target=normed_target,
        actor_loss=actor_loss, value_loss=value_loss, value_logits=v_logits,
    )


# --------------------------------------------------------------------------------------
# Plan2Explore                                                      exploration.py:40-135
# --------------------------------------------------------------------------------------
@dataclass
class P2EConfig:
    disag_models: int = 10  # configs.yaml:122
    disag_layers: int = 4  # configs.yaml:124
    disag_units: int = 400  # configs.yaml:125
    disag_target: str = "stoch"  # configs.yaml:120
    disag_offset: int = 1  # configs.yaml:123
    disag_log: bool = True  # configs.yaml:121
    disag_action_cond: bool = False  # configs.yaml:126
    expl_intr_scale: float = 1.0  # configs.yaml:119
    expl_extr_scale: float = 0.0  # configs.yaml:118


# networks.MLP with its defaults dist="normal", std=1.0, min_std=0.1, max_std=1.0 (networks.py:597-600, 614):
# Normal(tanh(mean), (max_std - min_std) * sigmoid(1.0 + 2.0) + min_std)   (networks.py:693-696)
P2E_STD = 0.9 / (1.0 + math.exp(-3.0)) + 0.1


def p2e_member_mean(c: P2EConfig, pp, i: int, x: Tensor) -> Tensor:
    """`head(inputs)` of ensemble member i: trunk named "NoName" (networks.py:607) + mean_layer; the distribution's
    mean == mode (tools.py:581-587, absmax None) is tanh(mean_layer(h))."""
    pre = f"_networks.{i}."
    h = mlp_trunk(pp, pre, "NoName", c.disag_layers, x)
    return torch.tanh(h @ pp[pre + "mean_layer.weight"].t() + pp[pre + "mean_layer.bias"])


def p2e_ensemble_loss(c: P2EConfig, pp, inputs: Tensor, targets: Tensor) -> Tensor:
    """Plan2Explore._train_ensemble (exploration.py:123-133): inputs [B,T,F(+A)], targets [B,T,W] ->
    -mean_i mean_{b,t} log N(target_{t+offset}; mode_i(input_t), P2E_STD) summed over W."""
    if c.disag_offset:
        targets, inputs = targets[:, c.disag_offset:], inputs[:, :-c.disag_offset]
    targets, inputs = targets.detach(), inputs.detach()
    likes = []
    for i in range(c.disag_models):
        mu = p2e_member_mean(c, pp, i, inputs)
        lp = -((targets - mu) ** 2) / (2 * P2E_STD**2) - math.log(P2E_STD) - math.log(math.sqrt(2 * math.pi))
        likes.append(lp.sum(-1).mean()[None])
    return -torch.mean(torch.cat(likes, 0))


def p2e_intrinsic_reward(c: P2EConfig, pp, feat: Tensor, action: Tensor, extr: Optional[Tensor] = None) -> Tensor:
    """Plan2Explore._intrinsic_reward (exploration.py:108-121): [..,F] -> [..,1]."""
    x = torch.cat([feat, action], -1) if c.disag_action_cond else feat
    preds = torch.stack([p2e_member_mean(c, pp, i, x) for i in range(c.disag_models)], 0)
    disag = torch.mean(torch.std(preds, 0), -1)[..., None]
    if c.disag_log:
        disag = torch.log(disag)
    reward = c.expl_intr_scale * disag
    if c.expl_extr_scale:
        reward = reward + c.expl_extr_scale * extr
    return reward


# --------------------------------------------------------------------------------------
# optimizer step                                                        tools.py:760-776
# --------------------------------------------------------------------------------------
def clip_and_adam(params, grads, state, lr: float, eps: float, clip: float):
    """clip_grad_norm_(params, clip) then torch.optim.Adam(lr, eps) step, betas (0.9, 0.999).

    params/grads: lists of tensors; state: dict with 'step', 'm', 'v' lists.  In-place.
    Returns the pre-clip global grad norm (the reference's `{name}_grad_norm` metric).
    """
    norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(clip / (norm + 1e-6), max=1.0)
    state["step"] += 1
    t = state["step"]
    b1, b2 = 0.9, 0.999
    for i, (w, g) in enumerate(zip(params, grads)):
        g = g * coef
        state["m"][i].mul_(b1).add_(g, alpha=1 - b1)
        state["v"][i].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1**t
        bc2 = 1 - b2**t
        denom = (state["v"][i].sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(state["m"][i], denom, value=-lr / bc1)
    return norm
