"""Deterministic inputs shared by the golden-vector generator and the tests.

Nothing here imports the reference.  Everything is derived from numpy RandomState streams
so that the generator (build container, reference importable) and the tests (GPU box, no
reference) regenerate bit-identical weights, batches and sampling noise from a seed.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import numpy as np

# ---------------------------------------------------------------------------------------
# shape configs
# ---------------------------------------------------------------------------------------
import os as _os
import sys as _sys

_PKG = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), "dreamerv3-torch_amd")
if _PKG not in _sys.path:
    _sys.path.insert(0, _PKG)
from dv3hip.shapes import PROPRIO_KEYS, SHAPES  # noqa: E402,F401  (one table for the product, the fixtures and the tests)


def path_config(name: str):
    """oracle.PathConfig for a named shape config."""
    from oracle.dv3_oracle import PathConfig

    s = SHAPES[name]
    kw = dict(
        stoch=s["stoch"], discrete=s["discrete"], deter=s["deter"], hidden=s["hidden"], units=s["units"],
        num_actions=s["A"], cnn_depth=s["cnn_depth"], actor_dist=s["actor_dist"],
        imag_gradient=s["imag_gradient"], horizon=s["H"], encoder=s["encoder"],
        actor_layers=s.get("actor_layers", 2), reward_layers=s.get("reward_layers", 2),
        cont_layers=s.get("cont_layers", 2), critic_layers=s.get("critic_layers", 2),
        imag_gradient_mix=s.get("imag_gradient_mix", 0.0),
    )
    if s["encoder"] in ("mlp", "both"):
        kw.update(mlp_keys=PROPRIO_KEYS, enc_mlp_units=s["enc_mlp_units"], enc_mlp_layers=s["enc_mlp_layers"])
    return PathConfig(**kw)


# ---------------------------------------------------------------------------------------
# parameter shapes by reference state_dict name
# ---------------------------------------------------------------------------------------
def param_shapes(name: str) -> Dict[str, Tuple[int, ...]]:
    s = SHAPES[name]
    S, D, De, Hd, U, A, d = s["stoch"], s["discrete"], s["deter"], s["hidden"], s["units"], s["A"], s["cnn_depth"]
    SD, F = S * D, S * D + De
    sh: Dict[str, Tuple[int, ...]] = {}

    def ln(prefix, n):
        sh[prefix + ".weight"] = (n,)
        sh[prefix + ".bias"] = (n,)

    def mlp(prefix, nm, layers, inp, units):
        for i in range(layers):
            sh[f"{prefix}layers.{nm}_linear{i}.weight"] = (units, inp if i == 0 else units)
            ln(f"{prefix}layers.{nm}_norm{i}", units)

    E = 0
    if s["encoder"] in ("cnn", "both"):
        E += d * 8 * 16
        cin = 3
        for i in range(4):
            cout = d * 2**i
            sh[f"encoder._cnn.layers.{3 * i}.weight"] = (cout, cin, 4, 4)
            ln(f"encoder._cnn.layers.{3 * i + 1}.norm", cout)
            cin = cout
    if s["encoder"] in ("mlp", "both"):
        Em = s["enc_mlp_units"]
        E += Em
        mlp("encoder._mlp.", "Encoder", s["enc_mlp_layers"], sum(w for _, w in PROPRIO_KEYS), Em)
    sh["dynamics.W"] = (1, De)
    sh["dynamics._img_in_layers.0.weight"] = (Hd, SD + A)
    ln("dynamics._img_in_layers.1", Hd)
    sh["dynamics._cell.layers.GRU_linear.weight"] = (3 * De, Hd + De)
    ln("dynamics._cell.layers.GRU_norm", 3 * De)
    sh["dynamics._img_out_layers.0.weight"] = (Hd, De)
    ln("dynamics._img_out_layers.1", Hd)
    sh["dynamics._obs_out_layers.0.weight"] = (Hd, De + E)
    ln("dynamics._obs_out_layers.1", Hd)
    sh["dynamics._imgs_stat_layer.weight"] = (SD, Hd)
    sh["dynamics._imgs_stat_layer.bias"] = (SD,)
    sh["dynamics._obs_stat_layer.weight"] = (SD, Hd)
    sh["dynamics._obs_stat_layer.bias"] = (SD,)
    if s["encoder"] in ("cnn", "both"):
        Ec = d * 8 * 16
        sh["heads.decoder._cnn._linear_layer.weight"] = (Ec, F)
        sh["heads.decoder._cnn._linear_layer.bias"] = (Ec,)
        cin = d * 8
        for i in range(3):
            sh[f"heads.decoder._cnn.layers.{3 * i}.weight"] = (cin, cin // 2, 4, 4)
            ln(f"heads.decoder._cnn.layers.{3 * i + 1}.norm", cin // 2)
            cin //= 2
        sh["heads.decoder._cnn.layers.9.weight"] = (cin, 3, 4, 4)
        sh["heads.decoder._cnn.layers.9.bias"] = (3,)
    if s["encoder"] in ("mlp", "both"):
        Em = s["enc_mlp_units"]
        mlp("heads.decoder._mlp.", "Decoder", s["enc_mlp_layers"], F, Em)
        for k, w in PROPRIO_KEYS:
            sh[f"heads.decoder._mlp.mean_layer.{k}.weight"] = (w, Em)
            sh[f"heads.decoder._mlp.mean_layer.{k}.bias"] = (w,)
    for pre, nm, out in (("heads.reward.", "Reward", 255), ("heads.cont.", "Cont", 1)):
        mlp(pre, nm, s.get(nm.lower() + "_layers", 2), F, U)
        sh[pre + "mean_layer.weight"] = (out, U)
        sh[pre + "mean_layer.bias"] = (out,)
    mlp("actor.", "Actor", s.get("actor_layers", 2), F, U)
    sh["actor.mean_layer.weight"] = (A, U)
    sh["actor.mean_layer.bias"] = (A,)
    if s["actor_dist"] == "normal":
        sh["actor.std_layer.weight"] = (A, U)
        sh["actor.std_layer.bias"] = (A,)
    for pre in ("value.", "_slow_value."):
        mlp(pre, "Value", s.get("critic_layers", 2), F, U)
        sh[pre + "mean_layer.weight"] = (255, U)
        sh[pre + "mean_layer.bias"] = (255,)
    return sh


def p2e_param_shapes(name: str) -> Dict[str, Tuple[int, ...]]:
    """Plan2Explore's own parameters by state_dict name (exploration.py:66-73: `_networks.<i>` = networks.MLP with the
    default name "NoName"; its `_behavior.actor` / `_behavior.value` / `_behavior._slow_value` have the shapes of the
    task behaviour's)."""
    s = SHAPES[name]
    c = s["p2e"]
    SD, F = s["stoch"] * s["discrete"], s["stoch"] * s["discrete"] + s["deter"]
    inp = F + (s["A"] if c["disag_action_cond"] else 0)
    out = {"stoch": SD, "deter": s["deter"], "embed": s["cnn_depth"] * 8 * 16}[c["disag_target"]]
    U = c["disag_units"]
    sh: Dict[str, Tuple[int, ...]] = {}
    for i in range(c["disag_models"]):
        for j in range(c["disag_layers"]):
            sh[f"_networks.{i}.layers.NoName_linear{j}.weight"] = (U, inp if j == 0 else U)
            sh[f"_networks.{i}.layers.NoName_norm{j}.weight"] = (U,)
            sh[f"_networks.{i}.layers.NoName_norm{j}.bias"] = (U,)
        sh[f"_networks.{i}.mean_layer.weight"] = (out, U)
        sh[f"_networks.{i}.mean_layer.bias"] = (out,)
    for k, v in param_shapes(name).items():
        if k.split(".")[0] in ("actor", "value", "_slow_value"):
            sh["_behavior." + k] = v
    return sh


def make_p2e_weights(name: str, seed: int = 3) -> Dict[str, np.ndarray]:
    """Deterministic weights of the Plan2Explore module (same scheme as make_weights, its own seed so that the
    exploration actor / critic differ from the task behaviour's)."""
    out = {}
    for k, shp in p2e_param_shapes(name).items():
        rs = np.random.RandomState((zlib.crc32(k.encode()) + 7919 * seed) & 0x7FFFFFFF)
        if len(shp) == 1:
            w = (1.0 + 0.1 * rs.randn(*shp)) if k.endswith(".weight") and "norm" in k else 0.1 * rs.randn(*shp)
        else:
            w = rs.randn(*shp) * np.sqrt(2.0 / (shp[0] + shp[1]))
            if "value" in k and "mean_layer" in k:
                w *= 0.3
        out[k] = w.astype(np.float32)
    return out


def make_weights(name: str, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic fp32 weights, one numpy stream per parameter name.

    Not the reference's init scheme (that one needs torch's RNG); any weights pin parity.
    Scales follow tools.weight_init (tools.py:890-917) so activations sit in a realistic
    regime; LN affine and biases are perturbed so that no term is trivially 0 or 1.
    """
    out = {}
    for k, shp in param_shapes(name).items():
        rs = np.random.RandomState((zlib.crc32(k.encode()) + 7919 * seed) & 0x7FFFFFFF)
        if k == "dynamics.W":
            w = 0.5 * rs.randn(*shp)
        elif len(shp) == 1:
            if k.endswith(".weight"):  # every 1-D ".weight" on the path is a LayerNorm scale
                w = 1.0 + 0.1 * rs.randn(*shp)
            else:
                w = 0.1 * rs.randn(*shp)
        else:
            if len(shp) == 4:
                # Conv2d weight [out,in,kh,kw]; ConvTranspose2d weight [in,out,kh,kw]; fan avg is symmetric
                fan = (shp[0] + shp[1]) * shp[2] * shp[3] / 2.0
            else:
                fan = (shp[0] + shp[1]) / 2.0
            w = rs.randn(*shp) * np.sqrt(1.0 / fan)
            if "mean_layer" in k and ("reward" in k or "value" in k):
                w *= 0.3  # reference zero-inits these (outscale 0.0); keep them small but non-zero
        out[k] = w.astype(np.float32)
    return out


def make_batch(name: str, seed: int = 0, extra_first: bool = True) -> Dict[str, np.ndarray]:
    """Synthetic replay minibatch per SURVEY.md §8d."""
    s = SHAPES[name]
    B, T, A = s["B"], s["T"], s["A"]
    rs = np.random.RandomState(seed)
    data = {}
    data["image"] = rs.randint(0, 256, size=(B, T, 64, 64, 3)).astype(np.uint8)
    if s["actor_dist"] == "onehot":
        idx = rs.randint(0, A, size=(B, T))
        data["action"] = np.eye(A, dtype=np.float32)[idx]
    else:
        data["action"] = rs.uniform(-1, 1, size=(B, T, A)).astype(np.float32)
    data["reward"] = rs.randn(B, T).astype(np.float32)
    data["discount"] = np.ones((B, T), np.float32)
    first = np.zeros((B, T), bool)
    first[:, 0] = True
    if extra_first:
        for b in range(B):
            if b % 2 == 0 and T > 2:
                first[b, rs.randint(1, T)] = True
    data["is_first"] = first
    term = np.zeros((B, T), bool)
    term[B // 2, T - 1] = True  # one terminal so the cont head sees both classes
    data["is_terminal"] = term
    if s["encoder"] in ("mlp", "both"):
        for k, w in PROPRIO_KEYS:
            data[k] = rs.randn(B, T, w).astype(np.float32)
    return data


def make_noise(name: str, seed: int = 0) -> Dict[str, np.ndarray]:
    """Every random draw of one update, as explicit fp32 arrays.

    observe: q_prior, q_post [T,B,S,D] ~ Exp(1);  imagine: act [H,N,A] (N(0,1) for the normal
    actor, Exp(1) for the onehot actor), q_img [H,N,S,D] ~ Exp(1).
    """
    s = SHAPES[name]
    B, T, H, S, D, A = s["B"], s["T"], s["H"], s["stoch"], s["discrete"], s["A"]
    N = B * T
    rs = np.random.RandomState(1000 + seed)
    out = {
        "q_prior": rs.exponential(size=(T, B, S, D)).astype(np.float32),
        "q_post": rs.exponential(size=(T, B, S, D)).astype(np.float32),
        "q_img": rs.exponential(size=(H, N, S, D)).astype(np.float32),
    }
    if s["actor_dist"] == "onehot":
        out["act"] = rs.exponential(size=(H, N, A)).astype(np.float32)
    else:
        out["act"] = rs.randn(H, N, A).astype(np.float32)
    # guard against an exact zero (division by q)
    for k in ("q_prior", "q_post", "q_img"):
        np.maximum(out[k], 1e-20, out=out[k])
    return out


def observe_tape(noise: Dict[str, np.ndarray]) -> List[np.ndarray]:
    """Noise in the order the reference's observe scan draws it."""
    tape = []
    for t in range(noise["q_prior"].shape[0]):
        tape.append(noise["q_prior"][t])
        tape.append(noise["q_post"][t])
    return tape


def imagine_tape(noise: Dict[str, np.ndarray]) -> List[np.ndarray]:
    tape = []
    for t in range(noise["q_img"].shape[0]):
        tape.append(noise["act"][t])
        tape.append(noise["q_img"][t])
    return tape


def make_video_noise(name: str, seed: int = 0) -> Dict[str, np.ndarray]:
    """Draws of WorldModel.video_pred (models.py:192-213): 5 observed steps of the first 6 sequences
    (q_prior, q_post [5,Bv,S,D]) and the open-loop rest (q_open [T-5,Bv,S,D]), all ~ Exp(1)."""
    s = SHAPES[name]
    bv, T, S, D = min(6, s["B"]), s["T"], s["stoch"], s["discrete"]
    rs = np.random.RandomState(2000 + seed)
    out = {"q_prior": rs.exponential(size=(5, bv, S, D)).astype(np.float32),
           "q_post": rs.exponential(size=(5, bv, S, D)).astype(np.float32),
           "q_open": rs.exponential(size=(T - 5, bv, S, D)).astype(np.float32)}
    for k in out:
        np.maximum(out[k], 1e-20, out=out[k])
    return out


def video_tape(noise: Dict[str, np.ndarray]) -> List[np.ndarray]:
    """Order in which the reference's video_pred draws: observe (prior, post per step), then the rollout."""
    tape = observe_tape(noise)
    tape.extend(noise["q_open"][t] for t in range(noise["q_open"].shape[0]))
    return tape


REPLAY_LENGTH, REPLAY_BATCH = 16, 4


def make_episodes():
    """Synthetic episode store in the reference's in-memory format (tools.load_episodes: ordered dict
    name -> dict of per-step arrays).  Lengths include a 1-step episode (skipped by the sampler), episodes
    shorter than the sequence length (joined) and long ones.  reward = 1000*episode + index is a unique id."""
    import collections

    rs = np.random.RandomState(77)
    eps = collections.OrderedDict()
    for e, n in enumerate((1, 7, 30, 50, 3, 21)):
        first = np.zeros(n, bool)
        first[0] = True
        term = np.zeros(n, bool)
        term[-1] = e % 2 == 0
        eps[f"ep{e}"] = {
            "image": rs.randint(0, 256, (n, 4, 4, 3)).astype(np.uint8),
            "action": rs.uniform(-1, 1, (n, 3)).astype(np.float32),
            "reward": (1000.0 * e + np.arange(n)).astype(np.float32),
            "discount": np.ones(n, np.float32),
            "is_first": first, "is_terminal": term,
            "logprob": rs.randn(n).astype(np.float32),
            "log_entropy": rs.randn(n).astype(np.float32),  # "log_" keys are dropped by the sampler
        }
    return eps


def checksum(x: np.ndarray) -> np.ndarray:
    """(sum, abs-sum, max-abs) in float64 -- cheap whole-tensor pin for large outputs."""
    x = np.asarray(x, np.float64)
    return np.array([x.sum(), np.abs(x).sum(), np.abs(x).max()], np.float64)
