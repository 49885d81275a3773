"""BASELINE.json benchmark / parity shape configs of the hot path, and the reference-style
config / observation-space objects built from them.  Product-side module: bench.py and the tools build their models
from here; tests/golden/common.py re-exports SHAPES so that fixtures and tests use the same table.  Nothing here
imports tests/ or oracle/."""
from __future__ import annotations

import argparse
import os
from typing import Tuple

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = {
    # name: dict(stoch, discrete, deter, hidden, units, A, cnn_depth, B, T, H, actor_dist, imag_gradient)
    "tiny": dict(stoch=4, discrete=4, deter=16, hidden=16, units=16, A=3, cnn_depth=2, B=3, T=6, H=4,
                 actor_dist="normal", imag_gradient="dynamics", encoder="cnn"),
    "tiny_onehot": dict(stoch=4, discrete=4, deter=16, hidden=16, units=16, A=5, cnn_depth=2, B=3, T=6, H=4,
                        actor_dist="onehot", imag_gradient="reinforce", encoder="cnn"),
    # 'both' actor gradient (models.py:670-676) with the continuous actor: the log-prob term sees the rsampled action
    "tiny_both": dict(stoch=4, discrete=4, deter=16, hidden=16, units=16, A=3, cnn_depth=2, B=3, T=6, H=4,
                      actor_dist="normal", imag_gradient="both", imag_gradient_mix=0.3, encoder="cnn"),
    "tiny_proprio": dict(stoch=4, discrete=4, deter=16, hidden=16, units=16, A=3, cnn_depth=2, B=3, T=6, H=4,
                         actor_dist="normal", imag_gradient="dynamics", encoder="mlp",
                         enc_mlp_units=32, enc_mlp_layers=2),
    # image AND vector observations together (networks.MultiEncoder / MultiDecoder, networks.py:293-445; the shipped
    # `minecraft` block, configs.yaml:206-207): embed = [cnn | mlp], the decoder predicts the image and every vector key
    "tiny_mixed": dict(stoch=4, discrete=4, deter=16, hidden=16, units=16, A=3, cnn_depth=2, B=3, T=6, H=4,
                       actor_dist="normal", imag_gradient="dynamics", encoder="both",
                       enc_mlp_units=24, enc_mlp_layers=2),
    # Plan2Explore (exploration.py:40-135) on the tiny model: defaults (stoch target, log disagreement, no action
    # conditioning, no extrinsic term) and the action-conditioned variant with an extrinsic term
    "tiny_p2e": dict(stoch=4, discrete=4, deter=16, hidden=16, units=16, A=3, cnn_depth=2, B=3, T=6, H=4,
                     actor_dist="normal", imag_gradient="dynamics", encoder="cnn",
                     p2e=dict(disag_models=3, disag_layers=2, disag_units=16, disag_target="stoch", disag_offset=1,
                              disag_log=True, disag_action_cond=False, expl_intr_scale=1.0, expl_extr_scale=0.0)),
    "tiny_p2e_ac": dict(stoch=4, discrete=4, deter=16, hidden=16, units=16, A=3, cnn_depth=2, B=3, T=6, H=4,
                        actor_dist="normal", imag_gradient="dynamics", encoder="cnn",
                        p2e=dict(disag_models=4, disag_layers=3, disag_units=24, disag_target="deter", disag_offset=1,
                                 disag_log=False, disag_action_cond=True, expl_intr_scale=0.7, expl_extr_scale=0.5)),
    "cfg1": dict(stoch=32, discrete=32, deter=512, hidden=512, units=512, A=6, cnn_depth=32, B=16, T=64, H=15,
                 actor_dist="normal", imag_gradient="dynamics", encoder="mlp",
                 enc_mlp_units=1024, enc_mlp_layers=5),
    "cfg2": dict(stoch=32, discrete=32, deter=512, hidden=512, units=512, A=6, cnn_depth=32, B=16, T=64, H=15,
                 actor_dist="normal", imag_gradient="dynamics", encoder="cnn"),
    "cfg3": dict(stoch=32, discrete=32, deter=1024, hidden=512, units=512, A=18, cnn_depth=32, B=32, T=64, H=15,
                 actor_dist="onehot", imag_gradient="reinforce", encoder="cnn"),
    # BASELINE cfg 4: dmc_vision with the crafter-size model (configs.yaml:165-169 sizes), batch 64 x 64
    "cfg4": dict(stoch=32, discrete=32, deter=4096, hidden=1024, units=1024, A=6, cnn_depth=96, B=64, T=64, H=15,
                 actor_dist="normal", imag_gradient="dynamics", encoder="cnn"),
    # BASELINE cfg 5: crafter block (configs.yaml:158-174: 5-layer actor / reward / cont heads, one-hot actor,
    # reinforce; its `value: {layers: 5}` key is not read by models.py, the critic keeps 2 layers) with
    # dyn_deter 2048, sequence length 256.  BASELINE.json states the batch as 128 x 256 at DP = 8: B here is the
    # per-GPU shard, 16 sequences (8 ranks x 16 = 128; bench.py scales weakly, so --gpus 8 is exactly that global
    # batch).  One GPU cannot hold the activations of all 128 x 256 frames at fp32 without recomputation (conv stacks
    # ~200 GB + 15 x 32768 imagination rows through five 1024-wide heads ~120 GB > 288 GB).
    "cfg5": dict(stoch=32, discrete=32, deter=2048, hidden=1024, units=1024, A=17, cnn_depth=96, B=16, T=256, H=15,
                 actor_dist="onehot", imag_gradient="reinforce", encoder="cnn", actor_layers=5, reward_layers=5,
                 cont_layers=5, global_B_dp8=128),
    # reduced-batch variants of the two (same layer widths: every kernel shape class of cfg 4 / cfg 5 at a size the
    # CPU oracle finishes in seconds)
    "cfg4_b4": dict(stoch=32, discrete=32, deter=4096, hidden=1024, units=1024, A=6, cnn_depth=96, B=4, T=8, H=5,
                    actor_dist="normal", imag_gradient="dynamics", encoder="cnn"),
    "cfg5_b4": dict(stoch=32, discrete=32, deter=2048, hidden=1024, units=1024, A=17, cnn_depth=96, B=4, T=8, H=5,
                    actor_dist="onehot", imag_gradient="reinforce", encoder="cnn", actor_layers=5, reward_layers=5,
                    cont_layers=5),
}
# walker_walk proprio keys, in the order the reference's obs_space dict would list them
PROPRIO_KEYS: Tuple[Tuple[str, int], ...] = (("orientations", 14), ("height", 1), ("velocity", 9))




class _Space:
    def __init__(self, shape):
        self.shape = shape


class _ObsSpace:
    def __init__(self, spaces):
        self.spaces = spaces


def make_config(name, device="cuda:0"):
    """argparse.Namespace with the reference's config keys (configs.yaml defaults + the block of the shape's task
    family + the BASELINE overrides of SURVEY.md Appendix B)."""
    import tools

    s = SHAPES[name]
    blocks = ["dmc_proprio"] if s["encoder"] == "mlp" else ["dmc_vision"]
    cfg = tools.load_config(os.path.join(PKG, "configs.yaml"), blocks)
    if s["encoder"] == "both":  # the vision block with the vector keys routed to the MLP as well (as `minecraft` does)
        keys = "|".join(k for k, _ in PROPRIO_KEYS)
        cfg["encoder"].update(mlp_keys=keys, cnn_keys="image")
        cfg["decoder"].update(mlp_keys=keys, cnn_keys="image")
    cfg.update(device=device, num_actions=s["A"], dyn_stoch=s["stoch"], dyn_discrete=s["discrete"],
               dyn_deter=s["deter"], dyn_hidden=s["hidden"], units=s["units"], batch_size=s["B"],
               batch_length=s["T"], imag_horizon=s["H"], imag_gradient=s["imag_gradient"],
               imag_gradient_mix=s.get("imag_gradient_mix", 0.0))
    cfg["encoder"]["cnn_depth"] = s["cnn_depth"]
    cfg["decoder"]["cnn_depth"] = s["cnn_depth"]
    if s["actor_dist"] == "onehot":
        cfg["actor"].update(dist="onehot", std="none")
    cfg["actor"]["layers"] = s.get("actor_layers", 2)
    cfg["critic"]["layers"] = s.get("critic_layers", 2)
    cfg["reward_head"]["layers"] = s.get("reward_layers", 2)
    cfg["cont_head"]["layers"] = s.get("cont_layers", 2)
    if s["encoder"] in ("mlp", "both"):
        for d in (cfg["encoder"], cfg["decoder"]):
            d.update(mlp_units=s["enc_mlp_units"], mlp_layers=s["enc_mlp_layers"])
    if "p2e" in s:
        cfg.update(expl_behavior="plan2explore", **s["p2e"])
    return argparse.Namespace(**cfg)


def obs_space(name):
    s = SHAPES[name]
    spaces = {}
    if s["encoder"] in ("mlp", "both"):
        for k, w in PROPRIO_KEYS:
            spaces[k] = _Space((w,))
    spaces["image"] = _Space((64, 64, 3))
    spaces["is_first"] = _Space((1,))
    spaces["is_terminal"] = _Space((1,))
    return _ObsSpace(spaces)


def synthetic_batch(name, seed=0):
    """Synthetic replay minibatch of SURVEY.md 8(d) as host numpy arrays: RandomState(seed); image u8, continuous
    action uniform(-1,1) / one-hot, reward randn, is_first[:,0] plus one extra reset on every other row."""
    import numpy as np

    s = SHAPES[name]
    rs = np.random.RandomState(seed)
    B, T, A = s["B"], s["T"], s["A"]
    data = {"image": rs.randint(0, 256, size=(B, T, 64, 64, 3)).astype(np.uint8)}
    if s["actor_dist"] == "onehot":
        data["action"] = np.eye(A, dtype=np.float32)[rs.randint(0, A, size=(B, T))]
    else:
        data["action"] = rs.uniform(-1, 1, size=(B, T, A)).astype(np.float32)
    data["reward"] = rs.randn(B, T).astype(np.float32)
    data["discount"] = np.ones((B, T), np.float32)
    first = np.zeros((B, T), np.float32)
    first[:, 0] = 1.0
    for b in range(0, B, 2):
        first[b, rs.randint(1, T)] = 1.0
    data["is_first"] = first
    data["is_terminal"] = np.zeros((B, T), np.float32)
    if s["encoder"] in ("mlp", "both"):
        for k, w in PROPRIO_KEYS:
            data[k] = rs.randn(B, T, w).astype(np.float32)
    return data
