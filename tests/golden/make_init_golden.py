#!/usr/bin/env python3
"""Init fixture: the parameters the REFERENCE's constructors produce under
torch.manual_seed(0) -- tools.weight_init / uniform_weight_init (tools.py:890-946) in the reference's module
construction order -- as per-tensor checksums + the first 4 values.  Build container only.

    python tests/golden/make_init_golden.py     # writes tests/golden/init.npz
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import common  # noqa: E402
from tests.golden import make_golden as MG  # noqa: E402


def main():
    tools, networks, models = MG.import_reference()
    out = {}
    for name in ("tiny", "tiny_onehot", "tiny_proprio", "cfg2"):
        s = common.SHAPES[name]
        blocks = ["dmc_proprio"] if s["encoder"] == "mlp" else ["dmc_vision"]
        ov = dict(device="cpu", compile=False, num_actions=s["A"], dyn_stoch=s["stoch"], dyn_discrete=s["discrete"],
                  dyn_deter=s["deter"], dyn_hidden=s["hidden"], units=s["units"], batch_size=s["B"],
                  batch_length=s["T"], imag_horizon=s["H"], imag_gradient=s["imag_gradient"],
                  encoder=dict(cnn_depth=s["cnn_depth"]), decoder=dict(cnn_depth=s["cnn_depth"]),
                  causal_world_model=False)
        if s["actor_dist"] == "onehot":
            ov["actor"] = dict(dist="onehot", std="none")
        if s["encoder"] == "mlp":
            ov["encoder"].update(mlp_units=s["enc_mlp_units"], mlp_layers=s["enc_mlp_layers"])
            ov["decoder"].update(mlp_units=s["enc_mlp_units"], mlp_layers=s["enc_mlp_layers"])
        cfg = MG.load_config(blocks, ov)
        spaces = {}
        if s["encoder"] == "mlp":
            for k, w in common.PROPRIO_KEYS:
                spaces[k] = MG.Space((w,))
        spaces.update(image=MG.Space((64, 64, 3)), is_first=MG.Space((1,)), is_terminal=MG.Space((1,)))
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            wm = models.WorldModel(MG.ObsSpace(spaces), None, 0, cfg)
            beh = models.ImagBehavior(cfg, wm)
        sd = dict(wm.state_dict())
        sd.update({k: v for k, v in beh.state_dict().items() if not k.startswith("_world_model.")})
        for k, v in sd.items():
            a = v.detach().cpu().numpy().astype(np.float64).reshape(-1)
            out[f"{name}/sum/{k}"] = common.checksum(a)
            out[f"{name}/head/{k}"] = a[:4].astype(np.float32)
        print(f"[init] {name}: {len(sd)} tensors")
    dst = os.path.join(HERE, "init.npz")
    np.savez_compressed(dst, **out)
    print(f"[init] wrote {dst} ({os.path.getsize(dst) / 1e3:.1f} kB)")


if __name__ == "__main__":
    main()
