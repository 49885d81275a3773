"""tools.weight_init / uniform_weight_init (tools.py:890-946).  Constructing the MI355X
models under torch.manual_seed(0) draws the SAME trunc_normal_ / uniform_ values, tensor by tensor, as the
reference's constructors do (module construction order included): tests/golden/init.npz holds per-tensor
checksums and the first values of the parameters the reference itself produced
(tests/golden/make_init_golden.py, build container).  CPU only: no kernel runs at construction."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.golden import common

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "init.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", ["tiny", "tiny_onehot", "tiny_proprio", "cfg2"])
def test_constructors_reproduce_the_reference_initialisation(name):
    import models

    cfg = Hh.make_config(name, "cpu")
    torch.manual_seed(0)
    wm = models.WorldModel(Hh.obs_space(name), None, 0, cfg)
    beh = models.ImagBehavior(cfg, wm)
    sd = dict(wm.state_dict())
    sd.update({k: v for k, v in beh.state_dict().items() if not k.startswith("_world_model.")})
    ref_keys = {k.split("/", 2)[2] for k in G.files if k.startswith(name + "/sum/")}
    assert set(sd) == ref_keys, set(sd) ^ ref_keys
    for k, v in sd.items():
        a = v.detach().numpy().astype(np.float64).reshape(-1)
        cs, ref = common.checksum(a), G[f"{name}/sum/{k}"]
        assert np.allclose(cs, ref, rtol=1e-6, atol=1e-9), f"{name} {k}: {cs} vs {ref}"
        assert np.array_equal(a[:4].astype(np.float32), G[f"{name}/head/{k}"]), f"{name} {k}: first values differ"


def test_weight_init_statistics():
    """Scale and truncation of the two initialisers on a free-standing layer (no fixture needed)."""
    import tools

    torch.manual_seed(1)
    lin = torch.nn.Linear(300, 500, bias=True)
    lin.apply(tools.weight_init)
    std = np.sqrt(1.0 / 400.0) / 0.87962566103423978
    w = lin.weight.detach().numpy()
    assert np.abs(w).max() <= 2.0 * std + 1e-7 and abs(w.std() - std * 0.8796) < 0.02 * std
    assert float(lin.bias.abs().max()) == 0.0
    lin.apply(tools.uniform_weight_init(1.0))
    lim = np.sqrt(3.0 / 400.0)
    w = lin.weight.detach().numpy()
    assert np.abs(w).max() <= lim and abs(w.std() - lim / np.sqrt(3)) < 0.02 * lim
    lin.apply(tools.uniform_weight_init(0.0))
    assert float(lin.weight.abs().max()) == 0.0
    ln = torch.nn.LayerNorm(8)
    ln.weight.data.fill_(3.0)
    ln.apply(tools.weight_init)
    assert float(ln.weight.min()) == 1.0 and float(ln.bias.abs().max()) == 0.0
