#!/usr/bin/env python3
"""Per-step error of the imagination rollout GPU vs oracle for a shape config (diagnostic)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import dv3_oracle as O  # noqa: E402
from tests import helpers as Hh  # noqa: E402
from tests.golden import common  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_b4"
    s = common.SHAPES[name]
    B, T, H = s["B"], s["T"], s["H"]
    exp = Hh.oracle_update(name, threads=16, piecewise=True)
    cfg, wm, beh = Hh.build_models(name)
    n = {k: torch.from_numpy(v).cuda() for k, v in common.make_noise(name).items()}
    wm.train_fwd_bwd(common.make_batch(name), noise=dict(q_prior=n["q_prior"], q_post=n["q_post"]))
    post = {k: v.clone() for k, v in wm._pending[0].items()}
    beh._update_slow_target = lambda: None
    im_noise = dict(act=Hh.to_time_major_rows(n["act"], B, T).contiguous(), q_img=Hh.to_time_major_rows(n["q_img"], B, T).contiguous())
    beh.train_fwd_bwd(post, noise=im_noise)
    im = beh._im
    eb = exp["beh0"]
    un = lambda x: Hh.from_time_major_rows(x, B, T)
    d = un(im["deter"]).cpu()
    print("post deter err", float((post["deter"].cpu() - exp["wm"]["post"]["deter"]).abs().max()))
    for t in range(H):
        e = (d[t] - eb["states"]["deter"][t]).abs()
        a = (un(im["action"])[t].cpu() - eb["actions"][t]).abs().max()
        st_eq = torch.equal(un(im["stoch"].view(H, -1, s["stoch"], s["discrete"]))[t].cpu(), eb["states"]["stoch"][t])
        print(f"t={t} deter max err {float(e.max()):.3e} mean {float(e.mean()):.3e}  action max err {float(a):.3e} stoch equal {st_eq}")
    # one step from the oracle's own state 0: intermediates
    pc = common.path_config(name)
    p = {k: torch.from_numpy(v) for k, v in common.make_weights(name).items()}
    st0 = {k: v[0] for k, v in eb["states"].items()}
    a0 = eb["actions"][0]
    x = torch.cat([st0["stoch"].flatten(1), a0], -1)
    x1pre = x @ p["dynamics._img_in_layers.0.weight"].t()
    x1 = F.silu(O.layer_norm(x1pre, p["dynamics._img_in_layers.1.weight"], p["dynamics._img_in_layers.1.bias"]))
    gpre = torch.cat([x1, st0["deter"]], -1) @ p["dynamics._cell.layers.GRU_linear.weight"].t()
    g = im["step"]
    print("x1pre err", float((un(g["x1pre"])[0].cpu() - x1pre).abs().max()), "scale", float(x1pre.abs().max()))
    print("x1 err", float((un(g["x1"])[0].cpu() - x1).abs().max()))
    print("gpre err", float((un(g["gpre"])[0].cpu() - gpre).abs().max()), "scale", float(gpre.abs().max()), "std", float(gpre.std()))
    d1 = O.gru_cell(p, x1, st0["deter"])
    print("deter1 err (oracle gates on oracle gpre vs gpu)", float((d[1] - d1).abs().max()))
    # gates applied by torch to the GPU's gpre
    parts = O.layer_norm(un(g["gpre"])[0].cpu(), p["dynamics._cell.layers.GRU_norm.weight"], p["dynamics._cell.layers.GRU_norm.bias"])
    De = s["deter"]
    r, c, u = torch.sigmoid(parts[:, :De]), None, torch.sigmoid(parts[:, 2 * De:] - 1)
    c = torch.tanh(r * parts[:, De:2 * De])
    d1g = u * c + (1 - u) * un(im["deter"])[0].cpu()
    print("deter1 err (torch gates on GPU gpre vs gpu gates)", float((d[1] - d1g).abs().max()))
    print("LN rstd of gpre rows:", float((1 / torch.sqrt(gpre.var(-1, unbiased=False) + 1e-3)).max()))


if __name__ == "__main__":
    main()
