"""One tiny pass of the hot path on cuda:0, checked against the CPU oracle (used by
__graft_entry__.smoke()).  The oracle is imported here as the checker only."""
from __future__ import annotations

import os
import sys

import torch

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)


def run(name: str = "tiny") -> None:
    from tests import helpers as Hh
    from tests.golden import common

    exp = Hh.oracle_update(name)
    cfg, wm, beh = Hh.build_models(name, device="cuda:0")
    s = common.SHAPES[name]
    n = {k: torch.from_numpy(v).cuda() for k, v in common.make_noise(name).items()}
    post, _, mets = wm._train(common.make_batch(name), noise=dict(q_prior=n["q_prior"], q_post=n["q_post"]))
    bm = beh._train(post, None, noise=dict(act=Hh.to_time_major_rows(n["act"], s["B"], s["T"]).contiguous(),
                                           q_img=Hh.to_time_major_rows(n["q_img"], s["B"], s["T"]).contiguous()))[-1]
    torch.cuda.synchronize()

    def chk(got, ref, what, tol=1e-4):
        got, ref = float(got), float(ref)
        assert abs(got - ref) <= tol * max(1.0, abs(ref)), f"smoke {what}: {got} vs oracle {ref}"

    chk(mets["model_loss"], exp["wm"]["model_loss"], "model_loss")
    chk(mets["model_grad_norm"], exp["model_grad_norm"], "model_grad_norm", 3e-4)
    chk(bm["actor_loss"], exp["beh"]["actor_loss"], "actor_loss")
    chk(bm["value_loss"], exp["beh"]["value_loss"], "value_loss")
    assert torch.equal(post["stoch"].cpu(), exp["wm"]["post"]["stoch"].detach()), "posterior samples differ"
    for k, v in wm.state_dict().items():
        err = (v.cpu() - exp["params_after"][k]).abs().max().item()
        assert err <= 1e-6 * max(1.0, exp["params_after"][k].abs().max().item()), f"smoke param {k}: {err}"
    print(f"[smoke] {name}: one full update on cuda:0 matches the CPU oracle "
          f"(model_loss {float(mets['model_loss']):.6f}, actor_loss {float(bm['actor_loss']):.6f})")
