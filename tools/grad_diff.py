#!/usr/bin/env python3
"""Per-tensor world-model gradient comparison GPU path vs CPU oracle for a named shape config (diagnostic)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from tests import helpers as Hh  # noqa: E402
from tests.golden import common  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_b4"
    exp = Hh.oracle_update(name, threads=16)
    cfg, wm, beh = Hh.build_models(name)
    n = {k: torch.from_numpy(v).cuda() for k, v in common.make_noise(name).items()}
    wm.train_fwd_bwd(common.make_batch(name), noise=dict(q_prior=n["q_prior"], q_post=n["q_post"]))
    rows = []
    tot_ref = tot_got = 0.0
    for k, p in wm.named_parameters():
        g, r = p.grad.detach().cpu().double(), exp["wm_grads"][k].double()
        tot_ref += float((r ** 2).sum())
        tot_got += float((g ** 2).sum())
        rows.append((float((g - r).norm()), float(r.norm()), float((g - r).abs().max()), float(r.abs().max()), k))
    rows.sort(reverse=True)
    print(f"norm ref {tot_ref ** 0.5:.4f} got {tot_got ** 0.5:.4f}  kernel-reported {float(wm._model_opt.bucket.grad_norm):.4f}")
    for dn, rn, dm, rm, k in rows[:14]:
        print(f"  |d|={dn:10.4e} |ref|={rn:10.4e} rel={dn / max(rn, 1e-30):9.2e}  maxd={dm:9.2e} maxref={rm:9.2e}  {k}")
    post = wm._pending[0]
    print("post stoch equal:", torch.equal(post["stoch"].cpu(), exp["wm"]["post"]["stoch"].detach()),
          " model_loss", float(wm._pending[3]), float(exp["wm"]["model_loss"]))


if __name__ == "__main__":
    main()
