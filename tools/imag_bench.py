#!/usr/bin/env python3
"""Device time of the imagination rollout (ImagBehavior._imagine forward) and of the behaviour update, as
hipGraph replays, plus a per-kernel (by shape) HIP-event breakdown of one eager rollout (MI355X only).

    python tools/imag_bench.py [cfg2] [--json out.json]
    python tools/imag_bench.py cfg2 --replays N      # ONLY N hipGraph replays of the rollout after one warm update:
                                                     # the command tools/pmc_timg.sh puts under rocprofv3 --pmc

Algorithmic FLOPs are the dense-equivalent 2*M*N*K of SURVEY.md 8(d) (the one-hot gather layers are priced as the
Linear they replace); "mfma_gflop" counts only what the MFMA kernels execute.
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests import helpers as Hh  # noqa: E402
from tests.golden import common  # noqa: E402


def replay_ms(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    from dv3hip import ops

    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    args = [a for a in sys.argv[1:] if not a.startswith("--") and a != out_json and not a.isdigit()]
    name = args[0] if args else "cfg2"
    cfg, wm, beh = Hh.build_models(name)
    s = common.SHAPES[name]
    H = s["H"]
    data = {k: torch.from_numpy(v).cuda() for k, v in common.make_batch(name).items()}
    post, _, _ = wm._train(data)
    post = {k: v.clone() for k, v in post.items()}
    beh._imagine_fwd(post, H)  # warm: code objects loaded, workspaces allocated
    torch.cuda.synchronize()
    if "--replays" in sys.argv:
        n = int(sys.argv[sys.argv.index("--replays") + 1])
        t = replay_ms(lambda: beh._imagine_fwd(post, H), reps=n)
        print(json.dumps({"config": name, "replays": n, "T_img_ms": t}))
        return
    ops.PROFILE.by_shape = True
    ops.PROFILE.start()
    beh._imagine_fwd(post, H)
    prof = ops.PROFILE.stop()
    gflop = sum(v["flops"] for v in prof.values()) / 1e9
    mfma = sum(v["flops"] for k, v in prof.items() if k.startswith("gemm_kernel")) / 1e9
    t_img = replay_ms(lambda: beh._imagine_fwd(post, H))

    def behaviour():
        beh.train_fwd_bwd(post)
        beh.train_opt(allreduce=False)

    t_beh = replay_ms(behaviour, reps=10)
    rows = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
    n_launch = sum(v["launches"] for v in prof.values())
    print(f"{name}: T_img {t_img:.3f} ms ({n_launch} launches, {gflop:.1f} GFLOP dense-equivalent = "
          f"{gflop / t_img:.1f} TFLOP/s = {gflop / t_img / 157.3 * 100:.1f} % of 157.3; MFMA kernels execute "
          f"{mfma:.1f} GFLOP)   T_beh {t_beh:.3f} ms")
    for k, v in rows[:24]:
        print(f"  {v['ms']:8.3f} ms  n={v['launches']:4d}  {v['ms'] * 1e3 / v['launches']:7.1f} us/launch  "
              f"{v['flops'] / max(v['ms'], 1e-9) / 1e9:7.1f} TF/s  {k}")
    if "--beh" in sys.argv:  # the whole behaviour update (imagine + returns + losses + backward + Adam), eager-timed
        ops.PROFILE.start()
        behaviour()
        pb = ops.PROFILE.stop()
        tot = sum(v["ms"] for v in pb.values())
        print(f"behaviour update, per kernel (eager event times, sum {tot:.2f} ms, {sum(v['launches'] for v in pb.values())} launches):")
        for k, v in sorted(pb.items(), key=lambda kv: -kv[1]["ms"])[:40]:
            print(f"  {v['ms']:8.3f} ms  n={v['launches']:4d}  {v['ms'] * 1e3 / v['launches']:7.1f} us/launch  "
                  f"{v['flops'] / max(v['ms'], 1e-9) / 1e9:7.1f} TF/s  {k}")
    if out_json:
        json.dump({"config": name, "T_img_ms": t_img, "T_beh_ms": t_beh, "launches": n_launch,
                   "gflop_dense_equivalent": gflop, "gflop_mfma": mfma,
                   "by_kernel": {k: v for k, v in rows}}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
