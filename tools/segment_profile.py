#!/usr/bin/env python3
"""Where the time of one update goes, segment by segment and kernel by kernel, ON A 128-CU LANE (MI355X only).

The pipelined update (UpdateRunner.step_pipelined, schedule "lanes") runs each phase end to end on one half of the
chip, and both halves are full (profiles/r04_pipe_lanes.txt): what shortens the update now is whatever shortens a
segment on 128 compute units.  This tool launches the update's phases eagerly on the side lane with a HIP event pair
around every launch (ops.PROFILE, keys by shape), tagged with the segment labels of engine.Cuts, and prints per segment
the launches that make up its time with their rate against the lane's fp32 MFMA peak (78.6 TFLOP/s) or HBM share.

    python tools/segment_profile.py [cfg2] [--whole] [--all]   # --whole: on the whole-chip stream; --all: every kernel
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402


class Tagger:
    """Stands where graph.PhaseRecorder stands during capture: a cut mark only renames the profile prefix."""

    def __init__(self, first):
        from dv3hip import ops

        self.ops = ops
        ops.PROFILE.prefix = first + "|"

    def mark(self, label):
        if "@" in label:
            return
        self.ops.PROFILE.prefix = label + "|"


def main():
    import models
    import tools
    from dv3hip import engine, ops, shapes
    from dv3hip.graph import UpdateRunner

    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "cfg2"
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = shapes.make_config(name, str(dev))
    wm = models.WorldModel(shapes.obs_space(name), None, 0, cfg).to(dev)
    beh = models.ImagBehavior(cfg, wm).to(dev)
    wm.requires_grad_(False), beh.requires_grad_(False)
    tools.default_rng(dev, seed=1234)
    data = {k: torch.from_numpy(v).to(dev) for k, v in shapes.synthetic_batch(name, 0).items()}
    data = {k: (v if k == "image" else v.to(torch.float32)) for k, v in data.items()}
    r = UpdateRunner(wm, beh, use_graph=False)
    for _ in range(3):
        r.step(data)
    torch.cuda.synchronize()
    lanes = engine.Lanes.get(dev)
    stream = lanes.streams["whole" if "--whole" in sys.argv else "side"]
    cus = 256 if "--whole" in sys.argv else lanes.cus["side"]
    peak = 157.3 * cus / 256
    P = ops.PROFILE
    P.by_shape = True
    res = {}
    with torch.cuda.stream(stream):
        for rep in range(3):
            P.start()
            engine.Cuts.recorder = Tagger("wm.pre")
            try:
                wm.train_fwd_bwd(data)
                P.prefix = "wm.opt|"
                post, ctx, m1 = wm.train_opt(allreduce=False)
                engine.Cuts.recorder = Tagger("bh.start")
                beh.train_fwd_bwd(post)
                P.prefix = "bh.opt|"
                beh.train_opt(allreduce=False)
            finally:
                engine.Cuts.recorder = None
                P.prefix = ""
            out = P.stop()
            for k, v in out.items():
                a = res.setdefault(k, dict(launches=v["launches"], flops=v["flops"], bytes=v["bytes"], ms=[]))
                a["ms"].append(v["ms"])
    segs = {}
    for k, v in res.items():
        seg, key = k.split("|", 1)
        segs.setdefault(seg, []).append((key, v["launches"], v["flops"], v["bytes"], min(v["ms"])))
    print(f"{name}: eager launches on the {'whole chip' if '--whole' in sys.argv else 'side lane'} ({cus} CUs; fp32 MFMA peak "
          f"{peak:.1f} TFLOP/s); per launch: device time between two HIP events, best of 3 updates")
    tot_all = 0.0
    for seg, rows in segs.items():
        tot = sum(r_[4] for r_ in rows)
        fl = sum(r_[2] for r_ in rows)
        tot_all += tot
        print(f"\n{seg}: {tot:.3f} ms in {sum(r_[1] for r_ in rows)} launches, {fl / 1e9:.1f} GFLOP "
              f"({fl / 1e9 / max(tot, 1e-9):.1f} TFLOP/s = {100 * fl / 1e9 / max(tot, 1e-9) / peak:.0f} % of the lane's peak)")
        for key, n, f, b, ms in sorted(rows, key=lambda r_: -r_[4])[:(10 ** 6 if "--all" in sys.argv else 14)]:
            us = ms * 1e3 / n
            rate = f"{f / 1e9 / ms:6.1f} TF/s ({100 * f / 1e9 / ms / peak:3.0f} %)" if f else (f"{b / 1e6 / ms:6.0f} GB/s" if b else "")
            print(f"   {ms:7.3f} ms  {n:4d} x {us:7.1f} us  {rate:22s} {key[:110]}")
    print(f"\nsum of launches: {tot_all:.3f} ms")


if __name__ == "__main__":
    main()
