#!/usr/bin/env python3
"""Two-update software pipeline (UpdateRunner.step_pipelined) against the serial update (MI355X only): ms per update
of a stream of updates, of pairs (two updates per call + flush, what Dreamer.__call__ issues at the dmc configs), and
the timeline of one pipelined iteration.

    python tools/pipe_bench.py [cfg2] [--steps 30]
    DV3HIP_LIB=.../libdv3hip_dev.so DV3_PIPE_DEFER=post DV3_PIPE_A_SPLIT=10 python tools/pipe_bench.py cfg2   # A/B of the plan
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import models
    import tools
    from dv3hip import shapes
    from dv3hip.graph import UpdateRunner

    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "cfg2"
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 30
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = shapes.make_config(name, str(dev))
    wm = models.WorldModel(shapes.obs_space(name), None, 0, cfg).to(dev)
    beh = models.ImagBehavior(cfg, wm).to(dev)
    wm.requires_grad_(False), beh.requires_grad_(False)
    tools.default_rng(dev, seed=1234)
    data = {k: torch.from_numpy(v).to(dev) for k, v in shapes.synthetic_batch(name, 0).items()}
    r = UpdateRunner(wm, beh)
    for _ in range(5):
        r.step(data)
    torch.cuda.synchronize()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def serial(n):
        for _ in range(n):
            r.step(data)

    def stream(n):
        for _ in range(n):
            r.step_pipelined(data)
        r.flush()

    def pairs(n):
        for _ in range(n // 2):
            r.step_pipelined(data)
            r.step_pipelined(data)
            r.flush()

    stream(4)  # captures
    res = {}
    for rep in range(3):
        for nm, fn in (("serial", serial), ("pipelined", stream), ("pairs", pairs)):
            res.setdefault(nm, []).append(timed(fn, steps))
    print(f"{name}: ms per update  " + "  ".join(f"{k} {min(v):.3f} (median {np.median(v):.3f})" for k, v in res.items()),
          f"| plan {r.pipe_plan} | model_loss {float(r.wm_metrics['model_loss']):.4f}")
    if r._pipe is None:
        print("pipeline not taken")
        return
    # timeline of steady iterations
    tls = []
    r.step_pipelined(data)
    for _ in range(5):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        r._pipe_trace = []
        with torch.cuda.stream(r._home[1]):
            e0.record()
        r.step_pipelined(data)
        e1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(r._home[1]):
            e1.record()
        torch.cuda.synchronize()
        tls.append((e0.elapsed_time(e1), [(lb, e0.elapsed_time(a), e0.elapsed_time(b)) for lb, a, b in r._pipe_trace]))
        r._pipe_trace = None
    r.flush()
    if "--segments" in sys.argv:
        # every segment alone: on the whole chip and on the 128-CU side lane (values are garbage afterwards)
        P = r._pipe
        whole, side = r._home[1], P["lanes"].streams["side"]
        print("segments alone, ms (whole chip / side lane):")
        for lb, g in list(P["W"]) + list(P["B"]) + [("wm.opt", P["wopt"]), ("bh.opt", P["bopt"])]:
            res = []
            for st in (whole, side):
                tt = []
                for _ in range(5):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    with torch.cuda.stream(st):
                        a.record()
                        g.replay()
                        b.record()
                    torch.cuda.synchronize()
                    tt.append(a.elapsed_time(b))
                res.append(sorted(tt)[2])
            print(f"  {lb:12s} {res[0]:7.3f} / {res[1]:7.3f}   x{res[1] / max(res[0], 1e-6):.2f}")
    tls.sort(key=lambda x: x[0])
    tot, tl = tls[len(tls) // 2]
    print(f"timeline of one pipelined iteration ({tot:.3f} ms, event-timed: each segment's events add a few us):")
    for lb, a, b in tl:
        print(f"  {lb:12s} {a:7.3f} - {b:7.3f}  ({b - a:.3f})")


if __name__ == "__main__":
    main()
