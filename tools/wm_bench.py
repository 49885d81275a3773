#!/usr/bin/env python3
"""Per-kernel table of the world-model half of one update (WorldModel.train_fwd_bwd + train_opt) at a BASELINE config
(MI355X only): eager launches timed with HIP events (each includes ~3-5 us of launch overhead), plus the hipGraph replay
time of the same sequence.

    python tools/wm_bench.py [cfg2]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from tests import helpers as Hh  # noqa: E402
from tests.golden import common  # noqa: E402


def replay_ms(fn, reps=10):
    """Median device time of fn's launch sequence replayed from hipGraphs (one per lane segment, as UpdateRunner does)."""
    from dv3hip.graph import SegmentRecorder

    from dv3hip import engine

    # main stream of the replays: the runner's blocking whole-chip stream (WM_MAIN=torch: a plain torch stream)
    st = (torch.cuda.Stream() if os.environ.get("WM_MAIN") == "torch"
          else engine.Lanes.get(torch.device("cuda", torch.cuda.current_device())).whole_chip_stream())
    with torch.cuda.stream(st):
        rec = SegmentRecorder(torch.cuda.graph_pool_handle(), torch.device("cuda", torch.cuda.current_device())).record(fn)
        rec.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rec.replay()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        # each segment alone, on its own lane
        parts = []
        for lane, g in rec.segments:
            if g is None:
                continue
            s_ = st if lane == "main" else rec.lanes.streams[lane]
            tt = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                with torch.cuda.stream(s_):
                    a.record()
                    g.replay()
                    b.record()
                torch.cuda.synchronize()
                tt.append(a.elapsed_time(b))
            parts.append(f"{lane} {sorted(tt)[2]:.3f}")
        # timeline of one full replay: start / end of every segment relative to the first launch
        tl = []
        for _ in range(5):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            marks = []
            rec.replay(trace=marks)
            torch.cuda.synchronize()
            tl.append([(lane, e0.elapsed_time(a), e0.elapsed_time(b)) for lane, a, b in marks])
        tl.sort(key=lambda m: m[-1][2])
        print("timeline (ms):", ", ".join(f"{lane} {a:.3f}-{b:.3f}" for lane, a, b in tl[2]))
        lanes_ = [(lane, g) for lane, g in rec.segments if lane != "main" and g is not None]
        if len(lanes_) == 2:
            # the two lane segments side by side: each one's own duration and the pair's
            res = []
            for _ in range(5):
                ev = {k: torch.cuda.Event(enable_timing=True) for k in ("f", "j", "a0", "b0", "a1", "b1")}
                torch.cuda.synchronize()
                ev["f"].record()
                for i, (lane, g) in enumerate(lanes_):
                    s_ = rec.lanes.streams[lane]
                    s_.wait_stream(st)
                    with torch.cuda.stream(s_):
                        ev[f"a{i}"].record()
                        g.replay()
                        ev[f"b{i}"].record()
                for lane, g in lanes_:
                    st.wait_stream(rec.lanes.streams[lane])
                ev["j"].record()
                torch.cuda.synchronize()
                res.append((ev["f"].elapsed_time(ev["j"]), ev["a0"].elapsed_time(ev["b0"]), ev["a1"].elapsed_time(ev["b1"]),
                            ev["f"].elapsed_time(ev["a1"])))
            res.sort()
            r = res[2]
            parts.append(f"| pair {r[0]:.3f} ({lanes_[0][0]} {r[1]:.3f}, {lanes_[1][0]} {r[2]:.3f}, second starts at +{r[3]:.3f})")
    print("segments alone (ms):", ", ".join(parts), f"| lanes: {rec.lanes.cus}")
    return ts[len(ts) // 2]


def main():
    from dv3hip import ops

    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "cfg2"
    cfg, wm, beh = Hh.build_models(name)
    data = {k: torch.from_numpy(v).cuda() for k, v in common.make_batch(name).items()}

    def step():
        wm.train_fwd_bwd(data)
        wm.train_opt(allreduce=False)

    step()
    torch.cuda.synchronize()
    ops.PROFILE.by_shape = True
    ops.PROFILE.start()
    step()
    prof = ops.PROFILE.stop()
    t = replay_ms(step, reps=10)
    tot = sum(v["ms"] for v in prof.values())
    print(f"{name}: world-model update {t:.3f} ms replayed; eager event sum {tot:.2f} ms, "
          f"{sum(v['launches'] for v in prof.values())} launches")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:45]:
        print(f"  {v['ms']:8.3f} ms  n={v['launches']:4d}  {v['ms'] * 1e3 / v['launches']:7.1f} us/launch  "
              f"{v['flops'] / max(v['ms'], 1e-9) / 1e9:7.1f} TF/s  {k}")


if __name__ == "__main__":
    main()
