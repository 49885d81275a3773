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
    """Median device time of fn's launch sequence replayed from a hipGraph."""
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
    ts.sort()
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
