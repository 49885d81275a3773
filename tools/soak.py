#!/usr/bin/env python3
"""Soak run (MI355X only): N hipGraph-replayed updates of a BASELINE config on fresh synthetic batches staged from the
host each step; checks that every logged metric stays finite and prints the loss trajectory.

    python tools/soak.py [cfg2] [--steps 200] [--pipelined | --pairs]

--pipelined: the run of updates through UpdateRunner.step_pipelined (the two-update pipeline), flushed every 10 updates
where the metrics are read; --pairs: two updates per call + flush, as Dreamer.__call__ issues them at the dmc configs.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import models
    from dv3hip import shapes
    from dv3hip.graph import UpdateRunner
    from dv3hip.staging import BatchStager

    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "cfg2"
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 200
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = shapes.make_config(name, str(dev))
    wm = models.WorldModel(shapes.obs_space(name), None, 0, cfg).to(dev)
    beh = models.ImagBehavior(cfg, wm).to(dev)
    wm.requires_grad_(False), beh.requires_grad_(False)
    runner = UpdateRunner(wm, beh)
    stager = BatchStager(dev)
    batches = [shapes.synthetic_batch(name, seed) for seed in range(8)]  # 8 different replay minibatches, cycled
    hist = []
    import time

    t0 = time.perf_counter()
    mode = "pipelined" if "--pipelined" in sys.argv else ("pairs" if "--pairs" in sys.argv else "serial")
    for i in range(steps):
        # (uploads and update on the stream the runner launches on: dv3hip.graph.UpdateRunner.launch_stream)
        with torch.cuda.stream(runner.launch_stream() or torch.cuda.current_stream()):
            (runner.step if mode == "serial" else runner.step_pipelined)(stager.stage(batches[i % len(batches)]))
            if mode == "pairs" and i % 2 == 1:
                runner.flush()
        if i % 10 == 9 or i == steps - 1:
            with torch.cuda.stream(runner.launch_stream() or torch.cuda.current_stream()):
                runner.flush()
            runner.last_metrics = {**runner.wm_metrics, **runner.beh_metrics}
            m = {k: float(v) for k, v in runner.last_metrics.items() if np.ndim(float(v)) == 0}
            bad = [k for k, v in m.items() if not np.isfinite(v)]
            assert not bad, (i, bad)
            hist.append((i + 1, m["model_loss"], m["actor_loss"], m["value_loss"], m["model_grad_norm"]))
            print(f"update {i + 1:4d}: model_loss {m['model_loss']:10.3f}  actor_loss {m['actor_loss']:9.4f}  "
                  f"value_loss {m['value_loss']:9.4f}  model_grad_norm {m['model_grad_norm']:9.2f}", flush=True)
    assert hist[-1][1] < hist[0][1], "model loss did not decrease"
    torch.cuda.synchronize()
    print(f"soak ok ({mode}): {steps} updates, {(time.perf_counter() - t0) / steps * 1e3:.2f} ms per update including the metric read-backs")


if __name__ == "__main__":
    main()
