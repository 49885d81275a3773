#!/usr/bin/env python3
"""Agent-level soak (MI355X only): dreamer.Dreamer driven as tools.simulate drives it -- `pretrain` back-to-back updates
inside one call (the two-update pipeline), then calls with two updates each (train_ratio 512 at 4 envs), host batches
from a generator, metrics logged every call.

    python tools/agent_soak.py [cfg2] [--pretrain 2000] [--calls 500]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


class Logger:
    def __init__(self):
        self.step, self.scalars, self.writes = 0, {}, 0

    def scalar(self, k, v):
        self.scalars[k] = v

    def video(self, *a, **k):
        pass

    def write(self, fps=False):
        self.writes += 1


def main():
    import dreamer
    from dv3hip import shapes

    args = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
    name = args[0] if args else "cfg2"
    opt = lambda k, d: int(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d
    pretrain, calls = opt("--pretrain", 2000), opt("--calls", 500)
    cfg = shapes.make_config(name, "cuda:0")
    cfg.pretrain, cfg.log_every, cfg.video_pred_log, cfg.envs = pretrain, 200, False, 4
    batches = [shapes.synthetic_batch(name, seed) for seed in range(8)]

    def dataset():
        i = 0
        while True:
            yield batches[i % len(batches)]
            i += 1

    torch.manual_seed(0)
    logger = Logger()
    agent = dreamer.Dreamer(shapes.obs_space(name), None, cfg, logger, dataset()).to(cfg.device)
    agent.requires_grad_(False)
    n_envs = 4
    obs = {"image": np.zeros((n_envs, 64, 64, 3), np.uint8), "is_first": np.ones(n_envs, bool),
           "is_terminal": np.zeros(n_envs, bool)}
    for k, w in (shapes.PROPRIO_KEYS if shapes.SHAPES[name]["encoder"] != "cnn" else ()):
        obs[k] = np.zeros((n_envs, w), np.float32)
    t0 = time.perf_counter()
    out, state = agent(obs, np.ones(n_envs, bool), None, training=True)  # `pretrain` updates, then acts
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"{name}: {agent._update_count} pretrain updates in one call: {(t1 - t0) / agent._update_count * 1e3:.2f} ms per update "
          f"(incl. host staging and the first call's captures); model_loss {logger.scalars.get('model_loss')}", flush=True)
    obs["is_first"][:] = False
    n0 = agent._update_count
    for c in range(calls):
        out, state = agent(obs, np.zeros(n_envs, bool), state, training=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = agent._update_count - n0
    assert agent._runner.use_graph and not agent._runner._pipe_pending
    for k in ("model_loss", "actor_loss", "value_loss", "model_grad_norm"):
        assert np.isfinite(logger.scalars[k]), (k, logger.scalars[k])
    print(f"{name}: {calls} calls, {n} updates ({n / calls:.1f} per call) + {calls} policy steps: "
          f"{(t2 - t1) / calls * 1e3:.2f} ms per call; model_loss {logger.scalars['model_loss']:.3f}, "
          f"{logger.writes} logger writes; pipeline captured: {agent._runner._pipe is not None}")
    print("agent soak ok")


if __name__ == "__main__":
    main()
