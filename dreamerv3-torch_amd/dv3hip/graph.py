"""hipGraph replay of the training update.

One update is ~3000 short kernel launches (two sequential scans of small GEMMs); launched eagerly
from Python it is host-bound.  The launch sequence is static (fixed shapes, no host reads, RNG and
Adam step counters live in device memory), so it is captured once into HIP graphs and replayed:
MI355X-native replacement for the reference's (inert) torch.compile switch (dreamer.py:75-79).

Graph segments, with the data-parallel collectives kept OUTSIDE capture (eager RCCL calls between
replays):   [world model fwd+bwd] -> all-reduce -> [WM clip+Adam | behaviour fwd+bwd] -> all-reduce x2
-> [actor / critic clip+Adam].  On a single rank the all-reduces are no-ops.
"""
from __future__ import annotations

from typing import Dict

import torch


class UpdateRunner:
    def __init__(self, wm, beh, use_graph: bool = True, warm: int = 2):
        self.wm, self.beh = wm, beh
        cfg = wm._config
        self.use_graph = use_graph and cfg.critic["slow_target_update"] == 1
        self.warm = warm
        self._calls = 0
        self._static: Dict[str, torch.Tensor] = {}
        self._graphs = None
        self.last_metrics = {}

    # -- eager reference sequence ----------------------------------------------------------------
    def _eager(self, data):
        post, _, m1 = self.wm._train(data)
        m2 = self.beh._train(post, None)[-1]
        self.last_metrics = {**m1, **m2}

    def _load(self, data):
        if not self._static:
            for k, v in data.items():
                self._static[k] = v.clone()
            return
        for k, v in data.items():
            self._static[k].copy_(v, non_blocking=True)

    def _capture(self):
        wm, beh = self.wm, self.beh
        g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        # thread_local: with a process group alive, RCCL's watchdog thread polls events while we capture; the
        # default (global) mode would treat that as a capture violation
        mode = dict(capture_error_mode="thread_local")
        with torch.cuda.graph(g1, pool=pool, **mode):
            wm.train_fwd_bwd(self._static)
        wm._model_opt.bucket.allreduce()
        with torch.cuda.graph(g2, pool=pool, **mode):
            post, _, m1 = wm.train_opt(allreduce=False)
            beh.train_fwd_bwd(post)
        beh._actor_opt.bucket.allreduce()
        beh._value_opt.bucket.allreduce()
        beh.sync_ema()
        with torch.cuda.graph(g3, pool=pool, **mode):
            m2 = beh.train_opt(allreduce=False)[-1]
        self._graphs = (g1, g2, g3)
        self.last_metrics = {**m1, **m2}

    def step(self, data, eager: bool = False):
        """data: dict of device tensors (image uint8 [B,T,64,64,3], action, reward, is_first, is_terminal ...)."""
        self._calls += 1
        if eager or not self.use_graph or self._calls <= self.warm:
            self._eager(data)
            return
        self._load(data)
        if self._graphs is None:
            torch.cuda.synchronize()
            try:
                self._capture()  # records only: nothing has executed yet, so fall through and replay
            except Exception as e:  # e.g. a runtime that refuses capture: keep training, launch eagerly
                import sys

                print(f"[dv3hip] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches",
                      file=sys.stderr)
                self.use_graph, self._graphs = False, None
                torch.cuda.synchronize()
                self._eager(data)
                return
        g1, g2, g3 = self._graphs
        g1.replay()
        self.wm._model_opt.bucket.allreduce()
        g2.replay()
        self.beh._actor_opt.bucket.allreduce()
        self.beh._value_opt.bucket.allreduce()
        self.beh.sync_ema()
        g3.replay()


class PolicyRunner:
    """hipGraph replay of the acting step (Dreamer._policy: preprocess -> encoder -> obs_step -> actor; SURVEY 8(f)
    N1).  Eager, the step is ~50 launches and host-bound (0.8 ms at 1-16 envs); its launch sequence is static for a
    given (number of envs, training flag), so it is captured once per signature and replayed.  Inputs are copied into
    static device buffers (host observations through one pinned staging buffer per key), the outputs of a replay are
    packed into one buffer inside the graph and leave it with a single clone, so what the caller gets back are fresh
    tensors exactly as from the eager path."""

    def __init__(self, agent):
        self.agent = agent
        self._sig = {}

    def _build(self, obs, state, training):
        ag = self.agent
        dyn = ag._wm.dynamics
        dev = torch.device(ag._config.device)
        n = len(obs["is_first"])
        S, D, De, A = dyn._stoch, dyn._discrete, dyn._deter, dyn._num_actions
        st = dict(obs={}, pin={})
        for k, v in obs.items():
            t = torch.as_tensor(v)
            dt = torch.uint8 if (k == "image" and t.dtype == torch.uint8) else torch.float32
            st["obs"][k] = torch.zeros(tuple(t.shape), dtype=dt, device=dev)
            st["pin"][k] = torch.zeros(tuple(t.shape), dtype=dt).pin_memory()
        st["state"] = {"stoch": torch.zeros(n, S, D, device=dev), "deter": torch.zeros(n, De, device=dev),
                       "logit": torch.zeros(n, S, D, device=dev)}
        st["action"] = torch.zeros(n, A, device=dev)
        sizes = [n * A, n, n * S * D, n * De, n * S * D]
        st["packed"] = torch.zeros(sum(sizes), device=dev)
        st["sizes"] = sizes

        def core():
            out, (latent, action) = ag._policy_eager(st["obs"], (st["state"], st["action"]), training)
            parts = [out["action"], out["logprob"], latent["stoch"], latent["deter"], latent["logit"]]
            off = 0
            for p_, sz in zip(parts, sizes):
                st["packed"][off:off + sz].copy_(p_.reshape(-1))
                off += sz

        self._load(st, obs, state)
        import tools

        rng = tools.default_rng(dev)
        saved = rng.state.clone()
        core()  # warm: code objects, workspaces
        rng.state.copy_(saved)  # the warm call is not a step: leave the Philox stream where the caller had it
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            core()
        st["graph"] = g
        return st

    @staticmethod
    def _load(st, obs, state):
        ev = st.get("h2d_done")
        if ev is not None:
            ev.synchronize()  # the pinned buffers are rewritten below: their previous upload must be over
        for k, v in obs.items():
            if isinstance(v, torch.Tensor):
                st["obs"][k].copy_(v, non_blocking=True)
            else:
                pin = st["pin"][k]
                pin.copy_(torch.as_tensor(v))
                st["obs"][k].copy_(pin, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["h2d_done"] = ev
        if state is None:
            # obs_step with prev_state None == every env resets (networks.py:176-180): force is_first
            st["obs"]["is_first"].fill_(1.0)
            st["action"].zero_()
        else:
            latent, action = state
            for k in ("stoch", "deter", "logit"):
                st["state"][k].copy_(latent[k], non_blocking=True)
            st["action"].copy_(action, non_blocking=True)

    def step(self, obs, state, training):
        n = len(obs["is_first"])
        key = (n, bool(training), tuple(sorted(obs)))
        st = self._sig.get(key)
        if st is None:
            st = self._build(obs, state, training)
            self._sig[key] = st
        self._load(st, obs, state)
        st["graph"].replay()
        flat = st["packed"].clone()
        dyn = self.agent._wm.dynamics
        S, D, De, A = dyn._stoch, dyn._discrete, dyn._deter, dyn._num_actions
        outs, off = [], 0
        for sz in st["sizes"]:
            outs.append(flat[off:off + sz])
            off += sz
        action, logprob = outs[0].view(n, A), outs[1].view(n)
        latent = {"stoch": outs[2].view(n, S, D), "deter": outs[3].view(n, De), "logit": outs[4].view(n, S, D)}
        return {"action": action, "logprob": logprob}, (latent, action)
