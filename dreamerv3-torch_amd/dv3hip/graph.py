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
        g3.replay()
