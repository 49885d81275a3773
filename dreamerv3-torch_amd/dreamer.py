"""`Dreamer` agent class with the reference's surface (dreamer.py:35-208), over the MI355X-native
`models` / `networks` / `tools`.  Only the class is provided: the reference's `main`, env
construction and CLI (dreamer.py:261-601) are host-side driver code outside the accelerated path and
keep working against this class unchanged.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

import exploration as expl
import models
import tools

to_np = lambda x: x.detach().cpu().numpy()


class Dreamer(nn.Module):
    def __init__(self, obs_space, act_space, config, logger, dataset):
        super().__init__()
        self._config = config
        self._logger = logger
        self._should_log = tools.Every(config.log_every)
        batch_steps = config.batch_size * config.batch_length
        self._should_train = tools.Every(batch_steps / config.train_ratio)
        self._should_pretrain = tools.Once()
        self._should_reset = tools.Every(config.reset_every)
        self._should_expl = tools.Until(int(config.expl_until / config.action_repeat))
        self._metrics = {}
        self._step = (logger.step if logger is not None else 0) // config.action_repeat
        self._update_count = 0
        self._dataset = dataset
        if getattr(config, "causal_world_model", False):
            raise NotImplementedError("causal world models are outside the accelerated path (SURVEY.md §2 #13-14)")
        self._wm = models.WorldModel(obs_space, act_space, self._step, config)
        self._task_behavior = models.ImagBehavior(config, self._wm)
        reward = lambda f, s, a: self._wm.heads["reward"](f).mean()  # extrinsic term of Plan2Explore (dreamer.py:80)
        self._expl_behavior = dict(
            greedy=lambda: self._task_behavior,
            random=lambda: expl.Random(config, act_space),
            plan2explore=lambda: expl.Plan2Explore(config, self._wm, reward),
        )[config.expl_behavior]().to(self._config.device)
        self._runner, self._stager, self._policy_runner = None, None, None
        self._metric_keys, self._metric_sum, self._metric_cnt, self._metric_idx = [], None, None, {}

    def __call__(self, obs, reset, state=None, training=True):
        step = self._step
        if training:
            steps = self._config.pretrain if self._should_pretrain() else self._should_train(step)
            for _ in range(steps):
                # back-to-back updates of one call: the behaviour phase of each is issued beside the next one's
                # world-model phase (dv3hip.graph.UpdateRunner.step_pipelined); the last one by _finish_updates()
                self._train(next(self._dataset), pipelined=steps > 1)
                self._update_count += 1
                self._metrics["update_count"] = self._update_count
            self._finish_updates()  # (the policy below, a checkpoint, the logger read the actor / critic / metrics)
            if self._should_log(step) and self._logger is not None:
                self._flush_metrics()
                for name, values in self._metrics.items():
                    self._logger.scalar(name, float(np.mean(values)))
                    self._metrics[name] = []
                if self._config.video_pred_log:
                    openl = self._wm.video_pred(next(self._dataset))
                    self._logger.video("train_openl", to_np(openl))
                self._logger.write(fps=True)
        policy_output, state = self._policy(obs, state, training)
        if training:
            self._step += len(reset)
            if self._logger is not None:
                self._logger.step = self._config.action_repeat * self._step
        return policy_output, state

    def _policy(self, obs, state, training, noise=None):
        """Acting step.  With config.hip_graph (default) the launch sequence is replayed from a hipGraph per
        (number of envs, training) signature (dv3hip.graph.PolicyRunner); injected noise (tests) takes the eager path."""
        # the Random explorer samples from torch's generator: not a static launch sequence
        random_now = training and self._config.expl_behavior == "random" and self._exploring()
        if (noise is None and bool(getattr(self._config, "hip_graph", True)) and self._policy_runner is not False
                and not random_now):
            from dv3hip.graph import CaptureRefused, PolicyRunner

            try:
                if self._policy_runner is None:
                    self._policy_runner = PolicyRunner(self)
                return self._policy_runner.step(obs, state, training)
            except CaptureRefused as e:  # ONLY a runtime that refuses capture; any other failure is a bug and propagates
                import sys

                print(f"[dv3hip] hipGraph capture of the policy step was refused ({e}); eager launches", file=sys.stderr)
                self._policy_runner = False
                torch.cuda.synchronize()  # surfaces an asynchronous HIP error instead of acting on a poisoned context
        return self._policy_eager(obs, state, training, noise)

    def _exploring(self):
        """dreamer.py:161: the exploration actor acts while `_should_expl` holds (expl_until 0 = always)."""
        return self._expl_behavior is not self._task_behavior and bool(self._should_expl(self._step))

    def _policy_eager(self, obs, state, training, noise=None):
        """dreamer.py:116-188: eval -> mode of the task actor; training -> a sample of the exploration actor while
        `_should_expl` holds (for 'greedy' that IS the task actor), else of the task actor (the counterfactual branch
        behind `_best_candidate` is unreachable, SURVEY.md section 0 gotcha 2).  noise (tests): dict(prior, post [n_envs,S,D] ~ Exp(1); act [n_envs,A], N(0,1) or
        Exp(1) for the one-hot actor) injected instead of the Philox stream."""
        latent, action = (None, None) if state is None else state
        nz = noise or {}
        obs = self._wm.preprocess(obs)
        embed = self._wm.encoder(obs)
        latent, _ = self._wm.dynamics.obs_step(latent, action, embed, obs["is_first"], noise=nz or None, prior=False)
        if getattr(self._config, "eval_state_mean", False):
            raise NotImplementedError("eval_state_mean needs continuous latents (dyn_discrete: 0)")
        feat = self._wm.dynamics.get_feat(latent)
        if training and self._exploring():
            actor = self._expl_behavior.actor(feat)
            action = actor.sample() if nz.get("act") is None else actor.sample(noise=nz["act"])
        else:
            actor = self._task_behavior.actor(feat)
            action = actor.sample(noise=nz.get("act")) if training else actor.mode()
        logprob = actor.log_prob(action)
        latent = {k: v.detach() for k, v in latent.items()}
        action = action.detach()
        if self._config.actor["dist"] == "onehot_gumble":
            raise NotImplementedError("actor dist onehot_gumble")
        return {"action": action, "logprob": logprob}, (latent, action)

    def _train(self, data, pipelined=False):
        """dreamer.py:192-208.  One update = WorldModel._train + ImagBehavior._train on the updated world model; the
        launch sequence is replayed from hipGraphs (dv3hip.graph.UpdateRunner, ~1130 launches per update) once it has
        been captured, with the host batch staged through pinned buffers (dv3hip.staging.BatchStager).  Metrics stay
        on the device: one running sum per key, read back once per log interval (see _flush_metrics).
        pipelined (the update loop of __call__): the behaviour phase of this update may still be pending when the call
        returns -- it is issued beside the next update's world-model phase, or by _finish_updates()."""
        if self._runner is None:
            from dv3hip.graph import UpdateRunner
            from dv3hip.staging import BatchStager

            self._runner = UpdateRunner(self._wm, self._task_behavior,
                                        use_graph=bool(getattr(self._config, "hip_graph", True)))
            self._stager = BatchStager(self._config.device)
            models.share_runner(self._wm, self._runner)  # (WorldModel._train / ImagBehavior._train share it)
        host = all(not isinstance(v, torch.Tensor) for v in data.values())
        explorer = self._expl_behavior is not self._task_behavior
        pipelined = pipelined and not explorer and bool(getattr(self._config, "pipeline_updates", True))
        # (the uploads go onto the stream the update is issued on; see UpdateRunner.launch_stream)
        with torch.cuda.stream(self._runner.launch_stream() or torch.cuda.current_stream()):
            staged = (self._stager.stage(data) if host else
                      {k: (v if k == "image" else v.to(torch.float32)) for k, v in data.items()})
            (self._runner.step_pipelined if pipelined else self._runner.step)(staged)
        mets = dict(self._runner.last_metrics)
        if explorer:
            # dreamer.py:201-203: the explorer trains on the same posterior states (eagerly: its objective runs torch
            # autograd, which a hipGraph segment cannot hold)
            staged = self._runner.last_data
            xm = self._expl_behavior.train(self._runner.last_post, self._runner.last_context, staged)[-1]
            mets.update({"expl_" + k: v for k, v in xm.items()})
        self._accumulate(mets)

    def _finish_updates(self):
        """End of a run of pipelined updates: issue the behaviour phase that is still pending."""
        r = self._runner
        if r is not None and r._pipe_pending:
            with torch.cuda.stream(r.launch_stream() or torch.cuda.current_stream()):
                r.flush()
            self._accumulate(dict(r.last_metrics))

    def _accumulate(self, mets):
        """Device-resident metrics -- the snapshots of the fused path (DeviceScalar) and the 0-d tensors an autograd-style
        caller's Optimizer.__call__ returns (the explorer) -- go into one running sum and one count per key (a pipelined
        call reports the world-model keys of one update and the behaviour keys of the previous one)."""
        dev = [(k, v._t if isinstance(v, models.DeviceScalar) else v.detach()) for k, v in mets.items()
               if isinstance(v, (models.DeviceScalar, torch.Tensor))]
        for k, v in mets.items():
            if not isinstance(v, (models.DeviceScalar, torch.Tensor)):
                self._metrics.setdefault(k, []).append(v)
        if not dev:
            return
        keys = tuple(k for k, _ in dev)
        idx = self._metric_idx.get(keys)
        if idx is None:
            new = [k for k in keys if k not in self._metric_keys]
            if new:
                self._metric_keys += new
                d = dev[0][1].device
                grow = torch.zeros(len(new), device=d, dtype=torch.float32)
                self._metric_sum = grow if self._metric_sum is None else torch.cat([self._metric_sum, grow])
                self._metric_cnt = grow.clone() if self._metric_cnt is None else torch.cat([self._metric_cnt, grow])
            idx = torch.tensor([self._metric_keys.index(k) for k in keys], device=dev[0][1].device)
            self._metric_idx[keys] = idx
        vals = torch.stack([t.reshape(()).to(torch.float32) for _, t in dev])
        self._metric_sum.index_add_(0, idx, vals)
        self._metric_cnt.index_add_(0, idx, torch.ones_like(vals))

    def _flush_metrics(self):
        """Device-side running sums -> self._metrics[name] = [mean] (what the reference's logging loop, which takes
        np.mean of each list, then reports): a single device-to-host copy per log interval."""
        if self._metric_sum is not None:
            both = torch.stack([self._metric_sum, self._metric_cnt]).cpu().numpy()
            for k, sm, n in zip(self._metric_keys, both[0], both[1]):
                if n > 0:
                    self._metrics.setdefault(k, []).append(float(sm / n))
            self._metric_sum.zero_()
            self._metric_cnt.zero_()
