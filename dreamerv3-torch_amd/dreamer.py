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


class _Accel:
    """The graph runners and the batch stager of one Dreamer.  No __dict__ (slots): the checkpoint's optimizer collector
    (tools.recursively_collect_optim_state_dict, dreamer.py:563-567) walks every attribute that has one, plain attributes
    first -- through `_runner.wm` it would reach the optimizers under paths the reference does not know
    ("_runner.wm._model_opt._opt") and a fresh agent cannot resolve (its runner does not exist yet)."""

    __slots__ = ("runner", "stager", "policy_runner")

    def __init__(self):
        self.runner = self.stager = self.policy_runner = None


def _accel_attr(name):
    return property(lambda self: getattr(self._accel, name), lambda self, v: setattr(self._accel, name, v))


class Dreamer(nn.Module):
    _runner, _stager, _policy_runner = _accel_attr("runner"), _accel_attr("stager"), _accel_attr("policy_runner")

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
        self._accel = _Accel()  # (._runner: graph.UpdateRunner, ._stager: staging.BatchStager, ._policy_runner: graph.PolicyRunner)
        self._macc = {}  # metric group ("wm" / "beh" / "expl") -> device-resident running sums (see _accumulate)

    @tools.on_config_device
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

    @tools.on_config_device
    def _policy(self, obs, state, training, noise=None):
        """Acting step.  With config.hip_graph (default) the launch sequence is replayed from a hipGraph per
        (number of envs, training) signature (dv3hip.graph.PolicyRunner); injected noise (tests) takes the eager path."""
        self._finish_updates()  # (a behaviour phase the update loop left pending trains the actor this step acts with)
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

    @tools.on_config_device
    def _train(self, data, pipelined=False):
        """dreamer.py:192-208.  One update = WorldModel._train + ImagBehavior._train on the updated world model; the
        launch sequence is replayed from hipGraphs (dv3hip.graph.UpdateRunner, ~1110 launches per update) once it has
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
            # the metrics of a phase are added up right behind its optimizer graph, on the stream that graph ran on
            self._runner.metric_sinks = (lambda m: self._accumulate(m, "wm"), lambda m: self._accumulate(m, "beh"))
        host = all(not isinstance(v, torch.Tensor) for v in data.values())
        explorer = self._expl_behavior is not self._task_behavior
        pipelined = pipelined and not explorer and bool(getattr(self._config, "pipeline_updates", True))
        # (the uploads go onto the stream the update is issued on; see UpdateRunner.launch_stream)
        with torch.cuda.stream(self._runner.launch_stream() or torch.cuda.current_stream()):
            staged = (self._stager.stage(data) if host else
                      {k: (v if k == "image" else v.to(torch.float32)) for k, v in data.items()})
            (self._runner.step_pipelined if pipelined else self._runner.step)(staged)
            if explorer:
                # dreamer.py:201-203: the explorer trains on the same posterior states (eagerly: its objective runs torch
                # autograd, which a hipGraph segment cannot hold)
                xm = self._expl_behavior.train(self._runner.last_post, self._runner.last_context,
                                               self._runner.last_data)[-1]
                self._accumulate({"expl_" + k: v for k, v in xm.items()}, "expl")

    def state_dict(self, *args, **kwargs):
        """The checkpoint of dreamer.py:563-567 holds the actor / critic of the LAST update: flush a pending phase first."""
        self._finish_updates()
        return super().state_dict(*args, **kwargs)

    @tools.on_config_device
    def _finish_updates(self):
        """End of a run of pipelined updates: issue the behaviour phase that is still pending."""
        r = self._runner
        if r is not None and r._pipe_pending:
            with torch.cuda.stream(r.launch_stream() or torch.cuda.current_stream()):
                r.flush()

    def _accumulate(self, mets, group):
        """Device-resident metrics -- the snapshots of the fused path (DeviceScalar) and the 0-d tensors an autograd-style
        caller's Optimizer.__call__ returns (the explorer) -- go into one running sum and one count per key, on the
        CURRENT stream: UpdateRunner calls this right behind the optimizer graph that wrote them (metric_sinks), so the
        two phases of the pipelined update add up their own keys on their own lanes and no queue waits for another."""
        dev = [(k, v._t if isinstance(v, models.DeviceScalar) else v.detach()) for k, v in mets.items()
               if isinstance(v, (models.DeviceScalar, torch.Tensor))]
        for k, v in mets.items():
            if not isinstance(v, (models.DeviceScalar, torch.Tensor)):
                self._metrics.setdefault(k, []).append(v)
        if not dev:
            return
        keys = tuple(k for k, _ in dev)
        acc = self._macc.get(group)
        if acc is None or acc["keys"] != keys:
            if acc is not None:
                self._flush_group(acc)  # (the key set of a group changed: rare -- settle what has been added up)
            d = dev[0][1].device
            acc = dict(keys=keys, sum=torch.zeros(len(keys), device=d, dtype=torch.float32), n=0)
            self._macc[group] = acc
        acc["sum"].add_(torch.stack([t.reshape(()).to(torch.float32) for _, t in dev]))
        acc["n"] += 1

    def _flush_group(self, acc):
        if acc["n"]:
            mean = (acc["sum"] / acc["n"]).cpu().numpy()
            for k, v in zip(acc["keys"], mean):
                self._metrics.setdefault(k, []).append(float(v))
            acc["sum"].zero_()
            acc["n"] = 0

    def _flush_metrics(self):
        """Device-side running sums -> self._metrics[name] = [mean] (what the reference's logging loop, which takes
        np.mean of each list, then reports): one device-to-host copy per metric group and log interval."""
        for acc in self._macc.values():
            self._flush_group(acc)
