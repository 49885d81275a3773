"""`exploration` module with the reference's surface (exploration.py:10-135): `Random` and `Plan2Explore`.

Plan2Explore trains an ensemble of one-step predictors on the replay batch and uses their disagreement on imagined
states as the reward of a second ImagBehavior.  It is written against the PUBLIC surface of this package only --
`networks.MLP(...)(inputs).log_prob / .mode`, `tools.Optimizer(...)(loss, params)`, `ImagBehavior._train(start,
objective)` -- exactly the calls the reference's own exploration.py makes, so either file drives the accelerated
modules: the ensemble runs on the HIP kernels forward and backward through dv3hip.autograd, and the intrinsic reward
enters the hand-written reverse imagination rollout through `ImagBehavior.train_fwd_bwd`'s objective hook.
State-dict keys (`_networks.<i>.layers...`, `_behavior.*`) are the reference's.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import distributions as torchd
from torch import nn

import models
import networks
import tools


class Random(nn.Module):
    """Uniform random policy (exploration.py:10-37): `actor(feat)` ignores feat and returns a distribution over one
    action per environment; `train` is a no-op."""

    def __init__(self, config, act_space):
        super().__init__()
        self._config, self._act_space = config, act_space

    def actor(self, feat):
        cfg = self._config
        if cfg.actor["dist"] == "onehot":
            return tools.OneHotDist(torch.zeros(cfg.envs, cfg.num_actions, device=cfg.device))
        bound = lambda b: torch.as_tensor(np.asarray(b), dtype=torch.float32, device=cfg.device).repeat(cfg.envs, 1)
        return torchd.independent.Independent(
            torchd.uniform.Uniform(bound(self._act_space.low), bound(self._act_space.high)), 1)

    def train(self, start, context, data):
        return None, {}


class Plan2Explore(nn.Module):
    def __init__(self, config, world_model, reward):
        super().__init__()
        if config.precision != 32:
            raise NotImplementedError("the path is fp32 (configs.yaml:18)")
        self._config, self._reward = config, reward
        self._behavior = models.ImagBehavior(config, world_model)
        self.actor = self._behavior.actor
        flat_stoch = config.dyn_stoch * (config.dyn_discrete or 1)
        feat_size = flat_stoch + config.dyn_deter
        # width of what the ensemble predicts; "feat" keeps the reference's (un-flattened) size, exploration.py:58-63
        target_size = dict(embed=world_model.embed_size, stoch=flat_stoch, deter=config.dyn_deter,
                           feat=config.dyn_stoch + config.dyn_deter)[config.disag_target]
        in_size = feat_size + (config.num_actions if config.disag_action_cond else 0)
        self._networks = nn.ModuleList(
            networks.MLP(inp_dim=in_size, shape=target_size, layers=config.disag_layers, units=config.disag_units,
                         act=config.act, device=config.device)
            for _ in range(config.disag_models))
        self._expl_opt = tools.Optimizer("explorer", self._networks.parameters(), config.model_lr, config.opt_eps,
                                         config.grad_clip, wd=config.weight_decay, opt=config.opt, use_amp=False)

    # -- one exploration update: ensemble regression on the replay batch, then the behaviour on its disagreement ----
    def train(self, start, context, data):
        cfg = self._config
        stoch = start["stoch"]
        if cfg.dyn_discrete:
            stoch = stoch.reshape(tuple(stoch.shape[:-2]) + (stoch.shape[-2] * stoch.shape[-1],))
        pick = dict(embed=lambda: context["embed"], stoch=lambda: stoch, deter=lambda: start["deter"],
                    feat=lambda: context["feat"])
        target = pick[cfg.disag_target]()
        inputs = context["feat"]
        if cfg.disag_action_cond:
            act = data["action"]
            act = act if isinstance(act, torch.Tensor) else torch.as_tensor(np.asarray(act))
            inputs = torch.cat([inputs, act.to(inputs.device, torch.float32)], -1)
        metrics = {}
        with tools.RequiresGrad(self._networks):
            metrics.update(self._train_ensemble(inputs, target))
        metrics.update(self._behavior._train(start, self._intrinsic_reward)[-1])
        return None, metrics

    def _train_ensemble(self, inputs, targets):
        """Each member maximises the likelihood of the (offset) target under its Normal head; one Adam step on the
        mean over members (exploration.py:123-135)."""
        off = self._config.disag_offset
        if off:
            targets, inputs = targets[:, off:], inputs[:, :-off]
        targets, inputs = targets.detach(), inputs.detach()
        like = torch.stack([head(inputs).log_prob(targets).mean() for head in self._networks])
        return self._expl_opt(-like.mean(), self._networks.parameters())

    def _intrinsic_reward(self, feat, state, action):
        """Disagreement = std over the members' predictions, averaged over the target dimensions (log'd with
        disag_log), scaled; plus the scaled extrinsic reward if configured (exploration.py:108-121)."""
        cfg = self._config
        x = torch.cat([feat, action], -1) if cfg.disag_action_cond else feat
        preds = torch.stack([head(x, torch.float32).mode() for head in self._networks], 0)
        disag = preds.std(0).mean(-1, keepdim=True)
        if cfg.disag_log:
            disag = disag.log()
        reward = cfg.expl_intr_scale * disag
        if cfg.expl_extr_scale:
            reward = reward + cfg.expl_extr_scale * self._reward(feat, state, action)
        return reward
