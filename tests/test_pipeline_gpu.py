"""The two-update software pipeline (dv3hip.graph.UpdateRunner.step_pipelined: the behaviour phase of update k beside
the world-model phase of update k+1) against the serial update sequence of the reference's loop (dreamer.py:95-97).

The pipeline must not change a number: it reads the same weights, the same posterior and the same Philox counters as
the serial order.  With the three learning rates at zero the weights stay put, every forward pass is deterministic and
each update's samples -- posterior states, imagined states, actions -- are compared BIT for bit; gradients within the
summation order of the reverse scan's atomic accumulations (as tests/test_fullsize_gpu.py::test_cu_lanes_...).  With
the real learning rates the two runs are compared like hipGraph replay against eager launches."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.golden import common

pytestmark = pytest.mark.gpu


def _batches(name, n):
    from dv3hip import shapes

    return [{k: torch.from_numpy(v).cuda() for k, v in shapes.synthetic_batch(name, seed).items()} for seed in range(n)]


def _run(name, n_calls, pipelined, lr_zero, pairs=False, plan=None):
    """n_calls updates on distinct batches -> per-update records.  warm=1: call 0 eager, from call 1 on hipGraph replay."""
    import tools
    from dv3hip.graph import UpdateRunner

    cfg, wm, beh = Hh.build_models(name)
    if lr_zero:
        for opt in (wm._model_opt, beh._actor_opt, beh._value_opt):
            opt._opt.param_groups[0]["lr"] = 0.0
    tools.default_rng("cuda:0", seed=7)
    r = UpdateRunner(wm, beh, warm=1)
    r.pipe_plan.update(plan or {})
    data = _batches(name, n_calls)
    rec = dict(post=[], im_stoch=[], im_action=[], g_model=[], g_actor=[], g_value=[], model_loss=[], actor_loss=[],
               value_loss=[])
    ws = wm.dynamics.engine.ws

    def grab_wm():
        torch.cuda.synchronize()
        rec["post"].append(r.last_post["stoch"].clone())
        rec["g_model"].append(wm._model_opt.bucket.grad.clone())
        rec["model_loss"].append(float(r.last_metrics["model_loss"]))

    def grab_beh():
        torch.cuda.synchronize()
        rec["im_stoch"].append(beh._im["stoch"].clone())
        rec["im_action"].append(beh._im["action"].clone())
        rec["g_actor"].append(beh._actor_opt.bucket.grad.clone())
        rec["g_value"].append(beh._value_opt.bucket.grad.clone())
        rec["actor_loss"].append(float(r.last_metrics["actor_loss"]))
        rec["value_loss"].append(float(r.last_metrics["value_loss"]))

    for i, d in enumerate(data):
        if not pipelined:
            r.step(d)
            grab_wm(), grab_beh()
            continue
        pending = r._pipe_pending
        r.step_pipelined(d)
        if pending:
            grab_beh()  # (the behaviour phase of the previous update ran beside this call's world-model phase)
        grab_wm()
        if not r._pipe_pending:
            grab_beh()  # (warm-up call: the whole update ran)
        elif pairs and i % 2 == 0:
            r.flush()
            grab_beh()
    if pipelined:
        was = r._pipe_pending
        r.flush()
        if was:
            grab_beh()
        assert r._pipe is not None, "the pipelined segments were never captured"
    assert r.use_graph, "hipGraph capture was refused"
    torch.cuda.synchronize()
    rec["params"] = {k: v.detach().clone() for k, v in list(wm.state_dict().items()) + list(beh.state_dict().items())}
    rec["ema"] = beh.ema_vals.clone()
    rec["rng"] = tools.default_rng("cuda:0").state.clone()
    return rec


PLANS = {"staged": dict(mode="staged", defer="post"), "staged_split": dict(mode="staged", defer="side", a_split=6, c_split=5),
         "lanes": dict(mode="lanes")}


@pytest.mark.parametrize("name,pairs,plan", [
    ("cfg2", False, "staged"), ("cfg2", True, "staged"), ("cfg2", False, "staged_split"), ("cfg2", False, "lanes"),
    ("cfg2", True, "lanes"), ("cfg3", False, "staged_split"), ("cfg3", False, "lanes"), ("cfg1", False, "lanes"),
    ("tiny", False, "staged"), ("tiny", True, "lanes")])
def test_pipelined_updates_draw_and_compute_what_the_serial_updates_do(name, pairs, plan):
    """Learning rates 0: every update's sampled states and actions bit-equal, gradients equal up to atomic order --
    for every schedule of the pipeline (graph._PIPE_PLAN)."""
    n = 6
    a = _run(name, n, pipelined=False, lr_zero=True)
    b = _run(name, n, pipelined=True, lr_zero=True, pairs=pairs, plan=PLANS[plan])
    assert torch.equal(a["rng"], b["rng"]), "the Philox stream ends elsewhere"
    for key in ("post", "im_stoch", "im_action"):
        assert len(a[key]) == len(b[key]) == n, (key, len(a[key]), len(b[key]))
        for i in range(n):
            assert torch.equal(a[key][i], b[key][i]), f"update {i}: {key} differs"
    for key in ("g_model", "g_actor", "g_value"):
        for i in range(n):
            scale = float(a[key][i].abs().max())
            err = float((a[key][i] - b[key][i]).abs().max())
            assert err <= 4e-6 * scale + 1e-12, f"update {i}: {key} differs by {err:.3e} (max |g| {scale:.3e})"
    for key in ("model_loss", "actor_loss", "value_loss"):
        np.testing.assert_allclose(b[key], a[key], rtol=2e-6, atol=1e-6, err_msg=key)
    assert torch.allclose(a["ema"], b["ema"], rtol=1e-6, atol=1e-7)
    for k, v in a["params"].items():
        if k.startswith("_slow_value."):
            assert torch.allclose(v, b["params"][k], rtol=0, atol=0), k  # (the slow critic moves: same EMA steps)
        else:
            assert torch.equal(v, b["params"][k]), k


@pytest.mark.parametrize("name", ["cfg2", "cfg3"])
def test_pipelined_updates_train_like_serial_updates(name):
    """Real learning rates: the two runs agree like hipGraph replay and eager launches do (the reverse scan's atomic
    summation order moves Adam's early, sign-like steps), and the model loss falls."""
    n = 6
    a = _run(name, n, pipelined=False, lr_zero=False)
    b = _run(name, n, pipelined=True, lr_zero=False, plan=PLANS["lanes"])
    assert torch.equal(a["rng"], b["rng"])
    for key, tol in (("model_loss", 2e-3), ("value_loss", 2e-2), ("actor_loss", 5e-2)):
        for x, y in zip(a[key], b[key]):
            assert abs(x - y) <= tol * max(1.0, abs(x)), (key, a[key], b[key])
    w0, w1 = a["params"]["dynamics.W"], b["params"]["dynamics.W"]
    assert float((w0 - w1).abs().max()) <= 1e-4 + 1e-4 * float(w0.abs().max())
    assert np.isfinite(b["model_loss"]).all() and np.isfinite(b["actor_loss"]).all()


class _Logger:
    def __init__(self):
        self.step, self.scalars = 0, {}

    def scalar(self, k, v):
        self.scalars[k] = v

    def video(self, *a, **k):
        pass

    def write(self, fps=False):
        pass


def test_dreamer_update_loop_pipelines_and_logs_like_the_serial_loop():
    """dreamer.Dreamer.__call__ (dreamer.py:87-106): the `pretrain` updates of one call go through step_pipelined, the
    last behaviour phase is flushed before the policy acts and before the logger reads, every metric key is the mean
    over exactly the updates of the interval, and the agent ends where the serial loop (`pipeline_updates: False`)
    ends up to the atomic summation order of the reverse scan."""
    import dreamer
    import tools

    name, n_upd = "tiny", 8
    res = []
    for pipelined in (False, True):
        cfg = Hh.make_config(name)
        cfg.pretrain, cfg.log_every, cfg.video_pred_log, cfg.train_ratio = n_upd, 1, False, 512
        cfg.pipeline_updates = pipelined
        logger = _Logger()

        def dataset():
            i = 0
            while True:
                yield common.make_batch(name, seed=i)
                i += 1

        torch.manual_seed(0)
        agent = dreamer.Dreamer(Hh.obs_space(name), None, cfg, logger, dataset()).to(cfg.device)
        agent.requires_grad_(False)
        tools.default_rng(cfg.device, seed=11)
        obs = {"image": np.zeros((2, 64, 64, 3), np.uint8), "is_first": np.ones(2, bool), "is_terminal": np.zeros(2, bool)}
        out, state = agent(obs, np.ones(2, bool), None, training=True)
        torch.cuda.synchronize()
        r = agent._runner
        assert agent._update_count == n_upd and not r._pipe_pending
        assert (r._pipe is not None) == pipelined, "the update loop did not take the pipeline"
        assert r.use_graph
        assert torch.isfinite(out["action"]).all()
        res.append((dict(logger.scalars), {k: v.detach().clone() for k, v in agent.state_dict().items()},
                    tools.default_rng(cfg.device).state.clone()))
    (m0, p0, g0), (m1, p1, g1) = res
    assert set(m0) == set(m1), set(m0) ^ set(m1)
    assert torch.equal(g0, g1), "the pipelined loop leaves the Philox stream elsewhere"
    for k in m0:
        assert abs(m0[k] - m1[k]) <= 2e-3 * max(1.0, abs(m0[k])), (k, m0[k], m1[k])
    for k in p0:
        d = (p0[k].double() - p1[k].double()).abs().max().item() if p0[k].numel() else 0.0
        assert d <= 3e-4 + 1e-3 * p0[k].double().abs().max().item(), (k, d)
