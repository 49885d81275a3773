"""Agent-level drop-in check: the reference's Dreamer class surface (dreamer.py:35-208) driven the way
tools.simulate drives it -- agent(obs, reset, state) -> (policy_output, state), training updates drawn from a
dataset iterator, metrics accumulated for the logger, checkpoint round trip through state_dict()."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.golden import common

pytestmark = pytest.mark.gpu


class _Logger:
    def __init__(self):
        self.step, self.scalars = 0, {}

    def scalar(self, k, v):
        self.scalars[k] = v

    def video(self, *a, **k):
        pass

    def write(self, fps=False):
        pass


def _dataset(name):
    i = 0
    while True:
        yield common.make_batch(name, seed=i)
        i += 1


def _obs(n_envs, first):
    rs = np.random.RandomState(0)
    return {"image": rs.randint(0, 256, (n_envs, 64, 64, 3)).astype(np.uint8),
            "is_first": np.full((n_envs,), first), "is_terminal": np.zeros((n_envs,), bool)}


@pytest.mark.parametrize("name", ["tiny", "tiny_onehot"])
def test_dreamer_agent_trains_and_acts(name):
    import dreamer

    cfg = Hh.make_config(name)
    cfg.pretrain, cfg.log_every, cfg.video_pred_log, cfg.train_ratio = 3, 1, False, 512
    logger = _Logger()
    agent = dreamer.Dreamer(Hh.obs_space(name), None, cfg, logger, _dataset(name)).to(cfg.device)
    agent.requires_grad_(False)
    n_envs, A = 2, common.SHAPES[name]["A"]
    reset = np.ones(n_envs, bool)
    out, state = agent(_obs(n_envs, True), reset, None, training=True)  # runs `pretrain` updates, then acts
    assert agent._update_count == 3
    assert out["action"].shape == (n_envs, A) and out["logprob"].shape == (n_envs,)
    assert torch.isfinite(out["action"]).all() and torch.isfinite(out["logprob"]).all()
    latent, action = state
    assert set(latent) == {"stoch", "deter", "logit"} and latent["stoch"].sum(-1).eq(1).all()
    # the logger received the reference's metric keys
    for k in ("model_loss", "model_grad_norm", "image_loss", "reward_loss", "cont_loss", "kl", "prior_ent",
              "post_ent", "actor_loss", "value_loss", "actor_grad_norm", "value_grad_norm", "actor_entropy",
              "EMA_005", "EMA_095", "update_count"):
        assert k in logger.scalars and np.isfinite(logger.scalars[k]), k
    # carry the state through further env steps (no reset), eval mode uses the mode of the actor
    for _ in range(3):
        out, state = agent(_obs(n_envs, False), np.zeros(n_envs, bool), state, training=False)
    assert torch.isfinite(out["action"]).all()
    if name == "tiny_onehot":
        assert out["action"].sum(-1).eq(1).all()
    else:
        assert out["action"].abs().max() <= 1.0 + 1e-6
    # checkpoint round trip: same keys as the reference's agent_state_dict, loads into a fresh agent
    sd = agent.state_dict()
    assert "_wm.dynamics._cell.layers.GRU_linear.weight" in sd and "_task_behavior.ema_vals" in sd
    assert "_task_behavior._world_model.dynamics.W" in sd  # the reference's aliasing of the world model
    agent2 = dreamer.Dreamer(Hh.obs_space(name), None, cfg, _Logger(), _dataset(name)).to(cfg.device)
    agent2.load_state_dict(sd)
    for k, v in agent2.state_dict().items():
        assert torch.equal(v, sd[k]), k


def _load_agent(name, n_envs=None):
    import dreamer

    cfg = Hh.make_config(name)
    cfg.pretrain = 0
    agent = dreamer.Dreamer(Hh.obs_space(name), None, cfg, _Logger(), _dataset(name)).to(cfg.device)
    w = common.make_weights(name)
    sd = agent.state_dict()
    for k in sd:
        key = k.replace("_wm.", "", 1) if k.startswith("_wm.") else k.replace("_task_behavior.", "", 1)
        if key.startswith("_world_model."):
            key = key[len("_world_model."):]
        if key in w:
            sd[k] = torch.from_numpy(w[key])
    agent.load_state_dict(sd)
    agent.requires_grad_(False)
    return agent, w


@pytest.mark.parametrize("name,n_envs", [("tiny", 3), ("tiny_onehot", 3), ("cfg2", 1), ("cfg2", 16)])
def test_policy_steps_match_oracle(name, n_envs):
    """The acting path: Dreamer._policy -> preprocess -> encoder -> obs_step -> actor, three
    consecutive env steps against the oracle on the same weights and injected noise:
      step 0  no carried state (all is_first), training: sampled posterior, sampled action, its log-prob;
      step 1  carried state, no reset, training;
      step 2  carried state, a reset on one env only (the reset_blend of state and action, networks.py:183-191),
              evaluation: posterior still sampled (obs_step's default), action = mode of the actor."""
    from oracle import dv3_oracle as O

    agent, w = _load_agent(name)
    s = common.SHAPES[name]
    S, D, A = s["stoch"], s["discrete"], s["A"]
    pc = common.path_config(name)
    p = {k: torch.from_numpy(v) for k, v in w.items()}
    rs = np.random.RandomState(5)
    state, ostate, oaction = None, None, None
    for step in range(3):
        first = np.ones(n_envs, bool) if step == 0 else np.zeros(n_envs, bool)
        if step == 2:
            first[n_envs // 2] = True
        obs = {"image": rs.randint(0, 256, (n_envs, 64, 64, 3)).astype(np.uint8), "is_first": first,
               "is_terminal": np.zeros(n_envs, bool)}
        qp = np.maximum(rs.exponential(size=(n_envs, S, D)), 1e-20).astype(np.float32)
        qq = np.maximum(rs.exponential(size=(n_envs, S, D)), 1e-20).astype(np.float32)
        onehot = s["actor_dist"] == "onehot"
        eps = (np.maximum(rs.exponential(size=(n_envs, A)), 1e-20) if onehot else rs.randn(n_envs, A)).astype(np.float32)
        training = step < 2
        noise = dict(prior=torch.from_numpy(qp).cuda(), post=torch.from_numpy(qq).cuda(), act=torch.from_numpy(eps).cuda())
        out, state = agent._policy(obs, state, training, noise=noise)
        # oracle
        img = torch.from_numpy(obs["image"]).float() / 255.0
        embed = O.conv_encoder(pc, p, img[:, None])[:, 0]
        post, _ = O.obs_step(pc, p, ostate, oaction, embed, torch.from_numpy(first).float(), torch.from_numpy(qp),
                             torch.from_numpy(qq))
        feat = O.get_feat(pc, post)
        if training:
            act = O.actor_sample(pc, p, feat, torch.from_numpy(eps))
        elif onehot:
            act = O.onehot_mode(O.actor_stats(pc, p, feat)[0], pc.unimix)
        else:
            act = O.actor_stats(pc, p, feat)[0]  # tanh(mean); |.| <= 1 so the absmax rescale is the identity
        lp = O.actor_logprob(pc, p, feat, act)
        latent, action = state
        what = f"{name} x{n_envs} step {step}"
        assert torch.equal(latent["stoch"].cpu(), post["stoch"]), what + ": sampled posterior"
        assert torch.allclose(latent["deter"].cpu(), post["deter"], atol=1e-4), what + ": deter"
        assert torch.allclose(latent["logit"].cpu(), post["logit"], atol=1e-4), what + ": logit"
        assert out["action"].shape == act.shape and torch.allclose(out["action"].cpu(), act, atol=1e-4), what + ": action"
        assert torch.allclose(out["logprob"].cpu(), lp, atol=2e-4, rtol=1e-4), what + ": logprob"
        ostate, oaction = {k: v.detach() for k, v in post.items()}, act.detach()


@pytest.mark.parametrize("name,n_envs", [("tiny", 3), ("tiny_onehot", 2), ("cfg2", 4)])
def test_policy_graph_replay_equals_eager(name, n_envs):
    """The hipGraph-replayed acting step (dv3hip.graph.PolicyRunner) returns what the eager launch sequence returns,
    call after call, with the state carried and a reset in between (same Philox stream position for both)."""
    import tools

    agent, _ = _load_agent(name)
    rs = np.random.RandomState(9)
    seq = []
    for step in range(4):
        first = np.ones(n_envs, bool) if step == 0 else np.zeros(n_envs, bool)
        if step == 2:
            first[0] = True
        seq.append({"image": rs.randint(0, 256, (n_envs, 64, 64, 3)).astype(np.uint8), "is_first": first,
                    "is_terminal": np.zeros(n_envs, bool)})
    outs = {}
    for mode in ("eager", "graph"):
        tools.default_rng(agent._config.device, seed=21)
        state, res = None, []
        for step, obs in enumerate(seq):
            training = step != 3
            if mode == "eager":
                out, state = agent._policy_eager(obs, state, training)
            else:
                out, state = agent._policy(obs, state, training)
            res.append((out["action"].clone(), out["logprob"].clone(), {k: v.clone() for k, v in state[0].items()}))
        outs[mode] = res
    assert agent._policy_runner not in (None, False) and len(agent._policy_runner._sig) == 2
    for (a0, l0, s0), (a1, l1, s1) in zip(outs["eager"], outs["graph"]):
        assert torch.equal(s0["stoch"], s1["stoch"]) and torch.allclose(s0["deter"], s1["deter"], atol=1e-6)
        assert torch.allclose(a0, a1, atol=1e-6) and torch.allclose(l0, l1, atol=1e-5)


def test_policy_graph_captured_before_the_first_update_follows_the_training():
    """The reference's loop EVALUATES before it trains (dreamer.py:534-549), so the acting step is captured while the
    optimizers' flat parameter buckets -- built lazily by the first update -- do not exist yet.  The graph holds raw
    pointers to the weights: PolicyRunner has to settle the weights into their buckets before it captures, or its
    replays would act with (and, after the next capture's empty_cache, fault on) the storage the weights have left.
    A later move of the weights (Module.to) drops the acting graphs; the update's graphs refuse to replay."""
    import tools

    agent, _ = _load_agent("tiny")
    obs = _obs(2, True)
    reseed = lambda: tools.default_rng(agent._config.device, seed=5)  # (the posterior state is a sample)
    reseed()
    out0, _ = agent._policy(obs, None, training=False)  # (the very first thing the agent is asked to do)
    runner = agent._policy_runner
    assert runner not in (None, False) and runner._buckets and all(b.settled() for b in runner._buckets)
    where = {k: v.data_ptr() for k, v in agent.named_parameters()}
    ds = _dataset("tiny")
    for _ in range(6):  # eager warm-up, serial graphs, pipelined segments: each capture empties the allocator's cache
        agent._train(next(ds), pipelined=True)
    agent._finish_updates()
    torch.cuda.synchronize()
    assert agent._runner.use_graph and agent._runner._g_wm is not None
    # (the slow critic is flattened by the first update as well: training reads it, the acting step does not)
    moved = [k for k, v in agent.named_parameters() if v.data_ptr() != where[k] and "_slow_value." not in k]
    assert not moved, f"the first update moved weights the acting graph reads: {moved}"
    reseed()
    out_g, _ = agent._policy(obs, None, training=False)
    reseed()
    out_e, _ = agent._policy_eager(obs, None, training=False)
    assert len(runner._sig) == 1, "the acting graph was rebuilt"
    assert torch.allclose(out_g["action"], out_e["action"], atol=1e-6), "the replay acts with stale weights"
    assert not torch.allclose(out_g["action"], out0["action"], atol=1e-6), "six updates left the actor's mode where it was"
    # the weights move: the acting graphs are rebuilt over the new storage, the update refuses to replay
    for prm in agent._task_behavior.actor.parameters():
        prm.data = prm.data.clone()
    reseed()
    out_m, _ = agent._policy(obs, None, training=False)
    assert all(b.settled() for b in runner._buckets)
    assert torch.allclose(out_m["action"], out_e["action"], atol=1e-6)
    with pytest.raises(RuntimeError, match="moved after"):
        agent._train(next(ds))


def test_resumed_agent_continues_where_the_checkpointed_one_does(tmp_path):
    """The reference's resume order (dreamer.py:502-506, 534-560): fresh agent -> load_state_dict ->
    recursively_load_optim_state_dict -> _should_pretrain._once = False -> EVALUATE -> train.  The resumed agent's next
    updates (pipelined, hipGraph replay) end where the original agent's same updates end, up to the reverse scan's
    atomic summation order."""
    import dreamer
    import tools

    name, n_envs = "tiny", 2
    cfg = Hh.make_config(name)
    cfg.pretrain, cfg.log_every, cfg.video_pred_log = 0, 1e9, False
    batches = [common.make_batch(name, seed=i) for i in range(10)]

    def make():
        torch.manual_seed(3)
        ag = dreamer.Dreamer(Hh.obs_space(name), None, cfg, _Logger(), iter(())).to(cfg.device)
        ag.requires_grad_(False)
        return ag

    def train(ag, lo, hi):
        for i in range(lo, hi):
            ag._train(batches[i], pipelined=True)
        ag._finish_updates()
        torch.cuda.synchronize()

    rng = tools.default_rng(cfg.device, seed=17)
    a = make()
    train(a, 0, 5)
    ckpt = {"agent_state_dict": {k: v.detach().clone() for k, v in a.state_dict().items()},
            "optims_state_dict": tools.recursively_collect_optim_state_dict(a)}
    # the reference's attribute paths (what its own checkpoints hold), although the runners hold the same optimizers
    assert set(ckpt["optims_state_dict"]) == {"_wm._model_opt._opt", "_task_behavior._actor_opt._opt",
                                              "_task_behavior._value_opt._opt"}
    torch.save(ckpt, tmp_path / "latest.pt")  # (dreamer.py:563-567 writes it, :503 reads it back)
    ckpt = torch.load(tmp_path / "latest.pt")
    rng_at = rng.state.clone()
    train(a, 5, 10)
    want = {k: v.detach().clone() for k, v in a.state_dict().items()}
    b = make()
    b.load_state_dict(ckpt["agent_state_dict"])
    tools.recursively_load_optim_state_dict(b, ckpt["optims_state_dict"])
    b._should_pretrain._once = False
    out, state = b(_obs(n_envs, True), np.ones(n_envs, bool), None, training=False)  # (evaluation comes first)
    out, state = b(_obs(n_envs, False), np.zeros(n_envs, bool), state, training=False)
    assert torch.isfinite(out["action"]).all() and b._update_count == 0
    rng.state.copy_(rng_at)  # (the evaluation drew its posterior samples from the shared Philox stream)
    train(b, 5, 10)
    assert b._runner.use_graph and b._runner._g_wm is not None
    got = b.state_dict()
    assert set(got) == set(want)
    for k in want:
        d = (want[k].double() - got[k].double()).abs().max().item() if want[k].numel() else 0.0
        assert d <= 3e-4 + 1e-3 * want[k].double().abs().max().item(), (k, d)
    # and it acts like the original agent after the same updates
    rng.state.copy_(rng_at)
    oa, _ = a._policy(_obs(n_envs, True), None, training=False)
    rng.state.copy_(rng_at)
    ob, _ = b._policy(_obs(n_envs, True), None, training=False)
    assert torch.allclose(oa["action"], ob["action"], atol=5e-3)


def test_policy_takes_the_observation_dict_tools_simulate_builds():
    """tools.simulate (tools.py:163-166) stacks EVERY key of the env's observation that does not start with "log_": beside
    image / is_first / is_terminal that is `reward`, `is_last`, `discount` and the proprioceptive vectors the vision
    config's encoder ignores -- float64 from dm_control.  The replayed acting step takes them like the eager one."""
    import tools

    agent, _ = _load_agent("tiny")
    rs = np.random.RandomState(5)
    n = 3

    def obs(first):
        o = _obs(n, first)
        o.update(reward=rs.randn(n), is_last=np.zeros(n, bool), discount=np.ones(n), orientations=rs.randn(n, 14),
                 velocity=rs.randn(n, 9).astype(np.float64), height=rs.randn(n))
        return o

    o0, o1 = obs(True), obs(False)
    res = {}
    for mode in ("eager", "graph"):
        tools.default_rng(agent._config.device, seed=8)
        fn = agent._policy_eager if mode == "eager" else agent._policy
        out, state = fn(o0, None, True)
        out, state = fn(o1, state, True)
        res[mode] = (out["action"].clone(), state[0]["stoch"].clone())
    assert agent._policy_runner not in (None, False), "the acting step fell back to eager launches"
    assert torch.equal(res["eager"][1], res["graph"][1])
    assert torch.allclose(res["eager"][0], res["graph"][0], atol=1e-6)


def test_agent_lifecycle_replayed_equals_eager():
    """One agent driven through the interleavings a training run produces -- evaluate, train (three, two, one update per
    call), act with another number of envs, video_pred, checkpoint into the live agent, evaluate again -- once with hipGraph
    replay + the two-update pipeline (the defaults) and once with every launch eager and serial (`hip_graph: False`): the
    two runs end in the same Philox position and, up to the reverse scan's atomic summation order, the same weights."""
    import dreamer
    import tools

    name = "tiny"
    res = []
    for replay in (True, False):
        cfg = Hh.make_config(name)
        cfg.pretrain, cfg.log_every, cfg.video_pred_log = 0, 1e9, False
        cfg.hip_graph, cfg.pipeline_updates = replay, replay
        torch.manual_seed(1)
        agent = dreamer.Dreamer(Hh.obs_space(name), None, cfg, _Logger(), _dataset(name)).to(cfg.device)
        agent.requires_grad_(False)
        rng = tools.default_rng(cfg.device, seed=23)
        per_call = [3, 2, 1, 2]
        agent._should_train = lambda step: per_call.pop(0) if per_call else 0
        agent._should_pretrain._once = False
        acts = []

        def evaluate(n):
            out, st = agent(_obs(n, True), np.ones(n, bool), None, training=False)
            out, st = agent(_obs(n, False), np.zeros(n, bool), st, training=False)
            acts.append(out["action"].clone())

        evaluate(2)
        out, st = agent(_obs(3, True), np.ones(3, bool), None, training=True)       # 3 updates, then acts
        out, st = agent(_obs(3, False), np.zeros(3, bool), st, training=True)        # 2 updates
        acts.append(out["action"].clone())
        video = agent._wm.video_pred(common.make_batch(name, seed=99))
        assert torch.isfinite(video).all()
        sd = {k: v.detach().clone() for k, v in agent.state_dict().items()}
        osd = tools.recursively_collect_optim_state_dict(agent)
        agent.load_state_dict(sd)                                                  # (into the live agent: in place)
        tools.recursively_load_optim_state_dict(agent, osd)
        evaluate(2)
        out, st = agent(_obs(3, False), np.zeros(3, bool), st, training=True)        # 1 update (not pipelined)
        out, st = agent(_obs(4, True), np.ones(4, bool), None, training=True)        # 2 updates, another number of envs
        acts.append(out["action"].clone())
        evaluate(2)
        torch.cuda.synchronize()
        assert agent._update_count == 8
        if replay:
            assert agent._runner.use_graph and agent._runner._pipe is not None and agent._policy_runner not in (None, False)
        else:
            assert agent._runner._g_wm is None and agent._policy_runner in (None, False)
        res.append((rng.state.clone(), {k: v.detach().clone() for k, v in agent.state_dict().items()}, acts))
    (r0, p0, a0), (r1, p1, a1) = res
    assert torch.equal(r0, r1), "the replayed run leaves the Philox stream elsewhere"
    for k in p0:
        d = (p0[k].double() - p1[k].double()).abs().max().item() if p0[k].numel() else 0.0
        assert d <= 3e-4 + 1e-3 * p0[k].double().abs().max().item(), (k, d)
    for x, y in zip(a0, a1):
        assert x.shape == y.shape and torch.allclose(x, y, atol=2e-2), float((x - y).abs().max())


def test_policy_graph_sees_a_state_the_caller_rewrote():
    """PolicyRunner keeps the carried state where the previous replay left it and skips the copy-in when the caller
    hands back exactly what it was given; a state edited in place (or any other tensors) must be copied in."""
    import tools

    agent, _ = _load_agent("tiny")
    rs = np.random.RandomState(3)
    mk = lambda first: {"image": rs.randint(0, 256, (2, 64, 64, 3)).astype(np.uint8),
                        "is_first": np.full(2, first, bool), "is_terminal": np.zeros(2, bool)}
    o0, o1, o2 = mk(True), mk(False), mk(False)
    tools.default_rng(agent._config.device, seed=4)
    _, state = agent._policy(o0, None, True)
    _, state = agent._policy(o1, state, True)  # untouched hand-back: no copy
    runner = agent._policy_runner
    st = next(iter(runner._sig.values()))
    assert runner._is_last_output(st, state)
    state[0]["deter"].mul_(0.5)  # in place: same storage, new version
    assert not runner._is_last_output(st, state)
    edited = ({k: v.clone() for k, v in state[0].items()}, state[1].clone())
    saved = tools.default_rng(agent._config.device).state.clone()
    out_g, _ = agent._policy(o2, state, True)
    tools.default_rng(agent._config.device).state.copy_(saved)
    out_e, _ = agent._policy_eager(o2, edited, True)
    assert torch.allclose(out_g["action"], out_e["action"], atol=1e-6)
    assert torch.allclose(out_g["logprob"], out_e["logprob"], atol=1e-5)


@pytest.mark.parametrize("name", ["tiny", "cfg2"])
def test_video_pred_matches_reference_golden(name):
    """WorldModel.video_pred (models.py:192-213) on the GPU path, same weights / batch / noise,
    against the video the reference itself produced (tests/golden/*_video.npz).  fp32, tolerance 1e-4."""
    import os

    import numpy as np

    path = os.path.join(os.path.dirname(__file__), "golden", name + "_video.npz")
    g = np.load(path, allow_pickle=False)
    _, wm, _ = Hh.build_models(name)
    noise = {k: torch.from_numpy(v).cuda() for k, v in common.make_video_noise(name).items()}
    video = wm.video_pred(common.make_batch(name), noise=noise).cpu().numpy()
    assert tuple(video.shape) == tuple(g["meta/shape"])
    if "video" in g.files:
        assert np.abs(video - g["video"]).max() <= 1e-4
    else:
        T = video.shape[1]
        assert np.abs(video[0, [0, 4, 5, T - 1]] - g["video_rows"]).max() <= 1e-4
    ref = g["sum/video"]
    got = common.checksum(video)
    assert abs(got[0] - ref[0]) <= 1e-4 * ref[1] and abs(got[1] - ref[1]) <= 1e-4 * ref[1]


@pytest.mark.parametrize("overlap", [False, True])
def test_batch_stager_matches_direct_upload_and_trains(overlap):
    """Pinned double-buffered staging: what lands in HBM is the host batch (uint8 images, flags
    as float), slots are recycled, and an update on a staged batch equals the update on the host batch -- with the
    uploads on the update's own stream (default) and on a separate copy stream."""
    from dv3hip.staging import BatchStager

    name = "tiny"
    stager = BatchStager("cuda:0", depth=2, overlap=overlap)
    batches = [common.make_batch(name, seed=s) for s in range(3)]
    for b in batches:  # 3 batches through 2 slots
        d = stager.stage(b)
        torch.cuda.synchronize()
        assert d["image"].dtype == torch.uint8 and d["is_first"].dtype == torch.float32
        for k, v in b.items():
            want = torch.from_numpy(np.ascontiguousarray(v))
            assert torch.equal(d[k].cpu().to(want.dtype), want), k
    noise = {k: torch.from_numpy(v).cuda() for k, v in common.make_noise(name).items()}
    outs = []
    for staged in (False, True):
        _, wm, _ = Hh.build_models(name)
        data = stager.stage(batches[0]) if staged else batches[0]
        post, _, mets = wm._train(data, noise=noise)
        outs.append((post["deter"].clone(), float(mets["model_loss"]),
                     torch.cat([p.detach().reshape(-1) for p in wm.parameters()]).clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]  # forward: bit-equal
    # the backward adds split-K partial sums with atomics (order not fixed): parameters agree to fp32 rounding
    # (Adam's first step is lr * g / (|g| + eps): where a gradient is ~0 a rounding-level difference moves a weight
    # by a fraction of lr, so: almost all within 1e-6, none further than 2 * lr apart)
    diff = (outs[0][2] - outs[1][2]).abs()
    assert float((diff > 1e-6).float().mean()) < 2e-3 and float(diff.max()) <= 2.1e-4


def test_optimizer_state_is_interchangeable_with_torch_adam():
    """Checkpoint contract (dreamer.py:563-567, tools.py:975-1011): `optims_state_dict` holds torch.optim.Adam
    state_dicts keyed by attribute path.  The flat-bucket optimizer (a) steps like clip_grad_norm_ + Adam,
    (b) serialises to that format, (c) resumes from a state_dict produced by torch.optim.Adam itself."""
    import tools

    name = "tiny"
    noise = {k: torch.from_numpy(v).cuda() for k, v in common.make_noise(name).items()}
    _, wm, _ = Hh.build_models(name)
    cfg = wm._config
    names = [n for n, _ in wm.named_parameters()]
    ref_params = [torch.nn.Parameter(p.detach().clone()) for p in wm.parameters()]
    ref = torch.optim.Adam(ref_params, lr=cfg.model_lr, eps=cfg.opt_eps)

    def both_step(seed):
        wm._train(common.make_batch(name, seed=seed), noise=noise)
        for rp, p in zip(ref_params, wm.parameters()):
            rp.grad = p.grad.detach().clone()  # the gradient the bucket just consumed (kept until the next begin())
        torch.nn.utils.clip_grad_norm_(ref_params, cfg.grad_clip)
        ref.step()

    both_step(0)
    for n, rp, p in zip(names, ref_params, wm.parameters()):
        assert torch.allclose(rp, p, rtol=0, atol=1e-6), n
    # (b) format: same structure and values as torch.optim.Adam's own state_dict
    mine, theirs = wm._model_opt.state_dict(), ref.state_dict()
    assert set(mine) == set(theirs) and set(mine["state"]) == set(theirs["state"])
    assert mine["param_groups"][0]["params"] == theirs["param_groups"][0]["params"]
    for k in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
        assert mine["param_groups"][0][k] == theirs["param_groups"][0][k], k
    for i in theirs["state"]:
        assert float(mine["state"][i]["step"]) == float(theirs["state"][i]["step"]) == 1.0
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.allclose(mine["state"][i][k], theirs["state"][i][k], rtol=1e-5, atol=1e-10), (i, k)
    # the reference's collector finds it under the reference's attribute path
    agent = type("Agent", (), {})()
    agent._wm = wm
    collected = tools.recursively_collect_optim_state_dict(agent)
    assert list(collected) == ["_wm._model_opt._opt"]
    # (c) resume a FRESH model from torch.optim.Adam's state (+ the reference-format weights), then step both again
    _, wm2, _ = Hh.build_models(name)
    wm2.load_state_dict({n: rp.detach() for n, rp in zip(names, ref_params)})
    agent2 = type("Agent", (), {})()
    agent2._wm = wm2
    tools.recursively_load_optim_state_dict(agent2, {"_wm._model_opt._opt": ref.state_dict()})
    wm_old, wm = wm, wm2
    both_step(1)
    for n, rp, p in zip(names, ref_params, wm2.parameters()):
        assert torch.allclose(rp, p, rtol=0, atol=2e-6), n
    assert float(wm2._model_opt.state_dict()["state"][0]["step"]) == 2.0
