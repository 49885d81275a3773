"""Full-size parity on the MI355X for the BASELINE configs that have a reference fixture
(cfg 1 dmc_proprio, cfg 2 dmc_vision, cfg 3 atari100k shapes: tests/golden/cfg{1,2,3}.npz; and the crafter-width
layers of cfg 4 / cfg 5 -- cnn_depth 96, deter 4096 / 2048, hidden = units = 1024, five-layer heads -- at the batch
the reference finishes in a minute on the build container's CPU: tests/golden/cfg{4,5}_b4.npz):

* against the golden vectors the REFERENCE produced for the config (world-model outputs, imagination rows,
  lambda-returns, per-frame losses, scalar losses, gradient and post-Adam parameter checksums),
* against the CPU oracle run live on the same minibatch / weights / noise (every tensor, every gradient, the
  Adam-updated parameters),
* free-running (no teacher forcing): rows compared up to their first sample flip, flip count printed and bounded,
* through size-independent properties (batch-row permutation equivariance, hipGraph replay == eager).

Sampling is argmax(p/q): an ulp-level difference in p can flip a draw and that row's future then differs
(SURVEY.md section 7.3).  The value comparisons therefore run TEACHER-FORCED: the kernels still make their own
draw from the same noise, but the state follows the oracle's draw and every disagreement is counted
(dv3_onehot_sample_fwd_ex).  Every assert below is unconditional; the per-draw flip rate is asserted <= 1e-5 and
printed, so a kernel that drifts shows up as flips, not as a skipped block.
"""
import os

import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.golden import common

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CONFIGS = ["cfg2", "cfg1", "cfg3"]
REF_CONFIGS = CONFIGS + ["cfg4_b4", "cfg5_b4"]
# Teacher-forced sample flips measured on the MI355X (run A: the forward kernels are deterministic, so these are
# exact): asserted, not just printed.  cfg 3's one flip in 1 079 296 draws is an argmax tie within one ulp.
EXPECTED_FLIPS = {"cfg1": (0, 0), "cfg2": (0, 0), "cfg3": (1, 1), "cfg4_b4": (0, 0), "cfg5_b4": (0, 0)}  # (wm, im) upper bounds
# fp32 tolerances (north_star: outputs within 1e-4), relative to max(1, max|ref|) of the tensor compared
TOL_OUT = 1e-4  # logits, deter, pixels, returns, values, rewards, actions
TOL_LOSS = 2e-5  # scalar losses (means over >= 1024 rows) against the live oracle
TOL_LOSS_REF = 5e-5  # ... against the reference's own float32 scalar (its CPU summation order differs again)
TOL_GRAD = 1e-3  # gradients, relative to the gradient tensor's own max (sums over 1024..15360 rows / 4M pixels)
MAX_FLIP_RATE = 1e-5  # per categorical draw


def close(got, ref, tol, what, floor=1.0):
    got = torch.as_tensor(np.asarray(got) if not isinstance(got, torch.Tensor) else got).detach().cpu().double()
    ref = torch.as_tensor(np.asarray(ref) if not isinstance(ref, torch.Tensor) else ref).detach().cpu().double()
    assert got.shape == ref.shape, (what, tuple(got.shape), tuple(ref.shape))
    err = (got - ref).abs().max().item() if got.numel() else 0.0
    scale = max(floor, ref.abs().max().item() if ref.numel() else floor)
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol:g} * scale {scale:.3e}"
    return err / scale


def checksum_close(got, ref_cs, tol, what):
    """got: tensor; ref_cs: common.checksum of the reference tensor (sum, abs-sum, max-abs in float64)."""
    cs = common.checksum(got.detach().cpu().numpy())
    scale = max(ref_cs[1], 1e-12)
    assert abs(cs[1] - ref_cs[1]) <= tol * scale, f"{what}: abs-sum {cs[1]:.9e} vs {ref_cs[1]:.9e}"
    assert abs(cs[0] - ref_cs[0]) <= tol * scale, f"{what}: sum {cs[0]:.9e} vs {ref_cs[0]:.9e}"
    assert abs(cs[2] - ref_cs[2]) <= max(tol * 10, 1e-3) * max(ref_cs[2], 1e-12), f"{what}: max {cs[2]} vs {ref_cs[2]}"


def adam_close(got, ref, lr, what):
    """Post-Adam parameters: equal to 2e-6 except where the first Adam step (lr * g / (|g| + eps)) turns a
    rounding-level difference of a near-zero gradient into a fraction of lr -- rare, bounded by 2 lr."""
    d = (got.detach().cpu().double() - ref.detach().cpu().double()).abs()
    assert float(d.max()) <= 2.1 * lr, f"{what}: max {float(d.max()):.3e}"
    assert float((d > 2e-6).double().mean()) <= 5e-3, f"{what}: {float((d > 2e-6).double().mean()):.3e} outliers"


def gpu_noise(name, exp, beh_key, flips):
    s = common.SHAPES[name]
    n = {k: torch.from_numpy(v).cuda() for k, v in common.make_noise(name).items()}
    wm_force, im_force = Hh.forced_draws(name, exp, beh_key)
    wm_noise = dict(q_prior=n["q_prior"].contiguous(), q_post=n["q_post"].contiguous(), flips=flips, **wm_force)
    im_noise = dict(act=Hh.to_time_major_rows(n["act"], s["B"], s["T"]).contiguous(),
                    q_img=Hh.to_time_major_rows(n["q_img"], s["B"], s["T"]).contiguous(), flips=flips, **im_force)
    return wm_noise, im_noise


def n_draws(s, onehot_actor):
    wm = 2 * s["T"] * s["B"] * s["stoch"]
    im = (s["H"] - 1) * s["B"] * s["T"] * s["stoch"] + (s["H"] * s["B"] * s["T"] if onehot_actor else 0)
    return wm, im


@pytest.fixture(scope="module", params=REF_CONFIGS)
def full(request):
    """Oracle expectation + two teacher-forced GPU runs: (A) the pieces, world-model forward/backward and the
    behaviour losses on the NOT yet updated world model (what the golden file's imag/*, grad/* hold); (B) the
    full update through _train (what its train/* and after/* hold)."""
    name = request.param
    s = common.SHAPES[name]
    exp = Hh.oracle_update(name, threads=min(16, os.cpu_count() or 1), piecewise=True)
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    data = common.make_batch(name)
    onehot_actor = s["actor_dist"] == "onehot"
    # ---- run A
    cfg, wm, beh = Hh.build_models(name)
    flipsA = torch.zeros(1, dtype=torch.int32, device="cuda")
    wm_noise, im_noise = gpu_noise(name, exp, "beh0", flipsA)
    wm.train_fwd_bwd(data, noise=wm_noise)
    post, _, wm_mets, wm_loss = wm._pending
    A = dict(post={k: v.clone() for k, v in post.items()}, wm_loss=wm_loss.clone(),
             wm_mets={k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in wm_mets.items()},
             wm_grads={k: p.grad.clone() for k, p in wm.named_parameters()},
             out={k: v.clone() for k, v in wm._last["out"].items()}, embed=wm._last["embed"].clone(),
             kl=wm._last["kl"].clone(), ent_post=wm._last["ent_post"].clone())
    ws = wm.dynamics.engine.ws
    TB = s["B"] * s["T"]
    A["lp_r"], A["lp_c"] = ws.get("wm.lp_r", (TB,)).clone(), ws.get("wm.lp_c", (TB,)).clone()
    if s["encoder"] == "cnn":
        A["recon"] = wm.heads["decoder"]._cnn.engine.ws.get("dec.recon", (TB, 64, 64, 3)).clone()
        A["loss_img"] = ws.get("wm.loss_img", (TB,)).clone()
    else:
        A["loss_vec"] = {k: ws.get(f"wm.loss.{k}", (TB,)).clone() for k, _ in common.PROPRIO_KEYS}
    wm_flips = int(flipsA.item())
    beh._update_slow_target = lambda: None  # the golden pieces use the critic's slow copy as initialised
    beh.train_fwd_bwd(A["post"], noise=im_noise)
    (_, imag_state, action, weights), beh_mets, (aloss, vloss) = beh._pending
    A.update(imag={k: v.clone() for k, v in imag_state.items()}, action=action.clone(), weights=weights.clone(),
             last={k: v.clone() for k, v in beh._last.items()}, ent=beh._im["ent"].clone(),
             actor_loss=aloss.clone(), value_loss=vloss.clone(), ema=beh.ema_vals.clone(),
             beh_grads={k: p.grad.clone() for k, p in beh.named_parameters()
                        if k.startswith("actor.") or k.startswith("value.")})
    im_flips = int(flipsA.item()) - wm_flips
    # ---- run B
    cfg, wm2, beh2 = Hh.build_models(name)
    flipsB = torch.zeros(1, dtype=torch.int32, device="cuda")
    wm_noise, im_noise = gpu_noise(name, exp, "beh", flipsB)
    post2, _, mets_wm = wm2._train(data, noise=wm_noise)
    post2 = {k: v.clone() for k, v in post2.items()}
    bres = beh2._train(post2, None, noise=im_noise)
    torch.cuda.synchronize()
    Bv = dict(wm=wm2, beh=beh2, mets_wm=mets_wm, mets_beh=bres[-1], flips=int(flipsB.item()),
              last={k: v.clone() for k, v in beh2._last.items()}, imag={k: v.clone() for k, v in bres[1].items()})
    d_wm, d_im = n_draws(s, onehot_actor)
    print(f"\n[{name}] teacher-forced sample flips: world model {wm_flips}/{d_wm}, imagination {im_flips}/{d_im}; "
          f"full update {Bv['flips']}/{d_wm + d_im}")
    return dict(name=name, s=s, cfg=cfg, exp=exp, g=g, A=A, B=Bv, wm_flips=wm_flips, im_flips=im_flips, data=data)


# ------------------------------------------------------------------------------------------------------
def test_sample_flip_rate(full):
    s = full["s"]
    d_wm, d_im = n_draws(s, s["actor_dist"] == "onehot")
    assert full["wm_flips"] <= max(1, MAX_FLIP_RATE * d_wm), (full["wm_flips"], d_wm)
    assert full["im_flips"] <= max(1, MAX_FLIP_RATE * d_im), (full["im_flips"], d_im)
    assert full["B"]["flips"] <= max(2, MAX_FLIP_RATE * (d_wm + d_im)), (full["B"]["flips"], d_wm + d_im)
    # the measured counts themselves (README / DESIGN quote them): world model + imagination of run A together
    lim_wm, lim_im = EXPECTED_FLIPS[full["name"]]
    assert full["wm_flips"] + full["im_flips"] <= max(lim_wm, lim_im), (full["name"], full["wm_flips"], full["im_flips"])


def test_world_model_outputs_vs_oracle_and_reference(full):
    s, A, g, ew = full["s"], full["A"], full["g"], full["exp"]["wm"]
    B, T = s["B"], s["T"]
    bt = lambda x: x.transpose(0, 1)  # time-major -> [B,T,...]
    # live oracle, every row
    close(bt(A["embed"]), ew["embed"], TOL_OUT, "embed")
    assert torch.equal(A["post"]["stoch"].cpu(), ew["post"]["stoch"].detach()), "teacher-forced posterior differs"
    close(A["post"]["logit"], ew["post"]["logit"], TOL_OUT, "post logit")
    close(A["post"]["deter"], ew["post"]["deter"], TOL_OUT, "deter")
    close(bt(A["out"]["prior_logit"]), ew["prior"]["logit"], TOL_OUT, "prior logit")
    close(bt(A["kl"]), ew["kl"], TOL_OUT, "kl value")
    close(bt(A["ent_post"]), ew["post_ent"], TOL_OUT, "posterior entropy")
    close(bt(-A["lp_r"].view(T, B)), ew["losses"]["reward"], TOL_OUT, "reward loss per step")
    close(bt(-A["lp_c"].view(T, B)), ew["losses"]["cont"], TOL_OUT, "cont loss per step")
    close(A["wm_loss"], ew["model_loss"], TOL_LOSS, "model_loss")
    # reference golden vectors: rows 0..1 of the batch, per-step losses of every row, scalars
    close(bt(A["embed"])[:2], g["embed"], TOL_OUT, "embed vs reference")
    for k in ("logit", "deter"):
        close(A["post"][k][:2], g["post/" + k], TOL_OUT, f"post {k} vs reference")
    assert np.array_equal(A["post"]["stoch"][:2].cpu().numpy(), g["post/stoch"]), "posterior draws vs reference"
    close(bt(A["out"]["prior_logit"])[:2], g["prior/logit"], TOL_OUT, "prior logit vs reference")
    close(bt(-A["lp_r"].view(T, B)), g["loss/reward"], TOL_OUT, "reward loss vs reference")
    close(bt(-A["lp_c"].view(T, B)), g["loss/cont"], TOL_OUT, "cont loss vs reference")
    close(bt(A["kl"]), g["kl_value"], TOL_OUT, "kl vs reference")
    close(A["wm_loss"], g["model_loss"], TOL_LOSS, "model_loss vs reference")
    if s["encoder"] == "cnn":
        close(A["recon"].view(T, B, 64, 64, 3).transpose(0, 1), ew["recon"], TOL_OUT, "decoded pixels")
        close(bt(A["loss_img"].view(T, B)), ew["losses"]["image"], TOL_OUT, "image loss per frame")
        # decoded pixels of frames (b=0, t=0..1): GPU rows t*B + 0
        rec = torch.stack([A["recon"][0], A["recon"][B]], 0)[None]
        close(rec, g["recon"], TOL_OUT, "decoded pixels vs reference")
        close(bt(A["loss_img"].view(T, B)), g["loss/image"], TOL_OUT, "image loss vs reference")
    else:
        for k, _ in common.PROPRIO_KEYS:
            close(bt(A["loss_vec"][k].view(T, B)), ew["losses"][k], TOL_OUT, f"{k} loss per step")
            close(bt(A["loss_vec"][k].view(T, B)), g["loss/" + k], TOL_OUT, f"{k} loss vs reference")


def test_world_model_gradients_vs_oracle_and_reference(full):
    A, g, exp = full["A"], full["g"], full["exp"]
    worst = 0.0
    for k, ref in exp["wm_grads"].items():
        worst = max(worst, close(A["wm_grads"][k], ref, TOL_GRAD, "grad " + k, floor=1e-12))
        checksum_close(A["wm_grads"][k], g["sum/grad/" + k], TOL_GRAD, "grad checksum vs reference " + k)
    gn = torch.sqrt(sum((v.double() ** 2).sum() for v in A["wm_grads"].values()))
    # tight against the float64 norms (of the oracle's gradient tensors; the fixture's is accumulated in float64 too);
    # torch's own float32 clip_grad_norm_ scalar drifts by ~1e-3 over 100-180 M parameters on the CPU
    true_norm = torch.sqrt(sum((v.double() ** 2).sum() for v in exp["wm_grads"].values()))
    close(gn, true_norm, 2e-5, "model grad norm (float64)")
    close(gn, g["model_grad_norm"], 2e-4, "model grad norm vs reference")
    close(gn, exp["model_grad_norm"], 2e-3 if full["s"]["deter"] >= 2048 else 2e-4, "model grad norm (oracle's float32 scalar)")
    print(f"\n[{full['name']}] worst world-model gradient error (relative to the tensor's max): {worst:.2e}")


def test_imagination_and_returns_vs_oracle_and_reference(full):
    s, A, g, eb = full["s"], full["A"], full["g"], full["exp"]["beh0"]
    B, T, H = s["B"], s["T"], s["H"]
    un = lambda x: Hh.from_time_major_rows(x, B, T)  # GPU rows t*B+b -> the reference's b*T+t
    stoch = un(A["imag"]["stoch"])
    assert torch.equal(stoch.cpu(), eb["states"]["stoch"]), "teacher-forced imagined states differ"
    close(un(A["imag"]["deter"]), eb["states"]["deter"], TOL_OUT, "imag deter")
    close(un(A["imag"]["logit"])[1:], eb["states"]["logit"][1:], TOL_OUT, "imag logit")
    close(un(A["action"]), eb["actions"], TOL_OUT, "imag action")
    close(un(A["last"]["reward"]), eb["reward"].squeeze(-1), TOL_OUT, "imag reward")
    close(un(A["last"]["value"]), eb["value"].squeeze(-1), TOL_OUT, "imag value")
    close(un(A["last"]["target"]), eb["target"].squeeze(-1), TOL_OUT, "lambda-return")
    close(un(A["weights"]), eb["weights"], TOL_OUT, "weights")
    close(un(A["ent"]), eb["actor_ent"], TOL_OUT, "actor entropy")
    close(A["actor_loss"], eb["actor_loss"], TOL_LOSS, "actor_loss")
    close(A["value_loss"], eb["value_loss"], TOL_LOSS, "value_loss")
    close(A["ema"], full["exp"]["ema0"], TOL_OUT, "ema_vals")
    # the reference's rows 0..7 (b = 0, t = 0..7) of every imagination tensor
    r8 = lambda x: un(x)[:, :8]
    assert np.array_equal(r8(A["imag"]["stoch"]).cpu().numpy(), g["imag/stoch"]), "imagined draws vs reference"
    close(r8(A["imag"]["deter"]), g["imag/deter"], TOL_OUT, "imag deter vs reference")
    close(r8(A["imag"]["logit"])[1:], g["imag/logit"][1:], TOL_OUT, "imag logit vs reference")
    feat = torch.cat([A["imag"]["stoch"].flatten(2), A["imag"]["deter"]], -1)
    close(r8(feat), g["imag/feat"], TOL_OUT, "imag feat vs reference")
    close(r8(A["action"]), g["imag/action"], TOL_OUT, "imag action vs reference")
    close(r8(A["last"]["reward"])[..., None], g["imag/reward"], TOL_OUT, "imag reward vs reference")
    close(r8(A["last"]["value"])[..., None], g["imag/value"], TOL_OUT, "imag value vs reference")
    close(r8(A["last"]["target"])[..., None], g["imag/target"], TOL_OUT, "lambda-return vs reference")
    close(r8(A["weights"]), g["imag/weights"], TOL_OUT, "weights vs reference")
    close(r8(A["ent"]), g["imag/actor_ent"], TOL_OUT, "actor entropy vs reference")
    close(A["actor_loss"], g["actor_loss"], TOL_LOSS_REF, "actor_loss vs reference")
    close(A["value_loss"], g["value_loss"], TOL_LOSS_REF, "value_loss vs reference")
    # whole-tensor pins of the reference (sum, abs-sum, max over all 15 x N rows)
    checksum_close(un(A["last"]["target"]), g["sum/imag/target"], 2e-4, "lambda-return checksum vs reference")
    checksum_close(un(A["last"]["reward"]), g["sum/imag/reward"], 2e-4, "reward checksum vs reference")
    checksum_close(un(A["imag"]["deter"]), g["sum/imag/deter"], 2e-4, "imag deter checksum vs reference")


def test_behaviour_gradients_vs_oracle_and_reference(full):
    A, g, exp = full["A"], full["g"], full["exp"]
    worst = 0.0
    for nm, ref_grads in (("actor", exp["actor_grads0"]), ("value", exp["value_grads0"])):
        for k, ref in ref_grads.items():
            worst = max(worst, close(A["beh_grads"][k], ref, TOL_GRAD, "grad " + k, floor=1e-12))
            checksum_close(A["beh_grads"][k], g["sum/grad/" + k], TOL_GRAD, "grad checksum vs reference " + k)
        gn = torch.sqrt(sum((v.double() ** 2).sum() for k, v in A["beh_grads"].items() if k.startswith(nm + ".")))
        close(gn, g[nm + "_grad_norm"], 3e-4, nm + " grad norm vs reference")
    print(f"\n[{full['name']}] worst actor/critic gradient error (relative to the tensor's max): {worst:.2e}")


def test_full_update_vs_oracle_and_reference(full):
    """Run B: WorldModel._train then ImagBehavior._train on the UPDATED world model (dreamer.py:194-200)."""
    s, Bv, g, exp = full["s"], full["B"], full["g"], full["exp"]
    B, T = s["B"], s["T"]
    un = lambda x: Hh.from_time_major_rows(x, B, T)
    mw, mb, eb = Bv["mets_wm"], Bv["mets_beh"], exp["beh"]
    f = lambda d, k: torch.tensor(float(d[k]))
    # This run's rollout uses the world model AFTER the GPU's own Adam step.  At crafter widths (100-180 M parameters)
    # a few hundred entries whose gradient is ~0 land up to 2 lr away from the oracle's (sign-like first Adam step on
    # a rounding-level gradient): test_rollout_on_the_oracles_updated_weights counts them and shows that on IDENTICAL
    # updated weights the rollout is within 1e-4; here the bound is the measured effect of those entries.
    wide = s["deter"] >= 2048
    TOL_UPD, TOL_UPD_LOSS = (1e-3, 2e-4) if wide else (TOL_OUT, TOL_LOSS)
    close(f(mw, "model_loss"), exp["wm"]["model_loss"], TOL_LOSS, "model_loss")
    close(f(mw, "model_grad_norm"), exp["model_grad_norm"], 2e-3 if wide else 2e-4, "model_grad_norm")
    close(f(mb, "actor_loss"), eb["actor_loss"], TOL_UPD_LOSS, "actor_loss")
    close(f(mb, "value_loss"), eb["value_loss"], TOL_UPD_LOSS, "value_loss")
    close(f(mb, "actor_grad_norm"), exp["actor_grad_norm"], 10 * TOL_UPD if wide else 3e-4, "actor_grad_norm")
    close(f(mb, "value_grad_norm"), exp["value_grad_norm"], 10 * TOL_UPD if wide else 3e-4, "value_grad_norm")
    assert torch.equal(un(Bv["imag"]["stoch"]).cpu(), eb["states"]["stoch"].detach())
    close(un(Bv["last"]["target"]), eb["target"].detach().squeeze(-1), TOL_UPD, "lambda-return (updated model)")
    close(un(Bv["last"]["reward"]), eb["reward"].detach().squeeze(-1), TOL_UPD, "reward (updated model)")
    close(un(Bv["last"]["value"]), eb["value"].detach().squeeze(-1), TOL_UPD, "value (updated model)")
    close(Bv["beh"].ema_vals, exp["ema"], TOL_UPD, "ema_vals")
    close(Bv["beh"].ema_vals, g["ema_vals_after"], TOL_UPD, "ema_vals vs reference")  # (aliases the buffer after _train)
    # the reference's own _train metrics for this minibatch
    for k in ("model_loss", "kl", "prior_ent", "post_ent"):
        close(f(mw, k), g["train/" + k], TOL_LOSS if k == "model_loss" else TOL_OUT, k + " vs reference")
    close(f(mw, "model_grad_norm"), g["train/model_grad_norm"], 2e-3 if wide else 2e-4, "model_grad_norm vs reference")
    for k, tol in (("actor_loss", max(TOL_LOSS_REF, TOL_UPD_LOSS)), ("value_loss", max(TOL_LOSS_REF, TOL_UPD_LOSS)),
                   ("actor_grad_norm", 10 * TOL_UPD if wide else 3e-4), ("value_grad_norm", 10 * TOL_UPD if wide else 3e-4),
                   ("actor_entropy", TOL_UPD), ("EMA_005", TOL_UPD), ("EMA_095", TOL_UPD),
                   ("target_mean", TOL_UPD), ("target_std", TOL_UPD), ("imag_reward_mean", TOL_UPD),
                   ("value_mean", TOL_UPD)):
        close(f(mb, k), g["train/" + k], tol, k + " vs reference")
    # Adam-updated parameters: oracle tensors and the reference's checksums
    sd = dict(Bv["wm"].state_dict())
    sd.update({k: v for k, v in Bv["beh"].state_dict().items() if not k.startswith("_world_model.")})
    lr = {"actor": 3e-5, "value": 3e-5}
    for k, ref in exp["params_after"].items():
        if k == "ema_vals":
            continue
        if k.startswith("_slow_value."):  # EMA of the critic (models.py:683-689), no optimizer step
            close(sd[k], ref, 1e-6, "after " + k)
            checksum_close(sd[k], g["sum/after/" + k], 1e-6, "after checksum vs reference " + k)
            continue
        this_lr = lr.get(k.split(".")[0], 1e-4)
        adam_close(sd[k], ref, this_lr, "after " + k)
        cs, ref_cs = common.checksum(sd[k].detach().cpu().numpy()), g["sum/after/" + k]
        n = sd[k].numel()
        # abs-sum moves by at most (outliers <= 5e-3 n) * 2 lr + n * 2e-6 rounding
        assert abs(cs[1] - ref_cs[1]) <= n * (2e-6 + 5e-3 * 2.1 * this_lr) + 1e-9, f"after checksum {k}"
        assert abs(cs[2] - ref_cs[2]) <= 2.1 * this_lr + 2e-6, f"after max {k}"


def test_rollout_on_the_oracles_updated_weights(full):
    """What separates the GPU's updated world model from the oracle's, measured: (1) the number of world-model
    parameters that differ by more than 2e-6 after the first Adam step and the largest difference (bounded by 2 lr:
    a sign-like step on a rounding-level gradient); (2) with the ORACLE's updated world model and slow critic loaded
    into the GPU modules, the imagination rollout, rewards, values, lambda-returns, both losses and every actor /
    critic gradient are within the 1e-4 bar at every width -- the rollout kernels carry no error of their own."""
    name, s, exp, Bv = full["name"], full["s"], full["exp"], full["B"]
    B, T = s["B"], s["T"]
    mid = exp["params_mid"]
    sd = Bv["wm"].state_dict()
    n_diff = n_all = 0
    worst = 0.0
    for k, v in sd.items():
        d = (v.detach().cpu().double() - mid[k].double()).abs()
        n_diff += int((d > 2e-6).sum())
        n_all += d.numel()
        worst = max(worst, float(d.max()))
    print(f"\n[{name}] world-model parameters differing by > 2e-6 after the first Adam step: {n_diff} of {n_all} "
          f"({n_diff / n_all:.2e}); largest difference {worst:.2e} (lr 1e-4)")
    assert worst <= 2.1e-4 and n_diff <= 5e-3 * n_all
    cfg, wm, beh = Hh.build_models(name)
    wsd = wm.state_dict()
    for k in wsd:
        wsd[k] = mid[k]
    wm.load_state_dict(wsd)
    bsd = beh.state_dict()
    for k in bsd:
        if k.startswith("_slow_value."):
            bsd[k] = mid[k]
        elif k.startswith("_world_model."):
            bsd[k] = mid[k[len("_world_model."):]]
    beh.load_state_dict(bsd)
    flips = torch.zeros(1, dtype=torch.int32, device="cuda")
    _, im_noise = gpu_noise(name, exp, "beh", flips)
    beh._update_slow_target = lambda: None  # already applied in params_mid
    # the oracle's posterior as a [B,T,...] view of time-major storage, which is how WorldModel._train hands it over
    # (rows of the rollout are then t*B+b, the order gpu_noise lays the draws out in)
    start = {k: v.detach().cuda().transpose(0, 1).contiguous().transpose(0, 1) for k, v in exp["wm"]["post"].items()}
    beh.train_fwd_bwd(start, noise=im_noise)
    (_, imag_state, action, weights), _, (aloss, vloss) = beh._pending
    un = lambda x: Hh.from_time_major_rows(x, B, T)
    eb = exp["beh"]
    assert int(flips.item()) <= 1, f"{int(flips.item())} teacher-forced flips"
    assert torch.equal(un(imag_state["stoch"]).cpu(), eb["states"]["stoch"].detach()), "imagined states differ"
    close(un(imag_state["deter"]), eb["states"]["deter"], TOL_OUT, "imag deter (oracle-updated weights)")
    close(un(action), eb["actions"], TOL_OUT, "imag action (oracle-updated weights)")
    close(un(beh._last["reward"]), eb["reward"].squeeze(-1), TOL_OUT, "reward (oracle-updated weights)")
    close(un(beh._last["value"]), eb["value"].squeeze(-1), TOL_OUT, "value (oracle-updated weights)")
    close(un(beh._last["target"]), eb["target"].squeeze(-1), TOL_OUT, "lambda-return (oracle-updated weights)")
    close(aloss, eb["actor_loss"], TOL_LOSS, "actor_loss (oracle-updated weights)")
    close(vloss, eb["value_loss"], TOL_LOSS, "value_loss (oracle-updated weights)")
    params = dict(beh.named_parameters())
    for k, ref in list(exp["actor_grads"].items()) + list(exp["value_grads"].items()):
        close(params[k].grad, ref, TOL_GRAD, "grad (oracle-updated weights) " + k, floor=1e-12)


# ------------------------------------------------------------------------------------------------------
def first_flip(got, ref):
    """got/ref one-hot [B,T,S,D] -> per-row index of the first differing step (T if none)."""
    diff = (got != ref).flatten(2).any(-1)  # [B,T]
    T = diff.shape[1]
    return torch.where(diff.any(1), diff.float().argmax(1), torch.full((diff.shape[0],), T))


def test_free_running_matches_until_first_flip(full):
    """No teacher forcing: the posterior of every replay row equals the oracle's up to the row's first sample
    flip (there should be none); the number of flipped rows is printed and bounded."""
    name, s, ew = full["name"], full["s"], full["exp"]["wm"]
    T = s["T"]
    cfg, wm, beh = Hh.build_models(name)
    n = {k: torch.from_numpy(v).cuda() for k, v in common.make_noise(name).items()}
    post, _, mets = wm._train(full["data"], noise=dict(q_prior=n["q_prior"], q_post=n["q_post"]))
    ff = first_flip(post["stoch"].cpu(), ew["post"]["stoch"].detach())
    flipped = int((ff < T).sum())
    print(f"\n[{name}] free-running: {flipped}/{s['B']} replay rows with a sample flip (first flips at "
          f"{[int(x) for x in ff[ff < T]]})")
    for b in range(s["B"]):
        t_ok = int(ff[b])
        close(post["logit"][b, :t_ok + 1], ew["post"]["logit"][b, :t_ok + 1], TOL_OUT, f"free-running post logit row {b}")
        close(post["deter"][b, :t_ok + 1], ew["post"]["deter"][b, :t_ok + 1], TOL_OUT, f"free-running deter row {b}")
    # a flip needs two ratios within an ulp: with 2*T*S draws per row the expected count is << 1 row
    assert flipped <= max(1, 0.01 * s["B"]), f"{flipped} rows flipped"
    if flipped == 0:
        close(torch.tensor(float(mets["model_loss"])), ew["model_loss"], TOL_LOSS, "free-running model_loss")


def test_batch_rows_are_independent(full):
    """Permuting the replay rows (and their noise) permutes the posterior: no cross-row coupling in the
    scan, the reset blend or the few-row GEMM tiling."""
    name, s = full["name"], full["s"]
    B = s["B"]
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    data = {k: v[perm.numpy()] for k, v in full["data"].items()}
    n = {k: torch.from_numpy(v).cuda() for k, v in common.make_noise(name).items()}
    pc = perm.cuda()
    cfg, wm, beh = Hh.build_models(name)
    ref_post, _, _ = wm._train(full["data"], noise=dict(q_prior=n["q_prior"], q_post=n["q_post"]))
    ref = {k: v[pc].clone() for k, v in ref_post.items()}
    cfg, wm, beh = Hh.build_models(name)
    noise = dict(q_prior=n["q_prior"][:, pc].contiguous(), q_post=n["q_post"][:, pc].contiguous())
    post, _, _ = wm._train(data, noise=noise)
    assert torch.equal(post["stoch"], ref["stoch"]), "permuted rows sampled differently"
    close(post["logit"], ref["logit"], 1e-6, "permuted logit")
    close(post["deter"], ref["deter"], 1e-6, "permuted deter")


@pytest.mark.parametrize("name", CONFIGS)
def test_graph_replay_matches_eager_and_trains(name):
    """hipGraph replay computes what the eager launch sequence computes (same Philox stream), and a few updates on
    one minibatch reduce the model loss."""
    import tools
    from dv3hip.graph import UpdateRunner

    outs = []
    for use_graph in (False, True):
        cfg, wm, beh = Hh.build_models(name)
        tools.default_rng("cuda:0", seed=7)
        data = {k: torch.from_numpy(v).cuda() for k, v in common.make_batch(name).items()}
        r = UpdateRunner(wm, beh, use_graph=use_graph, warm=1)
        losses = []
        for _ in range(4):
            r.step(data)
            losses.append(float(r.last_metrics["model_loss"]))
        torch.cuda.synchronize()
        assert r.use_graph == use_graph, "hipGraph capture was refused: the runner fell back to eager launches"
        assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
        for k in ("actor_loss", "value_loss", "actor_grad_norm", "value_grad_norm", "model_grad_norm"):
            assert np.isfinite(float(r.last_metrics[k])), k
        outs.append((losses[-1], float(r.last_metrics["actor_loss"]), float(r.last_metrics["value_loss"]),
                     wm.dynamics.W.detach().clone()))
    (ml0, al0, vl0, w0), (ml1, al1, vl1, w1) = outs
    assert abs(ml0 - ml1) <= 2e-3 * abs(ml0), (ml0, ml1)
    assert abs(vl0 - vl1) <= 2e-2 * max(1.0, abs(vl0)), (vl0, vl1)
    close(w1, w0, 1e-4, "learned initial state after 4 updates")


def _wm_gradients(name, mode, seed=11):
    """Gradient bucket of one world-model forward/backward at `name`, launched `mode`: "inline" (no lanes), "lanes"
    (eager, the reverse scan and the deferred weight gradients on the two CU-masked streams) or "segments" (the same cut
    into one hipGraph per lane and replayed twice -- the second replay must overwrite, not accumulate)."""
    import tools
    from dv3hip import engine as E
    from dv3hip.graph import SegmentRecorder

    cfg, wm, beh = Hh.build_models(name)
    data = {k: torch.from_numpy(v).cuda() for k, v in common.make_batch(name).items()}
    s = common.SHAPES[name]
    g = torch.Generator(device="cpu").manual_seed(seed)
    shape = (s["T"], s["B"], s["stoch"], s["discrete"])
    noise = dict(q_prior=torch.empty(shape).exponential_(1.0, generator=g).clamp_min(1e-20).cuda(),
                 q_post=torch.empty(shape).exponential_(1.0, generator=g).clamp_min(1e-20).cuda())
    tools.default_rng("cuda:0", seed=seed)
    saved = E.SideStream.lanes
    E.SideStream.lanes = mode != "inline"
    try:
        run = lambda: wm.train_fwd_bwd(data, noise=noise)
        used = []
        real_get = E.Lanes.get.__func__

        def spy(cls, device):
            used.append(device)
            return real_get(cls, device)

        st = E.Lanes.get("cuda:0").whole_chip_stream()  # (the lanes are taken only beside their own whole-chip stream)
        torch.cuda.synchronize()
        E.Lanes.get = classmethod(spy)
        try:
            with torch.cuda.stream(st):
                run()  # (also the warm call of the captured variant: workspaces exist before capture)
        finally:
            E.Lanes.get = classmethod(real_get)
        assert bool(used) == (mode != "inline"), (mode, used)
        if mode == "segments":
            with torch.cuda.stream(st):
                torch.cuda.synchronize()
                rec = SegmentRecorder(torch.cuda.graph_pool_handle(), torch.device("cuda:0")).record(run)
                lanes = [lane for lane, _ in rec.segments]
                assert lanes == ["main", "sync", "main", "side", "scan", "sync_lane", "scan", "main"], lanes
                rec.replay()
                rec.replay()
        torch.cuda.synchronize()
    finally:
        E.SideStream.lanes = saved
    post = wm._pending[0]
    return wm._model_opt.bucket.grad.clone(), post["stoch"].clone(), float(wm._pending[3])


def test_lanes_are_taken_where_they_were_measured_to_pay():
    """engine.RSSMEngine.lanes_pay: the image configs with a latency-bound scan (cfg 2, cfg 3) -- not the vector decoder of
    cfg 1 (too little deferred work), not the wide cells of cfg 4 / cfg 5 (the scan is bandwidth-bound there)."""
    from types import SimpleNamespace

    from dv3hip import engine as E

    def pays(name, heavy):
        s = common.SHAPES[name]
        eng = SimpleNamespace(De=s["deter"], Hd=s["hidden"], B=s["B"])
        return E.RSSMEngine.lanes_pay(eng, heavy)

    assert pays("cfg2", True) and pays("cfg3", True)
    assert not pays("cfg1", False) and not pays("cfg4", True) and not pays("cfg5", True)


@pytest.mark.parametrize("name", ["cfg2", "cfg3"])
def test_cu_lanes_compute_what_the_inline_sequence_computes(name):
    """The reverse observe scan beside the deferred weight gradients on two CU-masked streams (engine.Lanes), eagerly and
    as one hipGraph per lane (graph.SegmentRecorder), against the same launches in line on one stream: same samples,
    same loss, gradients equal up to the summation order of the scan's atomic accumulations (which differs between two
    inline runs as well)."""
    g0, s0, l0 = _wm_gradients(name, "inline")
    scale = float(g0.abs().max())
    for mode in ("lanes", "segments"):
        g1, s1, l1 = _wm_gradients(name, mode)
        assert torch.equal(s0, s1), f"{mode}: sampled states differ"
        assert abs(l0 - l1) <= 1e-6 * abs(l0), (mode, l0, l1)
        err = float((g0 - g1).abs().max())
        assert err <= 2e-6 * scale, f"{mode}: gradient differs by {err:.3e} (max |g| {scale:.3e})"


def test_segment_recorder_leaves_no_capture_behind_when_the_recorded_code_raises():
    """An error inside a lane segment ends that capture, restores SideStream.recorder and reaches the caller unchanged
    (only a refusal of the capture itself becomes CaptureRefused)."""
    from dv3hip import engine as E
    from dv3hip.graph import SegmentRecorder

    x = torch.zeros(1024, device="cuda")

    def body():
        x.add_(1.0)
        side = E.SideStream("cuda:0")
        side.run([lambda: x.add_(1.0)])
        with side.chain():
            x.add_(1.0)
            raise ValueError("shape bug in the recorded code")

    st = E.Lanes.get("cuda:0").whole_chip_stream()
    with torch.cuda.stream(st):
        rec = SegmentRecorder(torch.cuda.graph_pool_handle(), torch.device("cuda:0"))
        with pytest.raises(ValueError, match="shape bug"):
            rec.record(body)
        assert E.SideStream.recorder is None
        assert not torch.cuda.is_current_stream_capturing()
        for s in rec.lanes.streams.values():
            with torch.cuda.stream(s):
                assert not torch.cuda.is_current_stream_capturing()
        # ... and the streams are usable afterwards: the same body without the error records and replays
        def good():
            x.add_(1.0)
            side = E.SideStream("cuda:0")
            side.run([lambda: x.add_(2.0)])
            with side.chain():
                x.add_(4.0)
            side.join()
            x.mul_(2.0)

        x.zero_()
        torch.cuda.synchronize()
        rec2 = SegmentRecorder(torch.cuda.graph_pool_handle(), torch.device("cuda:0")).record(good)
        assert [lane for lane, _ in rec2.segments] == ["main", "side", "scan", "main"]
        rec2.replay()
        torch.cuda.synchronize()
    assert float(x[0]) == 14.0 and float(x.sum()) == 14.0 * 1024


def test_cu_masked_stream_restricts_a_launch_to_its_compute_units():
    """dv3_stream_create_cu_masked: a chip-filling product on a stream that owns half of the CUs takes about twice as
    long as on the whole chip, launched eagerly and from a hipGraph replayed on that stream (the mask belongs to the
    queue the graph is launched on)."""
    from dv3hip import engine as E
    from dv3hip import ops

    ln = E.Lanes.get("cuda:0")
    assert ln.cus["scan"] + ln.cus["side"] == ln.cus["whole"] == torch.cuda.get_device_properties(0).multi_processor_count
    A = torch.randn(4096, 2048, device="cuda")
    B = torch.randn(4096, 2048, device="cuda")
    C = torch.empty(4096, 4096, device="cuda")

    def work():
        for _ in range(4):
            ops.gemm(A, B, C, transB=True)

    def timed(fn, stream):
        best = 1e9
        with torch.cuda.stream(stream):
            for _ in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
        return best

    whole = timed(work, ln.streams["whole"])
    half = timed(work, ln.streams["side"])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=ln.streams["side"]):
        work()
    half_graph = timed(g.replay, ln.streams["side"])
    whole_graph = timed(g.replay, ln.streams["whole"])
    ratio = ln.cus["whole"] / ln.cus["side"]
    for label, t, ref in (("eager", half, whole), ("graph", half_graph, whole_graph)):
        assert 0.75 * ratio <= t / ref <= 1.35 * ratio, f"{label}: {t:.3f} ms on {ln.cus['side']} CUs vs {ref:.3f} ms on all"
    ref = A @ B.T
    assert float((C - ref).abs().max()) <= 1e-3 * float(ref.abs().max())


# ------------------------------------------------------------------------------------------------------
# BASELINE cfg 4 (dmc_vision, crafter-size model: deter 4096, hidden / units 1024, cnn_depth 96, batch 64 x 64) and
# cfg 5 (crafter: deter 2048, five-layer heads, 17-way one-hot actor, reinforce, sequence length 256; the per-GPU shard
# of the 128-sequence batch at DP = 8).  No reference fixture exists at these sizes (the reference needs minutes per
# update on the CPU); their kernel shape classes are pinned against the oracle at reduced batch (tests/test_path_gpu.py,
# cfg4_b4 / cfg5_b4) and per kernel (test_kernels_gpu.py: GRU rows of 6144 / 12288, LayerNorm 1024, conv depth 96).
# Here: size-independent properties at full size.
@pytest.mark.parametrize("name", ["cfg4", "cfg5"])
def test_large_configs_train_and_rows_are_independent(name):
    import models
    import tools
    from dv3hip import shapes
    from dv3hip.graph import UpdateRunner

    s = common.SHAPES[name]
    dev = "cuda:0"
    cfg = shapes.make_config(name, dev)
    torch.manual_seed(0)
    wm = models.WorldModel(shapes.obs_space(name), None, 0, cfg).to(dev)
    beh = models.ImagBehavior(cfg, wm).to(dev)
    wm.requires_grad_(False), beh.requires_grad_(False)
    tools.default_rng(dev, seed=5)
    host = shapes.synthetic_batch(name, seed=0)
    data = {k: torch.from_numpy(v).to(dev) for k, v in host.items()}
    # (1) batch-row permutation equivariance of the posterior (explicit noise so both runs draw the same)
    B, T, S, D = s["B"], s["T"], s["stoch"], s["discrete"]
    g = torch.Generator(device="cpu").manual_seed(1)
    q1 = torch.empty(T, B, S, D).exponential_(1.0, generator=g).clamp_min(1e-20).to(dev)
    q2 = torch.empty(T, B, S, D).exponential_(1.0, generator=g).clamp_min(1e-20).to(dev)
    wm.train_fwd_bwd(data, noise=dict(q_prior=q1, q_post=q2))
    ref = {k: v.clone() for k, v in wm._pending[0].items()}
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(2)).to(dev)
    wm.train_fwd_bwd({k: v[perm] for k, v in data.items()},
                     noise=dict(q_prior=q1[:, perm].contiguous(), q_post=q2[:, perm].contiguous()))
    post = wm._pending[0]
    assert torch.equal(post["stoch"], ref["stoch"][perm]), "permuted rows sampled differently"
    close(post["deter"], ref["deter"][perm], 1e-5, "permuted deter")
    close(post["logit"], ref["logit"][perm], 1e-5, "permuted logit")
    # (2) a few full updates on one minibatch, launched eagerly AND replayed from hipGraphs (the path the 384 / 256 ms
    # figures of DESIGN.md section 5 come from): finite everywhere, model loss decreasing, and the replayed sequence
    # computes what the eager one does (same Philox stream; as test_graph_replay_matches_eager_and_trains for cfg 1-3)
    del ref, post
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        wm = models.WorldModel(shapes.obs_space(name), None, 0, cfg).to(dev)
        beh = models.ImagBehavior(cfg, wm).to(dev)
        wm.requires_grad_(False), beh.requires_grad_(False)
        tools.default_rng(dev, seed=5)
        r = UpdateRunner(wm, beh, use_graph=use_graph, warm=1)
        losses = []
        for _ in range(4):
            r.step(data)
            losses.append(float(r.last_metrics["model_loss"]))
        assert r.use_graph == use_graph, "hipGraph capture was refused"
        assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
        for k in ("actor_loss", "value_loss", "actor_grad_norm", "value_grad_norm", "model_grad_norm", "kl", "actor_entropy"):
            assert np.isfinite(float(r.last_metrics[k])), k
        outs.append((losses, float(r.last_metrics["actor_loss"]), float(r.last_metrics["value_loss"]),
                     wm.dynamics.W.detach().clone()))
        ws_bytes = wm.dynamics.engine.ws.nbytes()
        print(f"\n[{name}] {'replayed' if use_graph else 'eager'}: losses {['%.2f' % x for x in losses]}; RSSM workspace "
              f"{ws_bytes / 2**30:.1f} GiB; peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
        del r, wm, beh
        import gc

        gc.collect()
        torch.cuda.empty_cache()
    (l0, al0, vl0, w0), (l1, al1, vl1, w1) = outs
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-3 * abs(a), (l0, l1)
    assert abs(vl0 - vl1) <= 2e-2 * max(1.0, abs(vl0)), (vl0, vl1)
    close(w1, w0, 1e-4, "learned initial state after 4 updates, replayed vs eager")
