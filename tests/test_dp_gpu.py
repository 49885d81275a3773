"""Data parallel on the GPU: two ranks (gloo; both on cuda:0 -- a one-GPU box, same code path as
RCCL) train on half of a global batch each; the world-model update they reach must equal the single-process
update on the whole batch: the loss is a mean over rows, so the SUM all-reduce of the flat gradient bucket with the
1/world scale folded into clip+Adam is the gradient of the global batch, and clipping sees the global norm.
(The behaviour update is only checked for replica consistency: its return normalisation uses per-rank quantiles,
as the reference's would.)"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.golden import common

pytestmark = pytest.mark.gpu
NAME = "tiny"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_inputs(rank):
    """Batch and observe noise of one rank: the tiny batch / noise with seed = rank."""
    return common.make_batch(NAME, seed=rank), common.make_noise(NAME, seed=rank)


def _worker(rank, world, port, outdir):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, wm, beh = Hh.build_models(NAME)  # identical replica
        batch, noise = _rank_inputs(rank)
        nz = {k: torch.from_numpy(v).cuda() for k, v in noise.items()}
        post, _, mets = wm._train(batch, noise=dict(q_prior=nz["q_prior"], q_post=nz["q_post"]))
        wm_params = torch.cat([p.detach().reshape(-1) for p in wm.parameters()]).cpu()
        wm_grad = torch.cat([p.grad.detach().reshape(-1) for p in wm.parameters()]).cpu() / world  # SUM over ranks
        beh._train(post, None)
        beh_params = torch.cat([p.detach().reshape(-1) for p in list(beh.actor.parameters()) + list(beh.value.parameters())]).cpu()
        torch.save({"wm": wm_params, "wm_grad": wm_grad, "beh": beh_params, "grad_norm": float(mets["model_grad_norm"]),
                    "loss": float(mets["model_loss"])}, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_global_batch():
    import torch.multiprocessing as mp

    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, _free_port(), outdir), nprocs=world, join=True)
        r0 = torch.load(os.path.join(outdir, "rank0.pt"), weights_only=True)
        r1 = torch.load(os.path.join(outdir, "rank1.pt"), weights_only=True)
    # replicas stay in sync (same all-reduced gradient, same optimizer state)
    assert torch.allclose(r0["wm"], r1["wm"], rtol=0, atol=1e-7)
    assert torch.allclose(r0["beh"], r1["beh"], rtol=0, atol=1e-7)
    assert r0["grad_norm"] == pytest.approx(r1["grad_norm"], rel=1e-6)  # the norm of the GLOBAL gradient on both
    # single process, global batch = the two rank batches side by side (batch axis 0; noise [T, B, ...] axis 1)
    (b0, n0), (b1, n1) = _rank_inputs(0), _rank_inputs(1)
    batch = {k: np.concatenate([b0[k], b1[k]], 0) for k in b0}
    noise = {k: torch.from_numpy(np.concatenate([n0[k], n1[k]], 1)).cuda() for k in ("q_prior", "q_post")}
    _, wm, _ = Hh.build_models(NAME)
    _, _, mets = wm._train(batch, noise=noise)
    single = torch.cat([p.detach().reshape(-1) for p in wm.parameters()]).cpu()
    single_grad = torch.cat([p.grad.detach().reshape(-1) for p in wm.parameters()]).cpu()
    assert float(mets["model_loss"]) == pytest.approx(0.5 * (r0["loss"] + r1["loss"]), rel=1e-5)
    assert float(mets["model_grad_norm"]) == pytest.approx(r0["grad_norm"], rel=1e-4)
    # the all-reduced gradient IS the gradient of the global batch (fp32 rounding: different batch shapes take
    # different tiles / summation orders)
    gscale = float(single_grad.abs().max())
    assert float((r0["wm_grad"] - single_grad).abs().max()) <= 2e-5 * gscale
    # ... and so are the parameters, except where the first Adam step (lr * g / (|g| + eps)) turns a rounding-level
    # difference of a near-zero gradient into a fraction of lr: those must be rare and bounded by 2 * lr
    diff = (single - r0["wm"]).abs()
    assert float((diff > 2e-6).float().mean()) < 2e-3
    assert float(diff.max()) <= 2.1 * wm._config.model_lr


def _rccl_worker(rank, world, port, outdir):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        _, wm, beh = Hh.build_models(NAME)
        batch, noise = _rank_inputs(0)
        nz = {k: torch.from_numpy(v).cuda() for k, v in noise.items()}
        post, _, mets = wm._train(batch, noise=dict(q_prior=nz["q_prior"], q_post=nz["q_post"]))
        beh._train(post, None)
        torch.save({"wm": torch.cat([p.detach().reshape(-1) for p in wm.parameters()]).cpu(),
                    "backend": dist.get_backend(), "ema": beh.ema_vals.cpu()}, os.path.join(outdir, "rccl.pt"))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_runs_the_update_on_one_rank():
    """backend "nccl" (= RCCL on ROCm) with the only GPU of this box: the process group initialises, the flat-bucket
    all-reduces and the EMA all-reduce go through RCCL (world size 1: the collective is the identity), and the update
    equals the no-process-group update bit for bit.  (Two ranks cannot share one device under RCCL; the 2-rank logic is
    covered with gloo above and on CPU in tests/test_dp_cpu.py.)"""
    import torch.multiprocessing as mp

    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_rccl_worker, args=(1, _free_port(), outdir), nprocs=1, join=True)
        r = torch.load(os.path.join(outdir, "rccl.pt"), weights_only=True)
    assert r["backend"] == "nccl"
    _, wm, beh = Hh.build_models(NAME)
    batch, noise = _rank_inputs(0)
    nz = {k: torch.from_numpy(v).cuda() for k, v in noise.items()}
    wm._train(batch, noise=dict(q_prior=nz["q_prior"], q_post=nz["q_post"]))
    single = torch.cat([p.detach().reshape(-1) for p in wm.parameters()]).cpu()
    diff = (single - r["wm"]).abs()
    assert float((diff > 1e-6).float().mean()) < 2e-3 and float(diff.max()) <= 2.1 * wm._config.model_lr


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (gloo here: two ranks share the
    one device of the test box; RCCL needs a device per rank) and reports the rank count it ran with; asking for more
    ranks than devices under the RCCL backend is refused instead of printing a one-GPU line."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3",
                        "--config", "tiny", "--no-cpu-baseline"], env=dict(env, DV3_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == "gloo" and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2 * common.SHAPES["tiny"]["B"] and out["value"] > 0
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1"],
                           env=dict(env, DV3_DIST_BACKEND="nccl"), capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and not any(ln.startswith("{") for ln in r.stdout.splitlines())


def _runner_worker(rank, world, port, outdir, mode):
    """Six updates through dv3hip.graph.UpdateRunner on rank-specific batches: "pipelined" (step_pipelined + flush, lanes
    schedule), "serial" (one update after the other) or "split" (serial with the world-model bucket cut into two
    all-reduces, UpdateRunner.dp_split)."""
    import torch.distributed as dist

    import tools
    from dv3hip import shapes
    from dv3hip.graph import UpdateRunner

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, wm, beh = Hh.build_models(NAME)
        tools.default_rng("cuda:0", seed=100 + rank)
        r = UpdateRunner(wm, beh, warm=1)
        r.dp_split = mode == "split"
        for i in range(6):
            data = {k: torch.from_numpy(v).cuda() for k, v in shapes.synthetic_batch(NAME, seed=10 * rank + i).items()}
            (r.step_pipelined if mode == "pipelined" else r.step)(data)
        r.flush()
        torch.cuda.synchronize()
        assert r.use_graph and (r._pipe is not None) == (mode == "pipelined")
        flat = lambda ps: torch.cat([p.detach().reshape(-1) for p in ps]).cpu()
        torch.save({"wm": flat(wm.parameters()), "beh": flat(list(beh.actor.parameters()) + list(beh.value.parameters())),
                    "ema": beh.ema_vals.cpu(), "norm": float(r.wm_metrics["model_grad_norm"])},
                   os.path.join(outdir, f"{mode}{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_through_the_update_runner_stay_in_sync():
    """Data parallel through the hipGraph runner, two gloo ranks on different batches: the replicas (world model, actor,
    critic) and the return-normalisation EMA -- which rides in the tail of the critic's gradient bucket -- are
    bit-identical on both ranks after six updates (the clipping norm is summed in a fixed order, dv3_sumsq_ordered: with
    an atomic sum the replicas' norms differed in the last bit and clipped updates drifted apart by 1e-9 each), in the
    pipelined schedule as one update after the other, and with the world-model bucket cut into two all-reduces."""
    import torch.multiprocessing as mp

    world, res = 2, {}
    with tempfile.TemporaryDirectory() as outdir:
        for mode in ("serial", "split", "pipelined"):
            mp.spawn(_runner_worker, args=(world, _free_port(), outdir, mode), nprocs=world, join=True)
            res[mode] = [torch.load(os.path.join(outdir, f"{mode}{r}.pt"), weights_only=True) for r in range(world)]
    for mode, (a, b) in res.items():
        for k in ("wm", "beh", "ema"):
            assert torch.equal(a[k], b[k]), f"{mode}: replicas differ in {k}"
        assert a["norm"] == b["norm"], mode
    assert float(res["serial"][0]["ema"].abs().sum()) > 0
    # across RUNS the numbers agree up to the atomic summation order of the reverse scan (two runs of one schedule differ
    # by as much): the cut bucket and the pipelined schedule compute the serial run's update
    for mode in ("split", "pipelined"):
        d = (res["serial"][0]["wm"] - res[mode][0]["wm"]).abs()
        assert float(d.max()) <= 2.1 * 6 * 1e-4 and float((d > 1e-5).float().mean()) < 5e-3, \
            (mode, float(d.max()), float((d > 1e-5).float().mean()))
        assert torch.allclose(res["serial"][0]["ema"], res[mode][0]["ema"], rtol=1e-4, atol=1e-6), mode
