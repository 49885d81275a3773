"""Data-parallel path on CPU: two gloo ranks exercise the flat gradient bucket + all-reduce logic that runs
over RCCL on the GPUs (one SUM all-reduce per optimizer between backward and clipping; the 1/world scale is
applied by the optimizer kernel).  The Adam arithmetic itself is checked against the oracle on the GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dv3hip.params import ParamBucket
from oracle import dv3_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(7, 5), (5,), (3, 2, 4, 4), (1, 9), (11,)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = _make_params()  # identical replica on every rank
        before = [p.detach().clone() for p in params]
        b = ParamBucket("model", params, allow_cpu=True).ensure()
        # flattening must not change values and must alias storage
        for p, q in zip(params, before):
            assert torch.equal(p.detach(), q)
        assert b.numel() == sum(p.numel() for p in params)
        assert all(p.data_ptr() >= b.flat.data_ptr() for p in params)
        # each rank's "backward" writes rank-dependent gradients through the .grad views
        b.zero_grad()
        g = torch.Generator().manual_seed(100 + rank)
        local = []
        for p in params:
            gr = torch.randn(p.shape, generator=g)
            p.grad.copy_(gr)
            local.append(gr)
        scale = b.allreduce()
        assert scale == 1.0 / world
        gathered = [None] * world
        dist.all_gather_object(gathered, [l.numpy() for l in local])
        for i, p in enumerate(params):
            want = sum(torch.from_numpy(gathered[r][i]) for r in range(world))
            assert torch.allclose(p.grad, want, atol=1e-6), i
        # padding between tensors stays zero (it enters the global norm)
        assert float(b.grad.abs().sum()) == pytest.approx(float(sum(p.grad.abs().sum() for p in params)), rel=1e-6)
        # mean-gradient step == what a single rank would do with the mean gradient (oracle Adam)
        mean_grads = [p.grad * scale for p in params]
        ref = [q.clone() for q in before]
        st = dict(step=0, m=[torch.zeros_like(q) for q in ref], v=[torch.zeros_like(q) for q in ref])
        norm = O.clip_and_adam(ref, mean_grads, st, lr=1e-2, eps=1e-8, clip=1.0)
        out[rank] = (float(norm), [r.numpy() for r in ref])
    finally:
        dist.destroy_process_group()


def test_two_rank_bucket_allreduce_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert set(out.keys()) == {0, 1}
    # both ranks reach bit-identical parameters (replicas stay in sync without a broadcast)
    n0, p0 = out[0]
    n1, p1 = out[1]
    assert n0 == n1
    for a, b in zip(p0, p1):
        assert (a == b).all()


def test_single_rank_allreduce_is_noop():
    params = _make_params()
    b = ParamBucket("x", params, allow_cpu=True).ensure()
    for p in params:
        p.grad.fill_(2.0)
    assert b.allreduce() == 1.0
    assert all(float(p.grad.mean()) == 2.0 for p in params)


def test_bucket_rebuilds_after_module_to():
    """Module.to()/load_state_dict style re-assignment of .data must not leave stale views behind."""
    params = _make_params()
    b = ParamBucket("x", params, allow_cpu=True).ensure()
    flat0 = b.flat
    params[1].data = params[1].data.clone()  # storage moved (what Module._apply does)
    b.ensure()
    assert b.flat is not flat0
    assert params[1].data_ptr() >= b.flat.data_ptr()


def _split_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        for mode in ("single", "split"):
            params = _make_params()
            b = ParamBucket("model", params, allow_cpu=True, extra=2).ensure()
            assert b.tail.shape == (2,) and b.tail.data_ptr() == b.grad.data_ptr() + 4 * b.grad.numel()
            g = torch.Generator().manual_seed(500 + rank)
            for p in params:
                p.grad.copy_(torch.randn(p.shape, generator=g) * 1e3)
            b.tail.copy_(torch.tensor([0.25 + rank, -3.0 * (rank + 1)]))  # (what ImagBehavior.ema_to_wire puts there)
            if mode == "single":
                assert b.allreduce() == 1.0 / world
            else:
                cut = b.offset_of(params[2])  # two collectives, the second half first (graph.UpdateRunner._wm_half)
                assert 0 < cut < b.grad.numel()
                w1 = b.allreduce_range(cut, None, async_op=True)
                w0 = b.allreduce_range(0, cut, async_op=True)
                w1.wait(), w0.wait()
            res[mode] = (b.grad.clone(), b.tail.clone())
        (g0, t0), (g1, t1) = res["single"], res["split"]
        # a sum over ranks is elementwise: the two ranges reduce to exactly what the one call does, tail included
        assert torch.equal(g0, g1) and torch.equal(t0, t1)
        want = torch.tensor([sum(0.25 + r for r in range(world)), sum(-3.0 * (r + 1) for r in range(world))])
        assert torch.equal(t0, want)
        out[rank] = True
    finally:
        dist.destroy_process_group()


def test_two_rank_split_allreduce_equals_single_and_carries_the_tail():
    """Data-parallel readiness (no hardware needed): the world-model bucket cut into two collectives reduces bit for bit
    to what the single all-reduce gives, and the two EMA floats ride in the critic bucket's tail through the same
    call (three collectives per update instead of four)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_split_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert set(out.keys()) == {0, 1}
