"""Per-kernel parity: every libdv3hip entry point against the CPU oracle's math on seeded inputs.

Floating-point bar (BASELINE.json north_star): fp32, 1e-4.  GEMM inputs are O(1) with K up to 4608,
so absolute tolerances scale with sqrt(K).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dv3_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    from dv3hip import ops as _ops

    return _ops


def dev(x):
    return x.cuda()


def assert_close(got, ref, tol=TOL, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item() if got.numel() else 0.0
    scale = max(1.0, ref.abs().max().item() if ref.numel() else 1.0)
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


# ------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [
    # M, N, K
    (1024, 512, 1030),   # img_in: K not a multiple of anything
    (1024, 1536, 1024),  # GRU
    (16, 1536, 1024),    # observe-scan GRU (M = batch)
    (16, 512, 4608),     # obs_out
    (1024, 255, 512),    # reward head: N odd
    (1024, 6, 512),      # actor mean
    (100, 70, 50),       # ragged everything
    (33, 129, 17),
    (1, 1, 1),
    (15360, 512, 1536),  # behaviour heads
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_gemm_nt_bias(ops, M, N, K, tile):
    if M * N * K > 5e9 and tile in (1, 2, 5, 6, 8, 9):
        pytest.skip("large shape: default tile only")
    if tile == 3 and M > 32:
        pytest.skip("skinny path is for M <= 32")
    if tile == 7 and N > 32:
        pytest.skip("narrow-output path is for N <= 32")
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    C = torch.full((M, N), float("nan")).cuda()
    ops.gemm(dev(A), dev(W), C, bias=dev(b), tile=tile)
    assert_close(C, A @ W.t() + b, what=f"gemm_nt {M}x{N}x{K} tile {tile}")


@pytest.mark.parametrize("M,N,K", [(1024, 1030, 512), (16, 1024, 1536), (77, 45, 130), (1024, 512, 255),
                                   (16, 512, 1536), (3, 19, 37), (32, 1030, 512), (1, 512, 1024), (24, 6, 512)])
def test_gemm_nn_dgrad(ops, M, N, K):
    g = torch.Generator().manual_seed(1)
    dY = torch.randn(M, K, generator=g)
    W = torch.randn(K, N, generator=g) / math.sqrt(K)
    C = torch.empty(M, N).cuda()
    ops.gemm(dev(dY), dev(W), C, transB=False)
    assert_close(C, dY @ W, what="gemm_nn")


@pytest.mark.parametrize("M,N,K", [(16, 512, 1536), (16, 1024, 1536), (16, 512, 1024), (3, 19, 37), (32, 1030, 512),
                                   (1, 512, 1024), (16, 512, 1030), (7, 33, 4608)])
@pytest.mark.parametrize("transB", [False, True])
@pytest.mark.parametrize("mode", [True, "atomic"])
def test_gemm_few_rows_accumulate(ops, M, N, K, transB, mode):
    """Few-row GEMM adding into C (strided C and bias included); "atomic" = K split over workgroups."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn((N, K) if transB else (K, N), generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    C0 = torch.randn(M, N + 5, generator=g)
    C = dev(C0.clone())
    ops.gemm(dev(A), dev(W), C[:, :N], transB=transB, bias=dev(b), accumulate=mode)
    want = C0.clone()
    want[:, :N] += A @ (W.t() if transB else W) + b
    assert_close(C, want, what=f"gemm few rows accumulate={mode}")


@pytest.mark.parametrize("M,N,K1,K2", [(1024, 512, 1024, 6), (1024, 1536, 512, 512), (77, 45, 32, 3), (100, 130, 48, 0),
                                       (33, 64, 16, 17), (1024, 512, 512, 0)])
@pytest.mark.parametrize("transB", [True, False])
@pytest.mark.parametrize("acc", [False, True])
def test_gemm_register_direct_kernel(ops, M, N, K1, K2, transB, acc):
    """tile 9 (no LDS staging, 16x16x4 MFMA, K over the waves): both B layouts, [A | A2] concat, bias,
    accumulate, ragged M / N / K."""
    g = torch.Generator().manual_seed(M + N + K1 + K2)
    K = K1 + K2
    A, A2 = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g)
    W = torch.randn((N, K) if transB else (K, N), generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    C = dev(C0.clone())
    ops.gemm(dev(A), dev(W), C, A2=dev(A2) if K2 else None, transB=transB, bias=dev(b), accumulate=acc, tile=9)
    want = torch.cat([A, A2], -1) @ (W.t() if transB else W) + b + (C0 if acc else 0)
    assert_close(C, want, what="gemm tile 9")


@pytest.mark.parametrize("M,N,K1,K2", [(1024, 1536, 512, 512), (1024, 1024, 512, 0), (1024, 512, 512, 1024),
                                       (2048, 1536, 1024, 0), (77, 100, 64, 32), (130, 200, 96, 0), (4096, 3072, 2048, 0)])
@pytest.mark.parametrize("acc", [False, True])
@pytest.mark.parametrize("tile", [11, 12, 13, 14, 15])
def test_gemm_kcontiguous_lds_tile(ops, M, N, K1, K2, acc, tile):
    """tiles 11-15 (both operands k-contiguous in LDS, ds_read_b128 fragments, 16x16x4 MFMA): y = [A | A2] W^T + b,
    accumulate, ragged M / N; 11 picks the tile shape by size, 12 / 13 / 14 / 15 force 64x96 / 64x64 / 32x64 / 128x128."""
    if tile not in (11, 15) and M * N * (K1 + K2) > 5e9:
        pytest.skip("large shape: default and 128x128 tiles only")
    g = torch.Generator().manual_seed(M + N + K1 + K2)
    K = K1 + K2
    A, A2 = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    C = dev(C0.clone())
    ops.gemm(dev(A), dev(W), C, A2=dev(A2) if K2 else None, bias=dev(b), accumulate=acc, tile=tile)
    want = torch.cat([A, A2], -1) @ W.t() + b + (C0 if acc else 0)
    assert_close(C, want, what=f"gemm tile {tile}")


@pytest.mark.parametrize("M,N,K1,K2", [(1024, 1536, 512, 512), (2048, 3072, 1536, 0), (2048, 1024, 1024, 0),
                                       (15360, 255, 512, 0), (14336, 1024, 512, 0), (130, 200, 96, 32)])
def test_l16_tiles_agree_bit_for_bit(ops, M, N, K1, K2):
    """The k-contiguous LDS tiles (12-17: 64x96, 64x64, 32x64, 128x128, 128x64, 64x128) run ONE K loop -- 32-wide K
    tiles, v_mfma_f32_16x16x4_f32 in ascending k, the [A | A2] seam on a K-tile boundary -- so which of them a launch
    takes changes no bit of the result.  ops.gemm relies on that: on a 128-CU lane of the pipelined update it picks
    another tile than on the whole chip (ops._LANE_TILES) and the lanes schedule still computes exactly the serial update."""
    g = torch.Generator().manual_seed(M + N)
    K = K1 + K2
    A, A2 = dev(torch.randn(M, K1, generator=g)), dev(torch.randn(M, K2, generator=g)) if K2 else None
    W, b = dev(torch.randn(N, K, generator=g) / math.sqrt(K)), dev(torch.randn(N, generator=g))
    outs = {}
    for tile in (12, 13, 14, 15, 16, 17):
        C = torch.empty(M, N, device="cuda")
        ops.gemm(A, W, C, A2=A2, bias=b, tile=tile)
        outs[tile] = C
    for tile, C in outs.items():
        assert torch.equal(C, outs[14]), f"tile {tile} differs from the 32x64 tile by {float((C - outs[14]).abs().max()):.3e}"


def test_gemm_picks_lane_tiles_on_a_cu_masked_stream(ops):
    """On a 128-CU lane ops.gemm takes the tile measured best there, on the whole chip the whole-chip choice; same bits."""
    from dv3hip import engine

    ln = engine.Lanes.get(torch.device("cuda", 0))
    if ln is None:
        pytest.skip("no CU-masked streams on this device")
    M, N, K = 1024, 1536, 1024
    g = torch.Generator().manual_seed(5)
    A, W = dev(torch.randn(M, K, generator=g)), dev(torch.randn(N, K, generator=g) / 32.0)
    outs, keys = [], []
    for st in (ln.streams["whole"], ln.streams["side"]):
        C = torch.empty(M, N, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            ops.PROFILE.start()
            ops.gemm(A, W, C)
            keys.append(list(ops.PROFILE.stop()))
        outs.append(C)
    assert keys[0] == ["gemm_kernel<l16_32x64,tA=0,tB=1>"] and keys[1] == ["gemm_kernel<l16_64x96,tA=0,tB=1>"], keys
    assert torch.equal(outs[0], outs[1])


def test_grouped_weight_gradients_equal_the_single_launches(ops):
    """ops.gemm_group (dv3_gemm_tn_grouped_f32): C_g += A_g^T B_g for a mixed bag of shapes in one grid -- ragged edges,
    row strides wider than the matrix, non-zero C to accumulate into -- against an fp64 product; reproducible (no
    atomics: two runs agree bit for bit); products the group does not take (long reductions, few rows) are launched as
    before; overlapping outputs and more than 48 products end the current grid instead of racing."""
    g = torch.Generator().manual_seed(11)
    shapes = [(512, 512, 1024), (512, 1024, 1024), (1536, 512, 1024), (255, 512, 1024), (100, 70, 96), (64, 64, 37),
              (4096, 512, 1024), (33, 1030, 1000)]

    def operands():
        ops_ = []
        for M, N, K in shapes:
            a = dev(torch.randn(K, M + 8, generator=g))[:, :M]   # (row stride wider than the matrix)
            b = dev(torch.randn(K, N + 4, generator=g))[:, :N]
            c = dev(torch.randn(M, N + 12, generator=g))[:, :N]
            ops_.append((a, b, c))
        return ops_

    trip = operands()
    want = [(c.double() + a.double().t() @ b.double()) for a, b, c in trip]
    outs = []
    for rep in range(2):
        cs = [c.clone() for _, _, c in trip]
        cs = [dev(torch.zeros(c.shape[0], c.shape[1] + 12))[:, :c.shape[1]].copy_(c) for c in cs]
        ops.PROFILE.start()
        ops.PROFILE.by_shape = False
        with ops.gemm_group():
            for (a, b, _), c in zip(trip, cs):
                ops.gemm(a, b, c, transA=True, transB=False, accumulate=True)
        keys = ops.PROFILE.stop()
        assert list(keys) == ["gemm_tn_grouped_kernel<64x64x32>"] and keys["gemm_tn_grouped_kernel<64x64x32>"]["launches"] == 1
        outs.append(cs)
    for (M, N, K), w, c0, c1 in zip(shapes, want, outs[0], outs[1]):
        err = float((c0.double() - w).abs().max())
        assert err <= 2e-5 * (1.0 + float(w.abs().max())), ((M, N, K), err)
        assert torch.equal(c0, c1), ("not reproducible", (M, N, K))
    # not taken: a 14 k-row reduction, a few-row output; both still computed
    a, b = dev(torch.randn(14336, 512, generator=g)), dev(torch.randn(14336, 256, generator=g))
    c = torch.zeros(512, 256, device="cuda")
    a2, b2 = dev(torch.randn(64, 16, generator=g)), dev(torch.randn(64, 48, generator=g))
    c2 = torch.zeros(16, 48, device="cuda")
    ops.PROFILE.start()
    with ops.gemm_group():
        ops.gemm(a, b, c, transA=True, transB=False, accumulate=True)
        ops.gemm(a2, b2, c2, transA=True, transB=False, accumulate=True)
    assert not any(k.startswith("gemm_tn_grouped") for k in ops.PROFILE.stop())
    assert float((c.double() - a.double().t() @ b.double()).abs().max()) <= 2e-2
    assert float((c2.double() - a2.double().t() @ b2.double()).abs().max()) <= 1e-4
    # overlapping outputs: the second product accumulates on top of the first one's result (two grids, in order)
    a, b = dev(torch.randn(256, 128, generator=g)), dev(torch.randn(256, 192, generator=g))
    c = torch.zeros(128, 192, device="cuda")
    ops.PROFILE.start()
    with ops.gemm_group():
        ops.gemm(a, b, c, transA=True, transB=False, accumulate=True)
        ops.gemm(a[:, :64], b, c[:64], transA=True, transB=False, accumulate=True)
    k = ops.PROFILE.stop()
    assert k["gemm_tn_grouped_kernel<64x64x32>"]["launches"] == 2
    w = a.double().t() @ b.double()
    w[:64] += a[:, :64].double().t() @ b.double()
    assert float((c.double() - w).abs().max()) <= 1e-3
    # the cluster's bias gradients ride along: one more grid
    g2 = torch.Generator().manual_seed(12)
    xs = [dev(torch.randn(R_, N_ + 4, generator=g2))[:, :N_] for R_, N_ in ((1024, 512), (1000, 255), (14336, 512), (64, 33))]
    outs2 = [dev(torch.randn(x.shape[1], generator=g2)) for x in xs]
    want2 = [o.double() + x.double().sum(0) for o, x in zip(outs2, xs)]
    ops.PROFILE.start()
    with ops.gemm_group():
        for x, o in zip(xs, outs2):
            ops.colsum(x, o, accumulate=True)
    k = ops.PROFILE.stop()
    assert list(k) == ["dv3_colsum_grouped"] and k["dv3_colsum_grouped"]["launches"] == 1
    for o, w in zip(outs2, want2):
        assert float((o.double() - w).abs().max()) <= 1e-4 * (1.0 + float(w.abs().max()))
    # more than 48 products
    cs = [torch.zeros(64, 64, device="cuda") for _ in range(50)]
    a, b = dev(torch.randn(128, 64, generator=g)), dev(torch.randn(128, 64, generator=g))
    ops.PROFILE.start()
    with ops.gemm_group():
        for c in cs:
            ops.gemm(a, b, c, transA=True, transB=False, accumulate=True)
    assert ops.PROFILE.stop()["gemm_tn_grouped_kernel<64x64x32>"]["launches"] == 2
    w = (a.double().t() @ b.double())
    assert all(float((c.double() - w).abs().max()) <= 1e-4 for c in cs)


@pytest.mark.parametrize("M", [33, 50, 64, 100, 128])
@pytest.mark.parametrize("transB", [True, False])
@pytest.mark.parametrize("acc", [False, True, "atomic"])
def test_gemm_few_row_kernel_up_to_128_rows(ops, M, transB, acc):
    """tile 3 beyond 32 rows (row blocks of 32 over grid.z: the acting step's encoder on a few images, replay batch 64):
    [A | A2] concat, bias, both weight layouts, accumulate modes."""
    g = torch.Generator().manual_seed(M)
    K1, K2, N = 96, 48, 300
    A, A2 = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g)
    W = torch.randn((N, K1 + K2) if transB else (K1 + K2, N), generator=g) / math.sqrt(K1 + K2)
    b = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    C = dev(C0.clone())
    ops.gemm(dev(A), dev(W), C, A2=dev(A2), transB=transB, bias=dev(b), accumulate=acc, tile=3)
    want = torch.cat([A, A2], -1) @ (W.t() if transB else W) + b + (C0 if acc else 0)
    assert_close(C, want, what=f"few-row gemm, {M} rows")


def test_gemm_kcontiguous_lds_tiles_random_shapes(ops):
    """Seeded sweep of ragged shapes over the five k-contiguous LDS tile shapes (11-15): M, N anything, K and the
    [A | A2] seam multiples of 32, operands as row slices of wider buffers, accumulate on and off."""
    rs = np.random.RandomState(123)
    g = torch.Generator().manual_seed(123)
    for case in range(40):
        tile = int(rs.choice([11, 12, 13, 14, 15]))
        M, N = int(rs.randint(1, 400)), int(rs.randint(1, 400))
        K1, K2 = 32 * int(rs.randint(1, 9)), 32 * int(rs.randint(0, 5))
        K = K1 + K2
        pad = 4 * int(rs.randint(0, 3))
        Aw, A2w = torch.randn(M, K1 + pad, generator=g), torch.randn(M, max(K2, 4) + pad, generator=g)
        W = torch.randn(N, K, generator=g) / math.sqrt(K)
        b = torch.randn(N, generator=g) if rs.rand() < 0.5 else None
        acc = bool(rs.rand() < 0.5)
        C0 = torch.randn(M, N, generator=g)
        C = dev(C0.clone())
        Ad, A2d = dev(Aw)[:, :K1], (dev(A2w)[:, :K2] if K2 else None)
        ops.gemm(Ad, dev(W), C, A2=A2d, bias=None if b is None else dev(b), accumulate=acc, tile=tile)
        want = torch.cat([Aw[:, :K1], A2w[:, :K2]], -1) @ W.t() + (0 if b is None else b) + (C0 if acc else 0)
        assert_close(C, want, what=f"case {case}: tile {tile} {M}x{N}x{K1}+{K2} pad {pad} acc {acc}")


def test_conv_kcontiguous_lds_tiles_random_shapes(ops):
    """Seeded sweep for the implicit-GEMM convolutions on the k-contiguous LDS tiles (conv_s2 with Co >= 64, convT_s2
    with Co >= 128 or a multiple of 96; Ci a multiple of 32) against torch's conv2d / conv_transpose2d."""
    rs = np.random.RandomState(7)
    g = torch.Generator().manual_seed(7)
    for case in range(14):
        Nimg, H = int(rs.randint(1, 6)), int(rs.choice([2, 4, 8, 16]))
        Ci = 32 * int(rs.randint(1, 5))
        Co = int(rs.choice([64, 72, 96, 128, 160, 192, 256, 384]))
        x = torch.randn(Nimg, Ci, H, H, generator=g)
        w = torch.randn(Co, Ci, 4, 4, generator=g) / math.sqrt(16 * Ci)
        wp = torch.empty(Co, 16 * Ci, device="cuda")
        ops.pack_conv_weight(dev(w), wp, transposed=False)
        y = torch.empty(Nimg, H // 2, H // 2, Co, device="cuda")
        ops.conv_s2_fwd(dev(nhwc(x)), wp, y, Ci=Ci, Co=Co)
        assert_close(y, nhwc(F.conv2d(F.pad(x, [1, 1, 1, 1]), w, None, 2)), what=f"case {case}: conv {Nimg}x{H} {Ci}->{Co}")
        if Co >= 128 or Co % 96 == 0:
            wt = torch.randn(Ci, Co, 4, 4, generator=g) / math.sqrt(4 * Ci)
            bias = torch.randn(Co, generator=g)
            wpt = torch.empty(4, Co, 4 * Ci, device="cuda")
            ops.pack_conv_weight(dev(wt), wpt, transposed=True)
            yt = torch.empty(Nimg, 2 * H, 2 * H, Co, device="cuda")
            ops.convT_s2_fwd(dev(nhwc(x)), wpt, yt, Ci=Ci, Co=Co, bias=dev(bias), out_add=0.5)
            assert_close(yt, nhwc(F.conv_transpose2d(x, wt, bias, 2, padding=1) + 0.5),
                         what=f"case {case}: convT {Nimg}x{H} {Ci}->{Co}")


@pytest.mark.parametrize("acc", [False, True])
def test_gemm_big_data_gradient_takes_the_transposed_copy_path(ops, acc):
    """C (+)= A @ B with B [K][N] and >= 1.4e10 flops: ops.gemm transposes B into scratch and runs the k-contiguous LDS
    tile; B a row slice of a wider weight (as the heads pass it)."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 8192, 1024, 1024
    A = torch.randn(M, K, generator=g)
    Wfull = torch.randn(K + 64, N, generator=g) / math.sqrt(K)
    C0 = torch.randn(M, N, generator=g)
    C = dev(C0.clone())
    Wd = dev(Wfull)
    ops.gemm(dev(A), Wd[64:], C, transB=False, accumulate=acc)
    assert_close(C, A @ Wfull[64:] + (C0 if acc else 0), tol=2e-5, what="big NN gemm")


@pytest.mark.parametrize("M,n1,n2,K", [(1024, 512, 512, 1536), (200, 48, 80, 64), (77, 16, 100, 96)])
def test_gemm_split_output(ops, M, n1, n2, K):
    """dv3_gemm_split_f32: one product, columns [0, n1) overwrite C, columns [n1, n1+n2) accumulate into a strided C2."""
    g = torch.Generator().manual_seed(M + n1 + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(n1 + n2, K, generator=g) / math.sqrt(K)
    C = torch.full((M, n1), float("nan")).cuda()
    wide0 = torch.randn(M, n2 + 8, generator=g)
    wide = dev(wide0.clone())
    assert ops.gemm_split_ok(dev(A), dev(W))
    ops.gemm_split(dev(A), dev(W), C, wide[:, 8:], accumulate2=True)
    ref = A @ W.t()
    assert_close(C, ref[:, :n1], what="first block")
    assert_close(wide[:, 8:], wide0[:, 8:] + ref[:, n1:], what="second block (accumulated)")
    assert torch.equal(wide[:, :8].cpu(), wide0[:, :8])


@pytest.mark.parametrize("rows,Nout,Kin", [(1024, 512, 1030), (1024, 1536, 1024), (333, 70, 45), (14336, 255, 512)])
def test_gemm_tn_wgrad_accumulate(ops, rows, Nout, Kin):
    g = torch.Generator().manual_seed(2)
    dY = torch.randn(rows, Nout, generator=g) / math.sqrt(rows)
    X = torch.randn(rows, Kin, generator=g)
    C0 = torch.randn(Nout, Kin, generator=g)
    C = dev(C0.clone())
    ops.gemm(dev(dY), dev(X), C, transA=True, transB=False, accumulate=True)
    assert_close(C, C0 + dY.t() @ X, what="gemm_tn")
    C = dev(C0.clone())  # register-direct weight-gradient kernel
    ops.gemm(dev(dY), dev(X), C, transA=True, transB=False, accumulate=True, tile=10)
    assert_close(C, C0 + dY.t() @ X, what="gemm_tn direct")
    C = torch.full((Nout, Kin), float("nan")).cuda()
    ops.gemm(dev(dY), dev(X), C, transA=True, transB=False, accumulate=False, tile=10)
    assert_close(C, dY.t() @ X, what="gemm_tn direct, overwrite")


@pytest.mark.parametrize("M,K1,K2,N", [(1024, 1024, 6, 512), (1024, 512, 512, 1536), (16, 512, 4096, 512),
                                       (48, 16, 3, 16), (9, 64, 5, 33), (16, 1024, 6, 512), (16, 512, 512, 1536),
                                       (3, 16, 3, 16), (32, 48, 7, 20)])
def test_gemm_two_segment_concat(ops, M, K1, K2, N):
    """[A | A2] @ W^T == torch.cat([A, A2], -1) @ W^T, including row-strided views."""
    g = torch.Generator().manual_seed(3)
    big = torch.randn(M, K1 + 11, generator=g)
    A = big[:, :K1]  # row-strided view
    A2 = torch.randn(M, K2, generator=g)
    W = torch.randn(N, K1 + K2, generator=g) / math.sqrt(K1 + K2)
    C = torch.empty(M, N).cuda()
    bigd = dev(big)
    ops.gemm(bigd[:, :K1], dev(W), C, A2=dev(A2))
    assert_close(C, torch.cat([A, A2], -1) @ W.t(), what="gemm_concat")


def test_gemm_strided_output_slice(ops):
    """dgrad into a column slice of a wider buffer (the backward of torch.cat)."""
    g = torch.Generator().manual_seed(4)
    dY = torch.randn(64, 48, generator=g)
    W = torch.randn(48, 100, generator=g)
    buf = torch.zeros(64, 130).cuda()
    ops.gemm(dev(dY), dev(W)[:, 30:], buf[:, 10:80], transB=False)
    ref = torch.zeros(64, 130)
    ref[:, 10:80] = dY @ W[:, 30:]
    assert_close(buf, ref, what="gemm slice")


def test_gemm_is_exact_fp32_on_integers(ops):
    """MFMA fp32 is an exact fma chain: small-integer data must reproduce bit-exactly."""
    g = torch.Generator().manual_seed(5)
    A = torch.randint(-4, 5, (130, 200), generator=g).float()
    W = torch.randint(-4, 5, (70, 200), generator=g).float()  # asymmetric on purpose (layout check)
    C = torch.empty(130, 70).cuda()
    ops.gemm(dev(A), dev(W), C)
    assert torch.equal(C.cpu(), A @ W.t())


def test_gemm_rejects_bad_shapes(ops):
    A = torch.zeros(8, 16).cuda()
    W = torch.zeros(4, 17).cuda()
    C = torch.zeros(8, 4).cuda()
    with pytest.raises(ValueError):
        ops.gemm(A, W, C)
    with pytest.raises(TypeError):
        ops.gemm(A.double(), W, C)


# ------------------------------------------------------------------------------------------ LN / GRU
@pytest.mark.parametrize("R,N", [(1024, 512), (16, 512), (1000, 32), (257, 64), (100, 128), (64, 256), (7, 1536),
                                 (33, 16), (50, 3), (5, 1024), (3, 2048), (4, 700)])
@pytest.mark.parametrize("act", [True, False])
def test_ln_act_fwd_bwd(ops, R, N, act):
    g = torch.Generator().manual_seed(R + N)
    x = (torch.randn(R, N, generator=g) * 2 + 0.3).requires_grad_(True)
    gam = (1 + 0.1 * torch.randn(N, generator=g)).requires_grad_(True)
    bet = (0.1 * torch.randn(N, generator=g)).requires_grad_(True)
    dy = torch.randn(R, N, generator=g)
    y_ref = O.layer_norm(x, gam, bet)
    if act:
        y_ref = F.silu(y_ref)
    y_ref.backward(dy)
    y = torch.empty(R, N).cuda()
    mean = torch.empty(R).cuda()
    rstd = torch.empty(R).cuda()
    xd, gd, bd = dev(x.detach()), dev(gam.detach()), dev(bet.detach())
    ops.ln_act_fwd(xd, gd, bd, y, mean, rstd, act=act)
    assert_close(y, y_ref, what="ln fwd")
    dx = torch.empty(R, N).cuda()
    dg = torch.zeros(N).cuda()
    db = torch.zeros(N).cuda()
    ops.ln_act_bwd(dev(dy), xd, gd, bd, mean, rstd, dx, dg, db, act=act)
    assert_close(dx, x.grad, what="ln dx")
    assert_close(dg, gam.grad, tol=2e-4, what="ln dgamma")
    assert_close(db, bet.grad, tol=2e-4, what="ln dbeta")


@pytest.mark.parametrize("R,N", [(70000, 32), (40000, 64), (20000, 128), (9000, 256), (5000, 512), (14336, 512)])
def test_ln_bwd_parameter_gradients_in_two_stages(ops, R, N):
    """dv3_ln_act_bwd_ws (big activations): every row block writes its column sums to its own row of a partial buffer, a
    second launch adds the rows up.  Against torch autograd; three launches in a row accumulate; the same from a hipGraph
    captured on a CU-masked lane (a stream that gets a partial buffer of its own)."""
    from dv3hip import engine

    assert R * N >= ops._LN_PART_MIN and ops._LN_TWO_STAGE
    g = torch.Generator().manual_seed(R + N)
    x = (torch.randn(R, N, generator=g) * 2 + 0.3).requires_grad_(True)
    gam = (1 + 0.1 * torch.randn(N, generator=g)).requires_grad_(True)
    bet = (0.1 * torch.randn(N, generator=g)).requires_grad_(True)
    dy = torch.randn(R, N, generator=g)
    F.silu(O.layer_norm(x, gam, bet)).backward(dy)
    xd, gd, bd, dyd = dev(x.detach()), dev(gam.detach()), dev(bet.detach()), dev(dy)
    y, mean, rstd = torch.empty(R, N).cuda(), torch.empty(R).cuda(), torch.empty(R).cuda()
    ops.ln_act_fwd(xd, gd, bd, y, mean, rstd, act=True)
    dx, dg, db = torch.empty(R, N).cuda(), torch.zeros(N).cuda(), torch.zeros(N).cuda()
    ops.PROFILE.start()
    for _ in range(3):
        ops.ln_act_bwd(dyd, xd, gd, bd, mean, rstd, dx, dg, db, act=True)
    assert list(ops.PROFILE.stop()) == ["dv3_ln_act_bwd"]
    scale = float(gam.grad.abs().max()) + float(bet.grad.abs().max())
    assert_close(dx, x.grad, what="ln dx")
    assert float((dg / 3 - dev(gam.grad)).abs().max()) <= 3e-5 * scale + 1e-4, "d-gamma"
    assert float((db / 3 - dev(bet.grad)).abs().max()) <= 3e-5 * scale + 1e-4, "d-beta"
    ln = engine.Lanes.get(torch.device("cuda", 0))
    if ln is None:
        return
    dg2, db2 = torch.zeros(N).cuda(), torch.zeros(N).cuda()
    gr = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    n_bufs = len(ops._LN_PART)
    with torch.cuda.graph(gr, stream=ln.streams["side"]):
        ops.ln_act_bwd(dyd, xd, gd, bd, mean, rstd, dx, dg2, db2, act=True)
    assert len(ops._LN_PART) >= n_bufs
    with torch.cuda.stream(ln.streams["side"]):
        gr.replay()
        gr.replay()
    torch.cuda.synchronize()
    assert float((dg2 / 2 - dev(gam.grad)).abs().max()) <= 3e-5 * scale + 1e-4
    assert float((db2 / 2 - dev(bet.grad)).abs().max()) <= 3e-5 * scale + 1e-4


def test_ln_chw_flatten_matches_reference_order(ops):
    """Encoder output is flattened (C,H,W) (networks.py:494); our activations are NHWC."""
    n_img, G, C = 5, 16, 24
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n_img * G, C, generator=g)
    gam, bet = torch.ones(C), torch.zeros(C)
    y = torch.empty(n_img, C * G).cuda()
    ops.ln_act_fwd(dev(x), dev(gam), dev(bet), y, act=True, chw_group=G)
    ref = F.silu(O.layer_norm(x, gam, bet)).reshape(n_img, G, C).permute(0, 2, 1).reshape(n_img, C * G)
    assert_close(y, ref, what="chw flatten")


@pytest.mark.parametrize("M,De", [(1024, 512), (16, 512), (5, 16), (33, 1024), (3, 40), (64, 4096), (16, 2048),
                                  (130, 3072), (7, 1280)])
def test_gru_fwd_bwd(ops, M, De):
    g = torch.Generator().manual_seed(M + De)
    p_pre = (torch.randn(M, 3 * De, generator=g)).requires_grad_(True)
    h = torch.randn(M, De, generator=g).requires_grad_(True)
    gam = (1 + 0.1 * torch.randn(3 * De, generator=g)).requires_grad_(True)
    bet = (0.1 * torch.randn(3 * De, generator=g)).requires_grad_(True)
    dh_new = torch.randn(M, De, generator=g)
    parts = O.layer_norm(p_pre, gam, bet)
    r, c, u = parts[:, :De], parts[:, De:2 * De], parts[:, 2 * De:]
    r = torch.sigmoid(r)
    c = torch.tanh(r * c)
    u = torch.sigmoid(u - 1)
    ref = u * c + (1 - u) * h
    ref.backward(dh_new)
    pd, hd, gd, bd = dev(p_pre.detach()), dev(h.detach()), dev(gam.detach()), dev(bet.detach())
    hn = torch.empty(M, De).cuda()
    mean, rstd = torch.empty(M).cuda(), torch.empty(M).cuda()
    ops.gru_fwd(pd, gd, bd, hd, hn, mean, rstd)
    assert_close(hn, ref, what="gru fwd")
    dp = torch.empty(M, 3 * De).cuda()
    dh = torch.empty(M, De).cuda()
    dg, db = torch.zeros(3 * De).cuda(), torch.zeros(3 * De).cuda()
    ops.gru_bwd(dev(dh_new), pd, gd, bd, hd, mean, rstd, dp, dh, dg, db)
    assert_close(dp, p_pre.grad, what="gru dp")
    assert_close(dh, h.grad, what="gru dh")
    assert_close(dg, gam.grad, tol=2e-4, what="gru dgamma")
    assert_close(db, bet.grad, tol=2e-4, what="gru dbeta")


# ------------------------------------------------------------------------------------------ categorical
@pytest.mark.parametrize("R,D", [(32768, 32), (48, 4), (1000, 18), (77, 5), (10, 64)])
def test_onehot_sample_mode_and_straight_through(ops, R, D):
    g = torch.Generator().manual_seed(R + D)
    logit = (2 * torch.randn(R, D, generator=g)).requires_grad_(True)
    q = torch.empty(R, D).exponential_(generator=g)
    gs = torch.randn(R, D, generator=g)
    ref = O.onehot_sample(logit, q, 0.01)
    ref.backward(gs)
    out = torch.empty(R, D).cuda()
    idx = torch.empty(R, dtype=torch.int32).cuda()
    ld = dev(logit.detach())
    ops.onehot_sample(ld, out, noise=dev(q), idx=idx, unimix=0.01)
    flips = (out.cpu() != ref.detach()).any(-1).float().mean().item()
    assert flips <= 2e-4, f"sample flip rate {flips}"
    assert torch.equal(out.sum(-1).cpu(), torch.ones(R))
    assert torch.equal(out.argmax(-1).int(), idx)
    dl = torch.zeros(R, D).cuda()
    ops.onehot_st_bwd(ld, dev(gs), dl, unimix=0.01)
    assert_close(dl, logit.grad, what="st bwd")
    # mode + its straight-through (through log p)
    logit2 = logit.detach().clone().requires_grad_(True)
    refm = O.onehot_mode(logit2, 0.01)
    refm.backward(gs)
    ops.onehot_sample(ld, out, unimix=0.01, mode=True)
    assert torch.equal(out.cpu(), torch.nn.functional.one_hot(logit.argmax(-1), D).float())
    ops.onehot_st_bwd(ld, dev(gs), dl, unimix=0.01, mode=True)
    assert_close(dl, logit2.grad, what="mode st bwd")


def test_onehot_sample_philox_is_a_categorical_sampler(ops):
    """In-kernel RNG path: empirical frequencies follow p_hat (chi-square style bound)."""
    D, R = 8, 1 << 18
    logit = torch.tensor([0.0, 1.0, -1.0, 2.0, 0.5, -2.0, 0.0, 1.5]).repeat(R, 1).cuda()
    st = ops.RngStream(torch.device("cuda"), 1234)
    out = torch.empty(R, D).cuda()
    ops.onehot_sample(logit, out, rng=st, unimix=0.01)
    freq = out.mean(0).cpu()
    p = F.softmax(logit[0].cpu(), -1) * 0.99 + 0.01 / D
    assert (freq - p).abs().max().item() < 5e-3
    out2 = torch.empty(R, D).cuda()
    ops.onehot_sample(logit, out2, rng=st, unimix=0.01)  # next counter range of the same stream
    assert (out != out2).any()
    st.commit()
    assert st.state[1].item() == 2 * (R * D // 4 + 1) and st.cursor == 0
    z = torch.empty(1 << 20).cuda()
    ops.fill_normal(z, st)
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1.0) < 5e-3


@pytest.mark.parametrize("rows,S,D", [(1024, 32, 32), (18, 4, 4), (7, 3, 5)])
def test_kl_fwd_bwd(ops, rows, S, D):
    g = torch.Generator().manual_seed(rows)
    post = (1.5 * torch.randn(rows, S, D, generator=g)).requires_grad_(True)
    prior = (1.5 * torch.randn(rows, S, D, generator=g)).requires_grad_(True)
    cfg = O.PathConfig(stoch=S, discrete=D)
    cfg.kl_free = 1.0 if D == 32 else 0.3  # make the clip bite on some rows only
    loss, value, dyn, rep = O.kl_loss(cfg, post, prior)
    up = 1.0 / rows
    (loss.sum() * up).backward()
    kl = torch.empty(rows).cuda()
    ep, eq = torch.empty(rows).cuda(), torch.empty(rows).cuda()
    pd, qd = dev(post.detach()), dev(prior.detach())
    ops.kl_fwd(pd, qd, kl, ep, eq, unimix=0.01)
    assert_close(kl, value, what="kl")
    assert_close(ep, O.onehot_entropy(post, 0.01), what="post ent")
    assert_close(eq, O.onehot_entropy(prior, 0.01), what="prior ent")
    dp, dq = torch.empty_like(pd), torch.empty_like(qd)
    ops.kl_bwd(pd, qd, kl, dp, dq, unimix=0.01, free=cfg.kl_free, dyn_scale=cfg.dyn_scale, rep_scale=cfg.rep_scale,
               upstream=up)
    assert_close(dp, post.grad, tol=1e-5, what="dpost")
    assert_close(dq, prior.grad, tol=1e-5, what="dprior")


def test_onehot_entropy_logprob_fwd_bwd(ops):
    R, D = 500, 18
    g = torch.Generator().manual_seed(9)
    logit = torch.randn(R, D, generator=g).requires_grad_(True)
    x = F.one_hot(torch.randint(0, D, (R,), generator=g), D).float()
    de, dl_ = torch.randn(R, generator=g), torch.randn(R, generator=g)
    cfg = O.PathConfig(actor_dist="onehot")
    lg = O.unimix_logits(logit, 0.01)
    pr = F.softmax(lg, -1)
    ent_ref = -(lg * pr).sum(-1)
    lp_ref = O.onehot_logprob(logit, x, 0.01)
    ((ent_ref * de).sum() + (lp_ref * dl_).sum()).backward()
    ent, lp = torch.empty(R).cuda(), torch.empty(R).cuda()
    ld = dev(logit.detach())
    ops.onehot_ent_logp_fwd(ld, dev(x), ent, lp, unimix=0.01)
    assert_close(ent, ent_ref, what="ent")
    assert_close(lp, lp_ref, what="logp")
    dlog = torch.empty(R, D).cuda()
    ops.onehot_ent_logp_bwd(ld, dev(x), dev(de), dev(dl_), dlog, unimix=0.01)
    assert_close(dlog, logit.grad, what="dlogit")


# ------------------------------------------------------------------------------------------ heads
def test_disc_head_mode_logprob_fwd_bwd(ops):
    R = 3000
    g = torch.Generator().manual_seed(11)
    logits = (2 * torch.randn(R, 255, generator=g)).requires_grad_(True)
    x = torch.cat([torch.randn(R - 6, generator=g) * 30,
                   torch.tensor([0.0, 1e9, -1e9, float(O.symexp(torch.tensor(20.0))), 0.15748, -485165184.0])])
    up_m, up_l = torch.randn(R, generator=g), torch.randn(R, generator=g)
    mode_ref = O.disc_mode(logits).squeeze(-1)
    lp_ref = O.disc_logprob(logits, x)
    ((mode_ref * up_m).sum() + (lp_ref * up_l).sum()).backward()
    ld = dev(logits.detach())
    mode, lp = torch.empty(R).cuda(), torch.empty(R).cuda()
    ops.disc_mode_fwd(ld, mode)
    ops.disc_logprob_fwd(ld, dev(x), lp)
    assert_close(mode, mode_ref, what="disc mode")
    assert_close(lp, lp_ref, what="disc logprob")
    dl = torch.zeros(R, 255).cuda()
    ops.disc_mode_bwd(ld, dev(up_m), dl)
    ops.disc_logprob_bwd(ld, dev(x), dev(up_l), dl, accumulate=True)
    assert_close(dl, logits.grad, what="disc dlogits")


def test_bernoulli_mse_symlog(ops):
    g = torch.Generator().manual_seed(12)
    n = 1000
    l = (3 * torch.randn(n, 1, generator=g)).requires_grad_(True)
    x = (torch.rand(n, 1, generator=g) > 0.5).float()
    up = torch.randn(n, generator=g)
    ref = O.bernoulli_logprob(l, x)
    (ref * up).sum().backward()
    out = torch.empty(n).cuda()
    ld = dev(l.detach().reshape(n))
    ops.bernoulli_logprob_fwd(ld, dev(x.reshape(n)), out)
    assert_close(out, ref, what="bern")
    dl = torch.empty(n).cuda()
    ops.bernoulli_logprob_bwd(ld, dev(x.reshape(n)), dev(up), dl)
    assert_close(dl, l.grad.reshape(n), what="bern bwd")
    # image mse with fused u8 decode
    img = torch.randint(0, 256, (6, 64, 64, 3), generator=g, dtype=torch.uint8)
    recon = torch.rand(6, 64, 64, 3, generator=g).requires_grad_(True)
    loss_ref = ((recon - img.float() / 255.0) ** 2).sum([1, 2, 3])
    (loss_ref.sum() * 0.25).backward()
    loss = torch.empty(6).cuda()
    dr = torch.empty(6, 64, 64, 3).cuda()
    ops.mse_image(dev(recon.detach()), dev(img), loss, dr, upstream=0.25)
    # time-major pairing: recon image t*B+b <-> replay image b*T+t
    B_, T_ = 2, 3
    recon_tm = recon.detach().reshape(B_, T_, 64, 64, 3).transpose(0, 1).contiguous()
    loss_tm = torch.empty(6).cuda()
    ops.mse_image(dev(recon_tm), dev(img), loss_tm, None, perm=(B_, T_))
    assert_close(loss_tm, loss_ref.detach().reshape(B_, T_).t().reshape(6), what='mse perm')
    assert_close(loss, loss_ref, what="mse")
    assert_close(dr, recon.grad, what="mse grad")
    f = torch.empty(6, 64, 64, 3).cuda()
    ops.image_to_f32(dev(img), f, n_images=6)
    assert torch.equal(f.cpu(), img.float() / 255.0 - 0.5)
    ops.image_to_f32(dev(img), f, n_images=6, perm=(B_, T_))
    assert torch.equal(f.cpu(), (img.float() / 255.0 - 0.5).reshape(B_, T_, 64, 64, 3).transpose(0, 1).reshape(6, 64, 64, 3))
    # symlog mse
    mode = torch.randn(40, 9, generator=g).requires_grad_(True)
    xv = torch.randn(40, 9, generator=g) * 5
    with torch.no_grad():
        mode[0, 0] = O.symlog(xv[0, 0])  # exercises the < 1e-8 zeroing
    ref = -O.symlog_mse_logprob(mode[None], xv[None])[0]
    (ref.sum() * 0.5).backward()
    sl = torch.empty(40).cuda()
    dm = torch.empty(40, 9).cuda()
    ops.symlog_mse(dev(mode.detach()), dev(xv), sl, dm, upstream=0.5)
    assert_close(sl, ref, what="symlog mse")
    assert_close(dm, mode.grad, what="symlog mse grad")
    y = torch.empty(40, 9).cuda()
    ops.symlog(dev(xv), y)
    assert_close(y, O.symlog(xv), what="symlog")


def test_actor_normal_fwd_bwd(ops):
    M, A = 700, 6
    g = torch.Generator().manual_seed(13)
    mr = torch.randn(M, A, generator=g).requires_grad_(True)
    sr = torch.randn(M, A, generator=g).requires_grad_(True)
    eps = torch.randn(M, A, generator=g) * 1.5
    mean, std = torch.tanh(mr), 0.9 * torch.sigmoid(sr + 2.0) + 0.1
    pre = mean + std * eps
    act_ref = pre * (1.0 / torch.clip(pre.abs(), min=1.0)).detach()
    ent_ref = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)).sum(-1)
    fixed = act_ref.detach() + 0.1
    lp_ref = (-((fixed - mean) ** 2) / (2 * std ** 2) - torch.log(std) - math.log(math.sqrt(2 * math.pi))).sum(-1)
    da, de, dl = torch.randn(M, A, generator=g), torch.randn(M, generator=g), torch.randn(M, generator=g)
    ((act_ref * da).sum() + (ent_ref * de).sum() + (lp_ref * dl).sum()).backward()
    mrd, srd = dev(mr.detach()), dev(sr.detach())
    action, ent, lp = torch.empty(M, A).cuda(), torch.empty(M).cuda(), torch.empty(M).cuda()
    ops.actor_normal_fwd(mrd, srd, dev(eps), action, ent)
    ops.actor_normal_logp(mrd, srd, dev(fixed), lp)
    assert_close(action, act_ref, what="action")
    assert_close(ent, ent_ref, what="entropy")
    assert_close(lp, lp_ref, what="logp")
    dm, ds = torch.empty(M, A).cuda(), torch.empty(M, A).cuda()
    ops.actor_normal_bwd(mrd, srd, dm, ds, eps=dev(eps), action=dev(fixed), daction=dev(da), dent=dev(de), dlogp=dev(dl))
    assert_close(dm, mr.grad, what="dmean_raw")
    assert_close(ds, sr.grad, what="dstd_raw")
    # log-prob of the rsampled action itself (policy.log_prob(imag_action), models.py:667): the path through the action
    mr.grad = sr.grad = None
    mean, std = torch.tanh(mr), 0.9 * torch.sigmoid(sr + 2.0) + 0.1
    pre = mean + std * eps
    act = pre * (1.0 / torch.clip(pre.abs(), min=1.0)).detach()
    lp2 = (-((act - mean) ** 2) / (2 * std ** 2) - torch.log(std) - math.log(math.sqrt(2 * math.pi))).sum(-1)
    (lp2 * dl).sum().backward()
    ops.actor_normal_bwd(mrd, srd, dm, ds, eps=dev(eps), action=dev(act.detach()), dlogp=dev(dl), logp_of_sample=True)
    assert_close(dm, mr.grad, what="dmean_raw (log-prob of the sample)")
    assert_close(ds, sr.grad, what="dstd_raw (log-prob of the sample)")


def test_lambda_return_fwd_bwd(ops):
    H, N = 15, 1024
    g = torch.Generator().manual_seed(14)
    reward = torch.randn(H, N, 1, generator=g).requires_grad_(True)
    value = torch.randn(H, N, 1, generator=g)
    cl = (2 * torch.randn(H, N, 1, generator=g)).requires_grad_(True)
    disc = 0.997 * torch.sigmoid(cl)
    tgt_ref = O.lambda_return(reward, value, disc, 0.95)
    w_ref = torch.cumprod(torch.cat([torch.ones_like(disc[:1]), disc[:-1]], 0), 0)
    dt = torch.randn(H - 1, N, 1, generator=g)
    (tgt_ref * dt).sum().backward()
    target, weights, d_out = torch.empty(H - 1, N).cuda(), torch.empty(H, N).cuda(), torch.empty(H, N).cuda()
    rd, vd, cd = dev(reward.detach().squeeze(-1)), dev(value.squeeze(-1)), dev(cl.detach().squeeze(-1))
    ops.lambda_return_fwd(rd, vd, cd, target, weights, d_out, gamma=0.997, lam=0.95)
    assert_close(target, tgt_ref.squeeze(-1), what="target")
    assert_close(weights, w_ref.squeeze(-1), what="weights")
    assert_close(d_out, disc.squeeze(-1), what="disc")
    dr, dc = torch.empty(H, N).cuda(), torch.empty(H, N).cuda()
    ops.lambda_return_bwd(dev(dt.squeeze(-1)), vd, cd, target, dr, dc, gamma=0.997, lam=0.95)
    assert_close(dr, reward.grad.squeeze(-1), what="dreward")
    assert_close(dc, cl.grad.squeeze(-1), what="dcont")


def test_reset_blend_fwd_bwd(ops):
    B, n = 16, 1030
    g = torch.Generator().manual_seed(15)
    x = torch.randn(B, n, generator=g).requires_grad_(True)
    init = torch.randn(n, generator=g).requires_grad_(True)
    first = (torch.rand(B, generator=g) > 0.6).float()
    ref = x * (1 - first[:, None]) + init[None] * first[:, None]
    go = torch.randn(B, n, generator=g)
    ref.backward(go)
    out = torch.empty(B, n).cuda()
    ops.reset_blend(dev(x.detach()), dev(init.detach()), dev(first), out)
    assert_close(out, ref, what="blend")
    dx, di = torch.empty(B, n).cuda(), torch.zeros(n).cuda()
    ops.reset_blend_bwd(dev(go), dev(first), dx, di)
    assert_close(dx, x.grad, what="dx")
    assert_close(di, init.grad, what="dinit")
    # fused observe-step form
    SD, De, A = 64, 40, 6
    ps, pd, ac = torch.randn(B, SD, generator=g), torch.randn(B, De, generator=g), torch.randn(B, A, generator=g)
    s0, d0 = torch.randn(SD, generator=g), torch.randn(De, generator=g)
    os_, od_, oa_ = torch.empty(B, SD).cuda(), torch.empty(B, De).cuda(), torch.empty(B, A).cuda()
    ops.obs_blend(dev(ps), dev(s0), dev(pd), dev(d0), dev(ac), dev(first), os_, od_, oa_)
    m = first[:, None]
    assert_close(os_, ps * (1 - m) + s0 * m, what="blend s")
    assert_close(od_, pd * (1 - m) + d0 * m, what="blend d")
    assert_close(oa_, ac * (1 - m), what="blend a")
    ops.obs_blend(None, dev(s0), None, dev(d0), dev(ac), torch.ones(B).cuda(), os_, od_, oa_)
    assert_close(os_, s0.expand(B, SD), what="blend init")
    gsp, gdp = torch.randn(B, SD, generator=g), torch.randn(B, De, generator=g)
    dsn, ddn = torch.randn(B, SD, generator=g), torch.randn(B, De, generator=g)
    gsd, gdd, a0, b0 = dev(gsp.clone()), dev(gdp.clone()), torch.zeros(SD).cuda(), torch.zeros(De).cuda()
    ops.obs_blend_bwd(dev(dsn), dev(ddn), dev(first), gsd, gdd, a0, b0)
    assert_close(gsd, gsp + dsn * (1 - m), what="carry s")
    assert_close(gdd, gdp + ddn * (1 - m), what="carry d")
    assert_close(a0, (dsn * m).sum(0), what="ds0")
    assert_close(b0, (ddn * m).sum(0), what="dd0")


def test_adam_clip_matches_oracle(ops):
    n = 100003
    g = torch.Generator().manual_seed(16)
    p0 = torch.randn(n, generator=g)
    st = dict(step=0, m=[torch.zeros(n)], v=[torch.zeros(n)])
    p_ref = p0.clone()
    pd = dev(p0.clone())
    m, v = torch.zeros(n).cuda(), torch.zeros(n).cuda()
    state = torch.zeros(4).cuda()
    for it in range(3):
        grad = torch.randn(n, generator=g) * (50.0 if it == 1 else 0.01)
        norm_ref = O.clip_and_adam([p_ref], [grad.clone()], st, lr=1e-2, eps=1e-8, clip=100.0)
        gd = dev(grad)
        ops.sumsq_accumulate(gd, state[1:2])
        ops.adam_step(pd, gd, m, v, state, lr=1e-2, eps=1e-8, clip=100.0)
        assert abs(state[2].item() - norm_ref.item()) <= 1e-4 * norm_ref.item()
        assert_close(pd, p_ref, tol=1e-5, what=f"adam it {it}")
    assert state[0].item() == 3.0 and state[1].item() == 0.0


@pytest.mark.parametrize("n", [1, 5, 4099, 1 << 20, 15687811])
def test_gradient_norm_is_summed_in_a_fixed_order(ops, n):
    """dv3_sumsq_ordered (the clipping norm of tools.py:768 on the flat bucket): equal to the sum of squares, accumulates
    into `out`, and bit-identical from launch to launch -- per-workgroup partial sums added up by index, no atomics --
    so that data-parallel replicas with the same all-reduced gradient take the same step (tests/test_dp_gpu.py)."""
    g = torch.Generator().manual_seed(n % 1000)
    x = dev(torch.randn(n + 4, generator=g))[:n] if n % 4 else dev(torch.randn(n, generator=g))
    x = x.contiguous()
    partial = torch.zeros(1024, device="cuda")
    outs = []
    for _ in range(4):
        out = torch.full((1,), 2.5, device="cuda")
        ops.sumsq_ordered(x, out, partial)
        outs.append(out.clone())
    want = float((x.double() ** 2).sum()) + 2.5
    assert abs(float(outs[0]) - want) <= 2e-6 * want
    assert all(torch.equal(o, outs[0]) for o in outs), [float(o) for o in outs]
    small = torch.zeros(3, device="cuda")  # a short scratch buffer limits the number of workgroups, not the result's value
    out = torch.zeros(1, device="cuda")
    ops.sumsq_ordered(x, out, small)
    assert abs(float(out) - (want - 2.5)) <= 2e-6 * want


# ------------------------------------------------------------------------------------------ conv stacks
CONV_CASES = [(4, 64, 64, 3, 32), (3, 32, 32, 32, 64), (2, 16, 16, 64, 128), (5, 8, 8, 128, 256),
              (3, 64, 64, 3, 2), (3, 32, 32, 2, 4), (2, 8, 8, 8, 16), (1, 4, 4, 6, 5),
              (2, 64, 64, 3, 32), (5, 2, 2, 3, 32), (2, 4, 4, 3, 96), (33, 8, 8, 3, 40),  # image-side wgrad path
              (2, 64, 96, 3, 96), (3, 32, 64, 3, 32),  # ... its tile-walk kernel: several tiles, non-square, width 96
              (2, 16, 16, 96, 192), (3, 8, 8, 64, 160), (1, 4, 4, 32, 72)]  # k-contiguous LDS tile path, crafter widths / ragged Co


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("N,H,W,Ci,Co", CONV_CASES)
def test_conv_s2_fwd_dgrad_wgrad(ops, N, H, W, Ci, Co):
    """Encoder layer: Conv2d(k4,s2,'same') forward, its input gradient (= convT with the same weight)
    and its weight gradient, against torch CPU conv2d autograd."""
    g = torch.Generator().manual_seed(N * H + Ci * Co)
    x = torch.randn(N, Ci, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Co, Ci, 4, 4, generator=g) / math.sqrt(16 * Ci)).requires_grad_(True)
    dy = torch.randn(N, Co, H // 2, W // 2, generator=g)
    y_ref = F.conv2d(F.pad(x, [1, 1, 1, 1]), w, None, 2)
    y_ref.backward(dy)
    xd, wd, dyd = dev(nhwc(x.detach())), dev(w.detach()), dev(nhwc(dy))
    wp = torch.empty(Co, 16 * Ci).cuda()
    ops.pack_conv_weight(wd, wp, transposed=False)
    y = torch.empty(N, H // 2, W // 2, Co).cuda()
    ops.conv_s2_fwd(xd, wp, y, Ci=Ci, Co=Co)
    assert_close(y, nhwc(y_ref), what="conv fwd")
    # dgrad: ConvTranspose2d with the Conv2d weight read as [in=Co][out=Ci]
    wpt = torch.empty(4, Ci, 4 * Co).cuda()
    ops.pack_conv_weight(wd, wpt, transposed=True)
    dx = torch.empty(N, H, W, Ci).cuda()
    ops.convT_s2_fwd(dyd, wpt, dx, Ci=Co, Co=Ci)
    assert_close(dx, nhwc(x.grad), what="conv dgrad")
    dw = torch.zeros(Co, Ci, 4, 4).cuda()
    ops.conv_s2_wgrad(dyd, xd, dw)
    assert_close(dw, w.grad, tol=2e-4, what="conv wgrad")


@pytest.mark.parametrize("N,H,W,Ci,Co", [(4, 4, 4, 256, 128), (3, 8, 8, 128, 64), (2, 16, 16, 64, 32),
                                         (3, 32, 32, 32, 3), (2, 4, 4, 16, 8), (2, 32, 32, 2, 3),
                                         (2, 4, 4, 384, 192), (3, 4, 4, 64, 160), (1, 8, 8, 32, 256),
                                         # the 16 x 16 input-tile kernel (Co <= 32, Ci in 16..64): several tiles per image
                                         # (halo across tile edges), ragged channel counts, non-square
                                         (2, 32, 32, 64, 32), (3, 16, 48, 32, 24), (1, 32, 16, 48, 7), (5, 16, 16, 16, 32)])
def test_convT_s2_fwd_dgrad_wgrad(ops, N, H, W, Ci, Co):
    """Decoder layer: ConvTranspose2d(k4,s2,p1) + bias + 0.5, its input gradient (= conv with the
    same weight) and its weight gradient."""
    g = torch.Generator().manual_seed(N + H + Ci + Co)
    x = torch.randn(N, Ci, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Ci, Co, 4, 4, generator=g) / math.sqrt(4 * Ci)).requires_grad_(True)
    b = torch.randn(Co, generator=g)
    dy = torch.randn(N, Co, 2 * H, 2 * W, generator=g)
    y_ref = F.conv_transpose2d(x, w, b, 2, padding=1) + 0.5
    y_ref.backward(dy)
    xd, wd, dyd = dev(nhwc(x.detach())), dev(w.detach()), dev(nhwc(dy))
    wpt = torch.empty(4, Co, 4 * Ci).cuda()
    ops.pack_conv_weight(wd, wpt, transposed=True)
    y = torch.empty(N, 2 * H, 2 * W, Co).cuda()
    ops.convT_s2_fwd(xd, wpt, y, Ci=Ci, Co=Co, bias=dev(b), out_add=0.5)
    assert_close(y, nhwc(y_ref), what="convT fwd")
    y2 = y.clone()
    ops.convT_s2_fwd(xd, wpt, y2, Ci=Ci, Co=Co, accumulate=True)  # += the plain product
    assert_close(y2, nhwc(2 * y_ref - 0.5 - b.view(1, -1, 1, 1)), tol=2e-4, what="convT fwd accumulate")
    # dgrad: Conv2d over dOut with the ConvTranspose2d weight read as [out=Ci][in=Co]
    wp = torch.empty(Ci, 16 * Co).cuda()
    ops.pack_conv_weight(wd, wp, transposed=False)
    dx = torch.empty(N, H, W, Ci).cuda()
    ops.conv_s2_fwd(dyd, wp, dx, Ci=Co, Co=Ci)
    assert_close(dx, nhwc(x.grad), what="convT dgrad")
    dw = torch.zeros(Ci, Co, 4, 4).cuda()
    ops.conv_s2_wgrad(xd, dyd, dw)
    assert_close(dw, w.grad, tol=2e-4, what="convT wgrad")


def test_conv_full_size_linearity_property(ops):
    """BASELINE cfg-2 size (1024 frames): conv(a*x1 + x2) == a*conv(x1) + conv(x2) -- a
    size-independent check where a CPU reference would take too long."""
    N, H, W, Ci, Co = 1024, 64, 64, 3, 32
    g = torch.Generator().manual_seed(0)
    x1 = torch.randn(N, H, W, Ci, generator=g).cuda()
    x2 = torch.randn(N, H, W, Ci, generator=g).cuda()
    w = dev(torch.randn(Co, Ci, 4, 4, generator=g) / math.sqrt(48))
    wp = torch.empty(Co, 16 * Ci).cuda()
    ops.pack_conv_weight(w, wp, transposed=False)
    y1, y2, y3 = (torch.empty(N, 32, 32, Co).cuda() for _ in range(3))
    ops.conv_s2_fwd(x1, wp, y1, Ci=Ci, Co=Co)
    ops.conv_s2_fwd(x2, wp, y2, Ci=Ci, Co=Co)
    ops.conv_s2_fwd(0.5 * x1 + x2, wp, y3, Ci=Ci, Co=Co)
    assert_close(y3, 0.5 * y1 + y2, tol=1e-5, what="linearity")
    # and the first 2 frames against the CPU reference
    ref = F.conv2d(F.pad(x1[:2].cpu().permute(0, 3, 1, 2), [1, 1, 1, 1]), w.cpu(), None, 2)
    assert_close(y1[:2], nhwc(ref), what="first frames")


def test_layout_helpers(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 7, 11, generator=g)
    y = torch.empty(7, 5, 11).cuda()
    ops.transpose01(dev(x), y)
    assert torch.equal(y.cpu(), x.transpose(0, 1).contiguous())
    m = torch.randn(1000, 255, generator=g)
    out = torch.empty(255).cuda()
    ops.colsum(dev(m), out)
    assert_close(out, m.sum(0), what="colsum")
    ops.colsum(dev(m), out, accumulate=True)
    assert_close(out, 2 * m.sum(0), what="colsum acc")
    narrow = torch.randn(70000, 3, generator=g)
    o3 = torch.empty(3).cuda()
    ops.colsum(dev(narrow), o3)
    assert_close(o3, narrow.sum(0), tol=3e-4, what="colsum narrow")
    one = torch.randn(15360, 1, generator=g)
    o1 = torch.zeros(1).cuda()
    ops.colsum(dev(one), o1, accumulate=True)
    assert_close(o1, one.sum(0), tol=3e-4, what="colsum 1 col")
    small = torch.randn(16, 6, generator=g)
    o2 = torch.empty(6).cuda()
    ops.colsum(dev(small), o2)
    assert_close(o2, small.sum(0), what="colsum small")
    t = torch.randn(1, 512, generator=g).requires_grad_(True)
    yt = torch.tanh(t)
    gy = torch.randn(1, 512, generator=g)
    yt.backward(gy)
    yd = torch.empty(1, 512).cuda()
    ops.tanh_fwd(dev(t.detach()), yd)
    assert_close(yd, yt, what="tanh")
    dxd = torch.empty(1, 512).cuda()
    ops.tanh_bwd(yd, dev(gy), dxd)
    assert_close(dxd, t.grad, what="tanh bwd")


@pytest.mark.parametrize("CW", [32, 96])
def test_three_channel_image_layers(ops, CW):
    """Dedicated kernels for the 3-channel side: encoder conv 3->CW and decoder convT CW->3 (+bias+0.5),
    weights in the reference layout, against torch CPU."""
    g = torch.Generator().manual_seed(CW)
    N, H = 3, 64
    x = torch.randn(N, 3, H, H, generator=g)
    w = torch.randn(CW, 3, 4, 4, generator=g) / math.sqrt(48)
    ref = F.conv2d(F.pad(x, [1, 1, 1, 1]), w, None, 2)
    y = torch.empty(N, H // 2, H // 2, CW).cuda()
    ops.conv_s2_c3_fwd(dev(nhwc(x)), dev(w), y, CW=CW)
    assert_close(y, nhwc(ref), what="conv c3")
    ops.conv_s2_c3_fwd(dev(nhwc(x)), dev(w), y, CW=CW, accumulate=True)
    assert_close(y, 2 * nhwc(ref), what="conv c3 accumulate")
    xi = torch.randn(N, CW, 32, 32, generator=g)
    wt = torch.randn(CW, 3, 4, 4, generator=g) / math.sqrt(4 * CW)
    b = torch.randn(3, generator=g)
    reft = F.conv_transpose2d(xi, wt, b, 2, padding=1) + 0.5
    yt = torch.empty(N, 64, 64, 3).cuda()
    ops.convT_s2_c3_fwd(dev(nhwc(xi)), dev(wt), yt, CW=CW, bias=dev(b), out_add=0.5)
    assert_close(yt, nhwc(reft), what="convT c3")
    # adjointness: conv_c3 with the ConvTranspose2d weight is the convT's input gradient
    xi2 = xi.clone().requires_grad_(True)
    dy = torch.randn(N, 3, 64, 64, generator=g)
    F.conv_transpose2d(xi2, wt, None, 2, padding=1).backward(dy)
    dx = torch.empty(N, 32, 32, CW).cuda()
    ops.conv_s2_c3_fwd(dev(nhwc(dy)), dev(wt), dx, CW=CW)
    assert_close(dx, nhwc(xi2.grad), what="convT c3 dgrad")


def test_fused_next_step_reset_blend_outputs(ops):
    """gru_fwd / onehot_sample with next_blend: the second output equals a separate reset blend of the first."""
    g = torch.Generator().manual_seed(5)
    M, De, S, D = 16, 512, 32, 32
    p = torch.randn(M, 3 * De, generator=g)
    h = torch.randn(M, De, generator=g)
    gamma, beta = torch.rand(3 * De, generator=g) + 0.5, torch.randn(3 * De, generator=g) * 0.1
    first = (torch.rand(M, generator=g) < 0.3).float()
    init_d = torch.randn(De, generator=g)
    hn, mean, rstd = torch.empty(M, De).cuda(), torch.empty(M).cuda(), torch.empty(M).cuda()
    nxt = torch.full((M, De + 8), float("nan")).cuda()  # row-strided destination
    ops.gru_fwd(dev(p), dev(gamma), dev(beta), dev(h), hn, mean, rstd, next_blend=(dev(first), dev(init_d), nxt[:, :De]))
    hn2 = torch.empty(M, De).cuda()
    ops.gru_fwd(dev(p), dev(gamma), dev(beta), dev(h), hn2, mean, rstd)
    assert torch.equal(hn, hn2)
    want = hn.cpu() * (1 - first[:, None]) + init_d[None] * first[:, None]
    assert_close(nxt[:, :De], want, what="gru next blend")
    logit = torch.randn(M, S, D, generator=g)
    q = torch.empty(M, S, D).exponential_(1.0, generator=g).clamp_min(1e-20)
    init_s = torch.randn(S * D, generator=g)
    st, nxt_s = torch.empty(M, S, D).cuda(), torch.empty(M, S, D).cuda()
    ops.onehot_sample(dev(logit), st, noise=dev(q), next_blend=(dev(first), dev(init_s), nxt_s))
    st2 = torch.empty(M, S, D).cuda()
    ops.onehot_sample(dev(logit), st2, noise=dev(q))
    assert torch.equal(st, st2)
    want = st.cpu() * (1 - first[:, None, None]) + init_s.view(1, S, D) * first[:, None, None]
    assert_close(nxt_s, want, what="sample next blend")
    # class indices of the blended state (the one-hot initial state's index where the next step resets)
    init_i = torch.randint(0, D, (S,), generator=g, dtype=torch.int32)
    init_oh = F.one_hot(init_i.long(), D).float().reshape(-1)
    idx, nidx = torch.empty(M * S, dtype=torch.int32, device="cuda"), torch.empty(M * S, dtype=torch.int32, device="cuda")
    ops.onehot_sample(dev(logit), st, noise=dev(q), idx=idx, next_blend=(dev(first), dev(init_oh), nxt_s, dev(init_i), nidx))
    assert torch.equal(idx.cpu().view(M, S), st.cpu().argmax(-1).int())
    assert torch.equal(nidx.cpu().view(M, S), nxt_s.cpu().argmax(-1).int())
    want_i = torch.where(first[:, None] > 0, init_i[None].expand(M, S), st.cpu().argmax(-1).int())
    assert torch.equal(nidx.cpu().view(M, S), want_i)


# ----------------------------------------------------------------------------- row-fused imagination layers
@pytest.mark.parametrize("M,N,S,D,A2,with_base", [(1024, 512, 32, 32, 6, False), (1024, 512, 32, 32, 0, True),
                                                   (16, 512, 32, 32, 6, False), (33, 1024, 32, 32, 17, True),
                                                   (7, 16, 4, 4, 3, False), (5, 40, 3, 5, 0, True),
                                                   (300, 256, 32, 32, 18, False), (64, 768, 32, 32, 6, True)])
def test_onehot_linear_ln_matches_dense_linear(ops, M, N, S, D, A2, with_base):
    """Gather of S weight columns per row == the dense Linear on the one-hot expansion (networks.py:216-218,
    657-668), with the LayerNorm + SiLU behind it; also the pre-activation-only form."""
    g = torch.Generator().manual_seed(M * 31 + N)
    K = S * D + A2
    W = torch.randn(N, K, generator=g) / math.sqrt(S + A2 + 1)  # the Linear's [out, in] weight
    idx = torch.randint(0, D, (M, S), generator=g, dtype=torch.int32)
    x2 = torch.randn(M, A2, generator=g) if A2 else None
    base = torch.randn(M, N, generator=g) if with_base else None
    gamma, beta = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    onehot = F.one_hot(idx.long(), D).float().reshape(M, S * D)
    x = onehot if x2 is None else torch.cat([onehot, x2], -1)
    pre_ref = x @ W.t() + (base if base is not None else 0.0)
    y_ref = F.silu(O.layer_norm(pre_ref, gamma, beta))
    WT = torch.empty(K, N, device="cuda")
    ops.transpose2d(dev(W), WT)
    assert torch.equal(WT.cpu(), W.t().contiguous())
    pre = dev(base.clone()) if base is not None else torch.empty(M, N, device="cuda")
    y, mean, rstd = torch.empty(M, N, device="cuda"), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.onehot_linear_ln(dev(idx), D, WT, pre, x2=None if x2 is None else dev(x2), base=pre if base is not None else None,
                         gamma=dev(gamma), beta=dev(beta), y=y, mean=mean, rstd=rstd)
    assert_close(pre, pre_ref, what="pre")
    assert_close(y, y_ref, what="y")
    assert_close(mean, pre_ref.mean(-1), what="mean")
    assert_close(rstd, 1.0 / torch.sqrt(pre_ref.var(-1, unbiased=False) + 1e-3), what="rstd")
    pre2 = torch.empty(M, N, device="cuda")
    ops.onehot_linear_ln(dev(idx), D, WT, pre2, x2=None if x2 is None else dev(x2),
                         base=dev(base) if base is not None else None)
    assert_close(pre2, pre_ref, what="pre only")
    # the class indices of a one-hot tensor
    back = torch.empty(M * S, dtype=torch.int32, device="cuda")
    ops.onehot_to_idx(dev(F.one_hot(idx.long(), D).float()), back)
    assert torch.equal(back.cpu().view(M, S), idx)


@pytest.mark.parametrize("M,U,A", [(1024, 512, 6), (16, 512, 6), (37, 16, 3), (100, 1024, 17), (9, 200, 1)])
def test_actor_head_continuous(ops, M, U, A):
    """LN + SiLU + mean/std heads + rsample + absmax rescale + entropy in one launch == the separate steps
    (networks.py:672-681, 693-700, tools.py:594-598)."""
    g = torch.Generator().manual_seed(U + A)
    pre = torch.randn(M, U, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(U, generator=g), 0.1 * torch.randn(U, generator=g)
    Wm, Ws = torch.randn(A, U, generator=g) / math.sqrt(U), torch.randn(A, U, generator=g) / math.sqrt(U)
    bm, bs = 0.1 * torch.randn(A, generator=g), 0.1 * torch.randn(A, generator=g)
    eps = torch.randn(M, A, generator=g)
    y_ref = F.silu(O.layer_norm(pre, gamma, beta))
    mr, sr = y_ref @ Wm.t() + bm, y_ref @ Ws.t() + bs
    mu, sd = torch.tanh(mr), 0.9 * torch.sigmoid(sr + 2.0) + 0.1
    a = mu + sd * eps
    a_ref = a / torch.clip(a.abs(), min=1.0)
    ent_ref = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(sd)).sum(-1)
    mk = lambda *s: torch.empty(*s, device="cuda")
    y, mean, rstd, om, os_, act, ent, eps_out = mk(M, U), mk(M), mk(M), mk(M, A), mk(M, A), mk(M, A), mk(M), mk(M, A)
    ops.actor_head(dev(pre), dev(gamma), dev(beta), y, mean, rstd, dev(Wm), dev(bm), dev(Ws), dev(bs), om, os_, act, ent,
                   noise=dev(eps), eps_out=eps_out, min_std=0.1, max_std=1.0)
    assert_close(y, y_ref, what="y")
    assert_close(om, mr, what="mean head"), assert_close(os_, sr, what="std head")
    assert_close(act, a_ref, what="action"), assert_close(ent, ent_ref, what="entropy")
    assert torch.equal(eps_out.cpu(), eps)
    # without injected noise: the N(0,1) draws are those of dv3_fill_normal at the same stream position
    r1, r2 = ops.RngStream("cuda", seed=11), ops.RngStream("cuda", seed=11)
    ops.actor_head(dev(pre), dev(gamma), dev(beta), y, mean, rstd, dev(Wm), dev(bm), dev(Ws), dev(bs), om, os_, act, ent,
                   rng=r1, eps_out=eps_out, min_std=0.1, max_std=1.0)
    want = ops.fill_normal(mk(M, A), r2)
    assert torch.allclose(eps_out, want, rtol=1e-5, atol=1e-6) and r1.cursor == r2.cursor
    a2 = mu.cuda() + sd.cuda() * eps_out
    assert_close(act, a2 / torch.clip(a2.abs(), min=1.0), what="action (philox)")


@pytest.mark.parametrize("M,U,A", [(2048, 512, 18), (16, 512, 18), (21, 16, 5), (50, 1024, 64)])
def test_actor_head_onehot(ops, M, U, A):
    """One-hot actor (networks.py:713-714): logits head + OneHotDist sample / entropy; teacher forcing counts flips."""
    g = torch.Generator().manual_seed(U * 3 + A)
    pre = torch.randn(M, U, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(U, generator=g), 0.1 * torch.randn(U, generator=g)
    Wm, bm = torch.randn(A, U, generator=g) / math.sqrt(U) * 3, 0.1 * torch.randn(A, generator=g)
    q = torch.empty(M, A).exponential_(1.0, generator=g).clamp_min(1e-20)
    y_ref = F.silu(O.layer_norm(pre, gamma, beta))
    lg = y_ref @ Wm.t() + bm
    samp_ref = O.onehot_sample(lg, q, 0.01)
    ent_ref = O.onehot_entropy(lg[:, None, :], 0.01)
    mk = lambda *s: torch.empty(*s, device="cuda")
    y, mean, rstd, om, act, ent = mk(M, U), mk(M), mk(M), mk(M, A), mk(M, A), mk(M)
    ai = torch.empty(M, dtype=torch.int32, device="cuda")
    ops.actor_head(dev(pre), dev(gamma), dev(beta), y, mean, rstd, dev(Wm), dev(bm), None, None, om, None, act, ent,
                   noise=dev(q), act_idx=ai, unimix=0.01, onehot=True)
    assert_close(om, lg, what="logits"), assert_close(ent, ent_ref, what="entropy")
    same = (act.cpu() == samp_ref).all(-1)
    assert same.float().mean() >= 1 - 2e-3, f"{(~same).sum()} of {M} draws differ"
    assert torch.equal(ai.cpu().long(), act.cpu().argmax(-1)) and act.sum(-1).eq(1).all()
    # teacher forcing: emit the given classes, count the rows whose own draw differs
    forced = torch.randint(0, A, (M,), generator=g, dtype=torch.int32)
    flips = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.actor_head(dev(pre), dev(gamma), dev(beta), y, mean, rstd, dev(Wm), dev(bm), None, None, om, None, act, ent,
                   noise=dev(q), act_idx=ai, forced=dev(forced), flips=flips, unimix=0.01, onehot=True)
    assert torch.equal(act.cpu().argmax(-1).int(), forced)
    assert abs(int(flips.item()) - int((samp_ref.argmax(-1).int() != forced).sum())) <= int((~same).sum())


def test_onehot_sample_teacher_forcing_counts_flips(ops):
    R, D = 4096, 32
    g = torch.Generator().manual_seed(3)
    logit = torch.randn(R, D, generator=g)
    q = torch.empty(R, D).exponential_(1.0, generator=g).clamp_min(1e-20)
    own = O.onehot_sample(logit, q, 0.01).argmax(-1).int()
    forced = own.clone()
    forced[::7] = (forced[::7] + 1) % D
    out, idx = torch.empty(R, D, device="cuda"), torch.empty(R, dtype=torch.int32, device="cuda")
    flips = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.onehot_sample(dev(logit), out, noise=dev(q), idx=idx, forced=dev(forced), flips=flips, unimix=0.01)
    assert torch.equal(idx.cpu(), forced) and torch.equal(out.cpu().argmax(-1).int(), forced)
    assert abs(int(flips.item()) - len(range(0, R, 7))) <= 2  # + the rare ulp-level disagreement with the oracle


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 512), (100, 128, 48), (33, 64, 512), (2048, 1024, 512)])
@pytest.mark.parametrize("mode", [False, True])
def test_gemm_with_sampling_epilogue_equals_gemm_then_sample(ops, M, N, K, mode):
    """dv3_gemm_sample_f32 == dv3_gemm_f32 (the tile it picks for the size: register-direct for the small shapes, the
    k-contiguous LDS tile for 1024 x 1024 and up) followed by dv3_onehot_sample_fwd: bit-equal logits and samples,
    with injected noise, with the Philox stream, and teacher-forced."""
    g = torch.Generator().manual_seed(M + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K) * 3, torch.randn(N, generator=g)
    q = torch.empty(M, N // 32, 32).exponential_(1.0, generator=g).clamp_min(1e-20)
    A, W, b, q = dev(A), dev(W), dev(b), dev(q)
    lg0, st0 = torch.empty(M, N, device="cuda"), torch.empty(M, N // 32, 32, device="cuda")
    i0 = torch.empty(M * N // 32, dtype=torch.int32, device="cuda")
    ops.gemm(A, W, lg0, bias=b, tile=ops.pick_gemm_tile(M, N))
    ops.onehot_sample(lg0.view(M, N // 32, 32), st0, noise=None if mode else q, idx=i0, mode=mode)
    lg1, st1 = torch.empty_like(lg0), torch.empty_like(st0)
    i1 = torch.empty_like(i0)
    ops.gemm_sample(A, W, lg1, st1, bias=b, noise=None if mode else q, idx=i1, mode=mode)
    assert torch.equal(lg0, lg1) and torch.equal(st0, st1) and torch.equal(i0, i1)
    assert_close(lg1, A.cpu() @ W.cpu().t() + b.cpu(), tol=2e-4, what="logits")
    if not mode:
        r0, r1 = ops.RngStream("cuda", seed=5), ops.RngStream("cuda", seed=5)
        ops.onehot_sample(lg0.view(M, N // 32, 32), st0, rng=r0)
        ops.gemm_sample(A, W, lg1, st1, bias=b, rng=r1)
        assert torch.equal(st0, st1) and r0.cursor == r1.cursor
        forced = torch.randint(0, 32, (M * N // 32,), dtype=torch.int32, generator=g).cuda()
        flips = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.gemm_sample(A, W, lg1, st1, bias=b, noise=q, idx=i1, forced=forced, flips=flips)
        assert torch.equal(i1, forced) and torch.equal(st1.argmax(-1).int().view(-1), forced)
        assert int(flips.item()) == int((i0 != forced).sum())


@pytest.mark.parametrize("N,H,Ci,Co", [(1, 64, 3, 32), (1, 32, 32, 64), (2, 8, 128, 256), (16, 16, 64, 128), (3, 4, 5, 7)])
def test_im2col_then_gemm_is_the_same_pad_stride2_conv(ops, N, H, Ci, Co):
    """Few-image encoder path (acting step): dv3_im2col_s2 + dv3_gemm_f32 against the weight as stored ==
    Conv2dSamePad k4 s2 (networks.py:771-798: pad 1 before, 1 after for even sizes)."""
    g = torch.Generator().manual_seed(N + H + Ci + Co)
    x = torch.randn(N, H, H, Ci, generator=g)
    w = torch.randn(Co, Ci, 4, 4, generator=g) / math.sqrt(16 * Ci)
    OH = H // 2
    cols = torch.empty(N * OH * OH, 16 * Ci, device="cuda")
    ops.im2col_s2(dev(x), cols)
    ref_cols = F.unfold(F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1)), kernel_size=4, stride=2)  # [N, Ci*16, OH*OH]
    assert torch.equal(cols.cpu(), ref_cols.transpose(1, 2).reshape(N * OH * OH, 16 * Ci))
    y = torch.empty(N * OH * OH, Co, device="cuda")
    ops.gemm(cols, dev(w).view(Co, 16 * Ci), y)
    ref = F.conv2d(F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1)), w, stride=2).permute(0, 2, 3, 1)
    assert_close(y.view(N, OH, OH, Co), ref, tol=2e-5, what="conv via im2col")


@pytest.mark.parametrize("M,S,N,A", [(16, 32, 512, 6), (5, 7, 256, 0), (32, 32, 1024, 18)])
@pytest.mark.parametrize("use_rng", [False, True])
def test_sample_plus_img_in_launch_equals_sample_then_gather(ops, M, S, N, A, use_rng):
    """Observe scan: dv3_onehot_sample_linear_ln_fwd == dv3_onehot_sample_fwd_ex (with the reset blend of the next
    step) followed by dv3_onehot_linear_ln_fwd, bit for bit (samples, indices, blended state, LayerNorm output)."""
    g = torch.Generator().manual_seed(M + S + N)
    D = 32
    logit = dev(torch.randn(M, S, D, generator=g) * 2)
    q = dev(torch.empty(M, S, D).exponential_(1.0, generator=g).clamp_min(1e-20))
    nf = dev((torch.rand(M, generator=g) < 0.3).float())
    init_idx = torch.randint(0, D, (S,), generator=g)
    init = dev(F.one_hot(init_idx, D).float().view(-1))
    init_idx = init_idx.int().cuda()
    WT = dev(torch.randn(S * D + A, N, generator=g) / math.sqrt(S))
    x2 = dev(torch.randn(M, A, generator=g)) if A else None
    gamma, beta = dev(1 + 0.1 * torch.randn(N, generator=g)), dev(0.1 * torch.randn(N, generator=g))
    forced = torch.randint(0, D, (M * S,), dtype=torch.int32, generator=g).cuda()
    res = []
    for fused in (False, True):
        for force in (False, True):
            rng = ops.RngStream("cuda", seed=11)
            out, nout = torch.empty(M, S, D, device="cuda"), torch.empty(M, S, D, device="cuda")
            idx = torch.empty(M * S, dtype=torch.int32, device="cuda")
            nidx = torch.empty(M, S, dtype=torch.int32, device="cuda")
            pre, y = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
            mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
            flips = torch.zeros(1, dtype=torch.int32, device="cuda")
            kw = dict(noise=None if use_rng else q, rng=rng if use_rng else None, idx=idx,
                      forced=forced if force else None, flips=flips if force else None)
            if fused:
                ops.onehot_sample_linear_ln(logit, out, next_first=nf, init=init, init_idx=init_idx, next_out=nout,
                                            next_idx=nidx.view(-1), WT=WT, x2=x2, pre=pre, gamma=gamma, beta=beta, y=y,
                                            mean=mean, rstd=rstd, **kw)
            else:
                ops.onehot_sample(logit, out, next_blend=(nf, init, nout, init_idx, nidx.view(-1)), **kw)
                ops.onehot_linear_ln(nidx, D, WT, pre, x2=x2, gamma=gamma, beta=beta, y=y, mean=mean, rstd=rstd)
            res.append((out, nout, idx, nidx, pre, y, mean, rstd, flips, torch.tensor(rng.cursor)))
    for a, b in ((res[0], res[2]), (res[1], res[3])):
        for x, z in zip(a, b):
            assert torch.equal(x.cpu(), z.cpu())
    assert torch.equal(res[3][2].cpu(), forced.cpu())


@pytest.mark.parametrize("B,S,D,De", [(16, 32, 32, 512), (5, 3, 7, 20), (32, 32, 32, 1024)])
def test_obs_carry_st_bwd_equals_blend_bwd_then_st_bwd(ops, B, S, D, De):
    """Reverse observe scan: the fused carry + straight-through launch == dv3_obs_blend_bwd followed by
    dv3_onehot_st_bwd(accumulate) on the previous step."""
    g = torch.Generator().manual_seed(B + S + De)
    SD = S * D
    wide = dev(torch.randn(B, SD + De + 4, generator=g))
    dsin, ddin = wide[:, :SD], wide[:, SD + 4:]
    first = dev((torch.rand(B, generator=g) < 0.3).float())
    logit = dev(torch.randn(B, S, D, generator=g))
    gs0, gd0 = torch.randn(B, SD, generator=g), torch.randn(B, De, generator=g)
    dl0 = torch.randn(B, S, D, generator=g)
    res = []
    for fused in (False, True):
        gs, gd, dl = dev(gs0.clone()), dev(gd0.clone()), dev(dl0.clone())
        ds0, dd0 = torch.zeros(SD, device="cuda"), torch.zeros(De, device="cuda")
        if fused:
            ops.obs_carry_st_bwd(dsin, ddin, first, gs, gd, ds0, dd0, logit, dl, unimix=0.01)
        else:
            ops.obs_blend_bwd(dsin, ddin, first, gs, gd, ds0, dd0)
            ops.onehot_st_bwd(logit, gs.view(B, S, D), dl, unimix=0.01, accumulate=True)
        res.append((gs, gd, dl, ds0, dd0))
    for a, b, nm in zip(res[0], res[1], ("gs", "gd", "dlogit", "dstoch0", "ddeter0")):
        assert_close(b, a, tol=1e-6, what=nm)


@pytest.mark.parametrize("n", [15360, 5, 2, 1025, 300000])
def test_tensorstats_kernel_matches_torch_reductions(ops, n):
    """tools.tensorstats (tools.py:949-958): mean / unbiased std / min / max in one launch, optionally of
    (x - shift) / scale (models.py:412-414)."""
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g) * 3 + 1.5
    out = torch.empty(4, device="cuda")
    ops.tensorstats(dev(x), out)
    ref = torch.stack([x.mean(), x.std(), x.min(), x.max()])
    assert_close(out, ref, tol=2e-6, what="stats")
    sh, sc = torch.tensor([0.7]), torch.tensor([2.5])
    ops.tensorstats(dev(x), out, shift=dev(sh), scale=dev(sc))
    y = (x - sh) / sc
    assert_close(out, torch.stack([y.mean(), y.std(), y.min(), y.max()]), tol=2e-6, what="shifted stats")


def test_tensorstats_multi_matches_single(ops):
    """Six logged tensors of different sizes (one with shift / scale) in one launch == six single launches."""
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(n, generator=g) * (i + 1) + i for i, n in enumerate((14336, 15360, 7, 1024, 90000, 33))]
    sh, sc = dev(torch.tensor([0.7])), dev(torch.tensor([2.5]))
    jobs = [(dev(x), sh if i == 4 else None, sc if i == 4 else None) for i, x in enumerate(xs)]
    out = torch.empty(6, 4, device="cuda")
    ops.tensorstats_multi(jobs, out)
    for i, (x, s_, c_) in enumerate(jobs):
        one = torch.empty(4, device="cuda")
        ops.tensorstats(x, one, shift=s_, scale=c_)
        assert torch.equal(out[i], one), i
    with pytest.raises(ValueError):
        ops.tensorstats_multi(jobs + jobs[:1], torch.empty(7, 4, device="cuda"))


@pytest.mark.parametrize("n", [14336, 7, 1, 2, 21, 1000, 20001, 28672, 458752])
def test_quantile_ema_matches_torch_quantile(ops, n):
    """models.RewardEMA (models.py:11-26): exact radix-selected 5 % / 95 % quantiles + EMA == torch.quantile + axpby.
    Ranks and the interpolation weight are computed in float32 exactly as torch.quantile does (n = 21, 20001: q (n - 1)
    is an integer in exact arithmetic, the case where a float64 rank would pick another neighbour), so the SAME two
    order statistics are interpolated with the same weight: the float32 result equals torch.quantile's to the last
    ulp of the interpolation (whether the host fuses the multiply-add is the only freedom left)."""
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g) * 3 + 0.5
    if n > 10:
        x[::5] = x[0]  # ties
        x[1] = -0.0
        x[2] = 0.0
    ref = torch.quantile(x.double(), torch.tensor([0.05, 0.95], dtype=torch.float64)).float()
    ref32 = torch.quantile(x, torch.tensor([0.05, 0.95]))
    out = torch.empty(2, device="cuda")
    ops.quantile2_ema(dev(x), 0.05, 0.95, out_q=out)
    ulp = lambda a, b: ((a.cpu() - b).abs() <= 1.2e-7 * b.abs().clamp_min(1.0)).all()
    assert ulp(out, ref32), (out.cpu(), ref32)
    assert_close(out, ref, tol=1e-6, what="quantiles")
    # concentrated values (what imagined returns look like): every element in one bucket for the first two passes
    y = 1.0 + 1e-3 * torch.randn(n, generator=g)
    ops.quantile2_ema(dev(y), 0.05, 0.95, out_q=out)
    assert ulp(out, torch.quantile(y, torch.tensor([0.05, 0.95])))
    # a NaN among the values makes both quantiles (and the EMA) NaN, as torch.quantile does
    if n > 1:
        z = x.clone()
        z[n // 2] = float("nan")
        ema_nan = torch.tensor([0.3, 2.0], device="cuda")
        ops.quantile2_ema(dev(z), 0.05, 0.95, out_q=out, ema=ema_nan, alpha=0.01)
        assert torch.isnan(out).all() and torch.isnan(ema_nan).all()
        assert torch.isnan(torch.quantile(z, torch.tensor([0.05, 0.95]))).all()
    ema = torch.tensor([0.3, 2.0], device="cuda")
    ops.quantile2_ema(dev(x), 0.05, 0.95, ema=ema, alpha=0.01)
    assert_close(ema, 0.01 * ref + 0.99 * torch.tensor([0.3, 2.0]), tol=1e-6, what="ema")


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 512), (100, 128, 48), (40, 64, 1024)])
def test_gemm_sample_with_layernorm_on_load(ops, M, N, K):
    """ln=...: SiLU(LN(A)) applied while the operand is loaded == dv3_ln_act_fwd followed by the plain fused GEMM;
    the A operand may be a column slice of a wider buffer (the stacked [x2pre | actor pre0] GEMM output)."""
    g = torch.Generator().manual_seed(M + K)
    wide = torch.randn(M, K + 32, generator=g) * 2 + 0.3
    W, b = torch.randn(N, K, generator=g) / math.sqrt(K) * 3, torch.randn(N, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    q = torch.empty(M, N // 32, 32).exponential_(1.0, generator=g).clamp_min(1e-20)
    wided = dev(wide)
    A = wided[:, :K]
    y, mean, rstd = torch.empty(M, K, device="cuda"), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.ln_act_fwd(A, dev(gamma), dev(beta), y, mean, rstd, act=True)
    lg0, st0 = torch.empty(M, N, device="cuda"), torch.empty(M, N // 32, 32, device="cuda")
    ops.gemm_sample(y, dev(W), lg0, st0, bias=dev(b), noise=dev(q))
    lg1, st1 = torch.empty_like(lg0), torch.empty_like(st0)
    m1, r1 = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.gemm_sample(A, dev(W), lg1, st1, bias=dev(b), noise=dev(q), ln=(dev(gamma), dev(beta), m1, r1))
    assert_close(m1, mean, tol=1e-6, what="mean"), assert_close(r1, rstd, tol=1e-6, what="rstd")
    assert_close(lg1, lg0, tol=2e-6, what="logits")
    ref = F.silu(O.layer_norm(wide[:, :K], gamma, beta)) @ W.t() + b
    assert_close(lg1, ref, tol=2e-4, what="logits vs torch")
    assert (st1 != st0).any(-1).float().mean() <= 1e-3



# ------------------------------------------------------------------------------------------ observe-scan fused launches
@pytest.mark.parametrize("M,K,N", [(16, 512, 1024), (32, 1024, 1024), (7, 256, 48), (64, 512, 256), (1, 1024, 1024)])
def test_scan_ln_gemm_equals_two_launches(ops, M, K, N):
    """dv3_scan_ln_gemm_fwd == dv3_ln_act_fwd followed by the few-row GEMM with bias (bit-equal), and against the
    plain math: SiLU(LN(x)) W^T + b (networks.py:197-200)."""
    g = torch.Generator().manual_seed(M + K + N)
    x = dev(torch.randn(M, K, generator=g) * 2 + 0.3)
    gam, bet = dev(1 + 0.1 * torch.randn(K, generator=g)), dev(0.1 * torch.randn(K, generator=g))
    W, bias = dev(torch.randn(N, K, generator=g) / math.sqrt(K)), dev(0.1 * torch.randn(N, generator=g))
    mk = lambda *s: torch.full(s, float("nan"), device="cuda")
    a = dict(y=mk(M, K), mean=mk(M), rstd=mk(M), C=mk(M, N))
    b = dict(y=mk(M, K), mean=mk(M), rstd=mk(M), C=mk(M, N))
    ops.ln_act_fwd(x, gam, bet, a["y"], a["mean"], a["rstd"], act=True)
    ops.gemm(a["y"], W, a["C"], bias=bias)
    ops.scan_ln_gemm(x, gam, bet, b["y"], b["mean"], b["rstd"], W, b["C"], bias=bias)
    for k in a:
        assert torch.equal(a[k], b[k]), (k, (a[k] - b[k]).abs().max().item())
    ref_y = F.silu(O.layer_norm(x.cpu(), gam.cpu(), bet.cpu()))
    assert_close(b["y"], ref_y, what="y")
    assert_close(b["C"], ref_y @ W.cpu().T + bias.cpu(), tol=TOL * math.sqrt(K / 64), what="C")
    # without the saved outputs, accumulating
    C2 = a["C"].clone()
    ops.scan_ln_gemm(x, gam, bet, None, None, None, W, C2, accumulate=True)
    assert_close(C2, 2 * a["C"].cpu() - bias.cpu(), tol=TOL * math.sqrt(K / 64), what="accumulate")


@pytest.mark.parametrize("M,K,N,pad", [(16, 512, 512, 4096), (16, 512, 1024, 6), (32, 1024, 1024, 18), (7, 256, 64, 0),
                                         (64, 512, 512, 8)])
def test_scan_lnbwd_gemm_equals_two_launches(ops, M, K, N, pad):
    """dv3_scan_lnbwd_gemm == dv3_ln_act_bwd followed by the atomically accumulating few-row data-gradient GEMM:
    dx and the LayerNorm parameter gradients to rounding, C += dx W (networks.py:195-197, 216-218 reversed)."""
    g = torch.Generator().manual_seed(M + K + N)
    x = dev(torch.randn(M, K, generator=g) * 1.5 + 0.2)
    dy = dev(torch.randn(M, K + 64, generator=g))[:, :K]  # a row-strided view, as the GRU matmul's [dx1 | ddin] is
    gam, bet = dev(1 + 0.1 * torch.randn(K, generator=g)), dev(0.1 * torch.randn(K, generator=g))
    W = dev(torch.randn(K, N + pad, generator=g) / math.sqrt(K))[:, :N]  # column slice of a wider weight
    C0 = dev(torch.randn(M, N, generator=g))
    y, mean, rstd = torch.empty(M, K).cuda(), torch.empty(M).cuda(), torch.empty(M).cuda()
    ops.ln_act_fwd(x, gam, bet, y, mean, rstd, act=True)
    a = dict(dx=torch.empty(M, K).cuda(), dg=torch.ones(K).cuda(), db=torch.ones(K).cuda(), C=C0.clone())
    b = dict(dx=torch.empty(M, K).cuda(), dg=torch.ones(K).cuda(), db=torch.ones(K).cuda(), C=C0.clone())
    ops.ln_act_bwd(dy, x, gam, bet, mean, rstd, a["dx"], a["dg"], a["db"], act=True)
    ops.gemm(a["dx"], W, a["C"], transB=False, accumulate="atomic")
    xhat, jac = torch.empty(M, K).cuda(), torch.empty(M, K).cuda()
    ops.scan_ln_factors(x, gam, bet, mean, rstd, xhat, jac)
    ops.scan_lnbwd_gemm(dy, xhat, jac, gam, rstd, b["dx"], W, b["C"], b["dg"], b["db"])
    assert_close(b["dx"], a["dx"].cpu(), tol=2e-6, what="dx")
    assert_close(b["dg"], a["dg"].cpu(), tol=1e-5, what="dgamma"), assert_close(b["db"], a["db"].cpu(), tol=1e-5, what="dbeta")
    assert_close(b["C"], a["C"].cpu(), tol=1e-5 * math.sqrt(K / 64), what="C")
    # and against autograd on the plain math
    xr = x.cpu().clone().requires_grad_(True)
    gr, br = gam.cpu().clone().requires_grad_(True), bet.cpu().clone().requires_grad_(True)
    F.silu(O.layer_norm(xr, gr, br)).backward(dy.cpu())
    assert_close(b["dx"], xr.grad, tol=2e-5, what="dx vs autograd")
    assert_close(b["dg"] - 1, gr.grad, tol=2e-4, what="dgamma vs autograd")
    assert_close(b["C"], C0.cpu() + xr.grad @ W.cpu(), tol=TOL * math.sqrt(K / 64), what="C vs autograd")
    # without parameter gradients
    C2 = C0.clone()
    ops.scan_lnbwd_gemm(dy, xhat, jac, gam, rstd, b["dx"], W, C2)
    assert_close(C2, a["C"].cpu(), tol=1e-5 * math.sqrt(K / 64), what="C (no dgamma)")


@pytest.mark.parametrize("B,S,De,N", [(16, 32, 512, 512), (32, 32, 1024, 512), (5, 8, 64, 64), (64, 32, 512, 1024)])
@pytest.mark.parametrize("with_carry", [True, False])
def test_scan_carry_st_gemm_equals_two_launches(ops, B, S, De, N, with_carry):
    """dv3_scan_carry_st_gemm == dv3_obs_carry_st_bwd (resp. dv3_onehot_st_bwd at the scan's last step) followed by the
    atomically accumulating few-row GEMM dx3 += dlogit W_obs."""
    D = 32
    SD = S * D
    g = torch.Generator().manual_seed(B + S + De + N)
    logit = dev(2 * torch.randn(B, S, D, generator=g))
    dlogit = dev(torch.randn(B, S, D, generator=g))
    gs = dev(torch.randn(B, SD, generator=g))
    gd0 = dev(torch.randn(B, De, generator=g))
    dsin = dev(torch.randn(B, SD, generator=g))
    dxd = dev(torch.randn(B, 48 + De, generator=g))
    ddin = dxd[:, 48:]  # strided view like dxd[t][:, Hd:]
    first = dev((torch.rand(B, generator=g) < 0.4).float())
    W = dev(torch.randn(SD, N, generator=g) / math.sqrt(SD))
    C0 = dev(torch.randn(B, N, generator=g))
    a = dict(gs=gs.clone(), gd=gd0.clone(), ds0=torch.zeros(SD).cuda(), dd0=torch.zeros(De).cuda(), dl=dlogit.clone(),
             C=C0.clone())
    if with_carry:
        ops.obs_carry_st_bwd(dsin, ddin, first, a["gs"], a["gd"], a["ds0"], a["dd0"], logit, a["dl"], unimix=0.01)
    else:
        ops.onehot_st_bwd(logit, a["gs"].view(B, S, D), a["dl"], unimix=0.01, accumulate=True)
    ops.gemm(a["dl"].view(B, SD), W, a["C"], transB=False, accumulate="atomic")
    b = dict(gd=gd0.clone(), ds0=torch.zeros(SD).cuda(), dd0=torch.zeros(De).cuda(),
             dl=torch.full((B, S, D), float("nan"), device="cuda"), C=C0.clone())
    ops.scan_carry_st_gemm(gs, logit, dlogit, b["dl"], W, b["C"], unimix=0.01,
                           carry=(dsin, ddin, first, b["gd"], b["ds0"], b["dd0"]) if with_carry else None)
    assert_close(b["dl"], a["dl"].cpu(), tol=2e-6, what="dlogit")
    assert_close(b["C"], a["C"].cpu(), tol=1e-5 * math.sqrt(SD / 64), what="C")
    if with_carry:
        assert torch.equal(b["gd"], a["gd"])
        assert_close(b["ds0"], a["ds0"].cpu(), tol=1e-5, what="dstoch0"), assert_close(b["dd0"], a["dd0"].cpu(), tol=1e-5, what="ddeter0")


@pytest.mark.parametrize("M,De,Hd", [(16, 512, 512), (32, 1024, 512), (5, 256, 64), (64, 512, 1024)])
def test_scan_grubwd_gemm_equals_two_launches(ops, M, De, Hd):
    """dv3_scan_gru_factors + dv3_scan_grubwd_gemm == dv3_gru_bwd followed by the atomically accumulating few-row GEMM
    dxd += dgpre W_gru, whose right half the cell's direct path pre-loads (networks.py:760-768 reversed)."""
    g = torch.Generator().manual_seed(M + De + Hd)
    N = Hd + De
    gpre = dev(torch.randn(M, 3 * De, generator=g))
    gam, bet = dev(1 + 0.1 * torch.randn(3 * De, generator=g)), dev(0.1 * torch.randn(3 * De, generator=g))
    h = dev(torch.randn(M, De, generator=g))
    gd = dev(torch.randn(M, De + 32, generator=g))[:, :De]  # row-strided view
    W = dev(torch.randn(3 * De, N, generator=g) / math.sqrt(3 * De))
    hn, mean, rstd = torch.empty(M, De).cuda(), torch.empty(M).cuda(), torch.empty(M).cuda()
    ops.gru_fwd(gpre, gam, bet, h, hn, mean, rstd)
    a = dict(dp=torch.empty(M, 3 * De).cuda(), dxd=torch.zeros(M, N).cuda(), dg=torch.ones(3 * De).cuda(),
             db=torch.ones(3 * De).cuda())
    b = dict(dp=torch.empty(M, 3 * De).cuda(), dxd=torch.zeros(M, N).cuda(), dg=torch.ones(3 * De).cuda(),
             db=torch.ones(3 * De).cuda())
    ops.gru_bwd(gd, gpre, gam, bet, h, mean, rstd, a["dp"], a["dxd"][:, Hd:], a["dg"], a["db"])
    ops.gemm(a["dp"], W, a["dxd"], transB=False, accumulate="atomic")
    fac = [torch.empty(M, 3 * De).cuda(), torch.empty(M, 3 * De).cuda(), torch.empty(M, De).cuda(),
           torch.empty(M, De).cuda(), torch.empty(M, De).cuda()]
    ops.scan_gru_factors(gpre, gam, bet, h, mean, rstd, *fac)
    ops.scan_grubwd_gemm(gd, *fac, gam, rstd, b["dp"], b["dxd"][:, Hd:], W, b["dxd"], b["dg"], b["db"])
    assert_close(b["dp"], a["dp"].cpu(), tol=5e-6, what="dp")
    assert_close(b["dxd"], a["dxd"].cpu(), tol=1e-5 * math.sqrt(3 * De / 64), what="dxd")
    assert_close(b["dg"], a["dg"].cpu(), tol=2e-5, what="dgamma"), assert_close(b["db"], a["db"].cpu(), tol=2e-5, what="dbeta")
