"""Tensor-level wrappers over the libdv3hip C ABI (one function per entry point).

Host-side contract checks live here: every wrapper validates device, dtype, contiguity and the
shapes the kernel and its grid assume BEFORE anything is launched (a faulting kernel can take the
whole node down).  All launches go to torch's current stream and are hipGraph-capturable.
PyTorch is used for device memory and streams only; no ATen math on this path.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _dev, _lib

F32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _f32(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != F32:
        raise TypeError(f"{name}: expected a CUDA/HIP float32 tensor, got "
                        f"{type(t).__name__} {getattr(t, 'dtype', None)} {getattr(t, 'device', None)}")


def _rows2d(t: torch.Tensor, name: str):
    """A 2-D (possibly row-strided) view with unit inner stride -> (rows, cols, ld)."""
    _f32(t, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1) or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        raise ValueError(f"{name}: need a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} "
                         f"strides {t.stride()}")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return t.shape[0], t.shape[1], ld


def _contig(t: torch.Tensor, name: str, dtype=F32) -> None:
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise TypeError(f"{name}: expected a contiguous CUDA/HIP {dtype} tensor")


class _Profile:
    """Optional per-launch accounting (bench.py's roofline leg): HIP events around every launch on the
    launch stream, keyed by kernel, with the algorithmic FLOPs / bytes the caller declares."""

    def __init__(self):
        self.enabled = False
        self.by_shape = False
        self.prefix = ""  # put in front of every key (tools/segment_profile.py: the segment of the update being launched)
        self.records = {}  # key -> [launches, flops, bytes, [event pairs]]

    def start(self):
        self.enabled, self.records = True, {}

    def stop(self):
        """-> {key: dict(launches, flops, bytes, ms)} (synchronises)."""
        self.enabled = False
        torch.cuda.synchronize()
        out = {}
        for k, (n, fl, by, evs) in self.records.items():
            out[k] = dict(launches=n, flops=fl, bytes=by, ms=sum(a.elapsed_time(b) for a, b in evs))
        return out


PROFILE = _Profile()


def _call(fn_name: str, *args, key=None, flops=0.0, nbytes=0.0) -> None:
    lib = _lib.load()
    if PROFILE.enabled:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(getattr(lib, fn_name)(*args), fn_name)
        b.record()
        rec = PROFILE.records.setdefault(PROFILE.prefix + (key or fn_name), [0, 0.0, 0.0, []])
        rec[0] += 1
        rec[1] += flops
        rec[2] += nbytes
        rec[3].append((a, b))
        return
    _lib.check(getattr(lib, fn_name)(*args), fn_name)


_TILE_NAMES = {0: "128x128x16", 1: "64x64x32", 2: "32x128x32", 3: "skinny16", 4: "128x128x32", 5: "64x64x64", 6: "32x64x64s2", 7: "narrowN",
               8: "32x32x64s4", 9: "direct32x64", 10: "directTN32x64", 11: "l16", 12: "l16_64x96", 13: "l16_64x64",
               14: "l16_32x64", 15: "l16_128x128", 16: "l16_128x64", 17: "l16_64x128"}


_TILE_TEMPLATES = {0: "2, 2, 2, 2, 16, 1", 1: "2, 2, 1, 1, 32, 1", 2: "1, 4, 1, 1, 32, 1", 4: "2, 2, 2, 2, 32, 1",
                   5: "2, 2, 1, 1, 64, 1", 6: "1, 2, 1, 1, 64, 2", 8: "1, 1, 1, 1, 64, 4"}


def _l16_auto_name(M: int, N: int) -> str:
    """Which k-contiguous LDS tile the library picks for tile 11 and for the split-output entry (csrc/gemm.hip
    launch_l16): profile keys name the kernel that RUNS, so that bench.py's dominant key is a row of the rocprofv3
    kernel trace."""
    wgs = lambda bm, bn: -(-M // bm) * -(-N // bn)
    return "l16_128x128" if wgs(128, 128) >= 448 else ("l16_64x64" if wgs(64, 64) >= 512 else "l16_32x64")


def kernel_symbol(key: str) -> str:
    """Profile key -> the C++ kernel name rocprofv3 prints (to match bench.py's roofline with profiles/)."""
    import re

    if key.startswith("gemm_kernel<l16+sample"):
        return "void dv3::gemm_l16_kernel<32, 64, 1, 1>(dv3::GemmParams)"
    if key.startswith("gemm_kernel<direct32x64+sample"):
        return "void dv3::gemm_direct_kernel<true, 4, 2, 1, 0>(dv3::GemmParams)"
    if key.startswith("gemm_tn_grouped_kernel"):
        return "void dv3::gemm_tn_grouped_kernel<dv3::TileShape<2, 2, 1, 1, 32, 1> >(dv3::GroupParams)"
    m = re.match(r"gemm_kernel<([^,]+),tA=(\d),tB=(\d)>", key)
    if m:
        tile = {v: k for k, v in _TILE_NAMES.items()}[m.group(1)]
        ta, tb = ("true" if m.group(2) == "1" else "false"), ("true" if m.group(3) == "1" else "false")
        if tile in (3, 7):
            return f"void dv3::gemm_skinny_kernel<{tb if tile == 3 else 'true'}, 1>(dv3::GemmParams)"
        if tile == 9:
            return (f"void dv3::gemm_direct_kernel<{tb}, 4, 2, 0, 0>(dv3::GemmParams)" if tb == "true"
                    else f"void dv3::gemm_direct_kernel<{tb}, 4, 1, 0, 1>(dv3::GemmParams)")
        if tile == 10:
            return "void dv3::gemm_direct_tn_kernel<4, 1>(dv3::GemmParams)"
        if tile >= 11:
            return {12: "void dv3::gemm_l16_kernel<64, 96, 1, 0>(dv3::GemmParams)",
                    13: "void dv3::gemm_l16_kernel<64, 64, 1, 0>(dv3::GemmParams)",
                    15: "void dv3::gemm_l16_kernel<128, 128, 1, 0>(dv3::GemmParams)",
                    16: "void dv3::gemm_l16_kernel<128, 64, 1, 0>(dv3::GemmParams)",
                    17: "void dv3::gemm_l16_kernel<64, 128, 1, 0>(dv3::GemmParams)"}.get(
                        tile, "void dv3::gemm_l16_kernel<32, 64, 1, 0>(dv3::GemmParams)")
        return f"void dv3::gemm_kernel<dv3::TileShape<{_TILE_TEMPLATES[tile]}>, {ta}, {tb}>(dv3::GemmParams)"
    if key.startswith("conv_wgrad_tile_kernel"):
        return "void dv3::conv_wgrad_tile_kernel<32, 64>(float const*, float const*, float*, int, int, int)"
    m = re.match(r"conv_wgrad_c3_kernel<(\d+)>", key)
    if m:
        return f"void dv3::conv_wgrad_c3_kernel<{m.group(1)}>(float const*, float const*, float*, int, int, int)"
    m = re.match(r"conv_wgrad_kernel<([^,>]+)(,c3)?>", key)
    if m:
        tile = {v: k for k, v in _TILE_NAMES.items()}[m.group(1)]
        return (f"void dv3::conv_wgrad_kernel<dv3::TileShape<{_TILE_TEMPLATES[tile]}>, "
                f"{'true' if m.group(2) else 'false'}>(dv3::WgradParams)")
    return key


# <= 512k outputs (1024 imagination rows x 512 columns): the register-direct kernel (tile 9: no LDS staging,
# 16x16x4 MFMA, 32 x 64 outputs per workgroup, K over its waves).  A/B inside one box, ms per update:
# 32x64 LDS tile (6) 22.46, 32x32 LDS tile with K over the waves (8) 21.65, direct (9) 21.42.
_SMALL_TILE = _dev.value("DV3_SMALL_TILE", 9)
# 512k .. 2M outputs (1024 rows x 1024 / 1536 columns): the k-contiguous LDS tile (11: 32 x 64, ds_read_b128
# fragments): GRU matmul 35.9 us against 44.0 direct (9) and 48.3 on the k-major 32x64 LDS tile (6); 1024x1024x512
# 14.5 against 16.7.  Falls back to 9 where its alignment preconditions do not hold.
_MID_TILE = _dev.value("DV3_MID_TILE", 11)


def l16_ok(A, A2, B, transA, transB, K, K1, lda, lda2, ldb) -> bool:
    """Preconditions of the k-contiguous LDS tile kernel (tile 11; csrc/gemm.hip l16_ok)."""
    return (not transA and transB and K >= 32 and K % 32 == 0 and K1 % 32 == 0 and lda % 4 == 0 and ldb % 4 == 0
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0
            and (A2 is None or (lda2 % 4 == 0 and A2.data_ptr() % 16 == 0)))


_BT_MIN_FLOPS = _dev.value("DV3_BT_MIN_FLOPS", 7e9, float)
_BT = {}


def _bt_scratch(K, N, device):
    """[N, K] scratch for the transposed copy of a [K, N] operand (persistent: graph replays read the same buffer).
    One per launch stream: two products of equal (K, N) on different streams (graph branches) must not share it."""
    key = (K, N, str(device), _stream())
    t = _BT.get(key)
    if t is None:
        t = torch.empty(N, K, device=device, dtype=F32)
        _BT[key] = t
    return t


def _legacy_tile(M: int, N: int) -> int:
    """Tile for y = x W^T shapes the k-contiguous LDS kernel cannot take (alignment, K % 32, transposed operands)."""
    t64 = -(-M // 64) * -(-N // 64)
    if t64 <= 512:
        return _SMALL_TILE if t64 <= 128 else 9
    c128 = ((-(-M // 128) * -(-N // 128)) + 255) // 256 * 4
    c64 = (t64 + 255) // 256 * 1
    return 1 if c64 < c128 else 4


def pick_gemm_tile(M: int, N: int, wgrad: bool = False, K: int = 0) -> int:
    """Tile choice (mirrors csrc/gemm.hip pick_tile).  Few rows: the few-row kernel.  Weight gradients: the 128x128
    LDS tile for big outputs with a long reduction, else the register-direct TN kernel.  y = x W^T by output size
    (t64 = number of 64 x 64 tiles), measured with tools/gemm_bench.py (gpurun_out/r02z):
      t64 <= 128            register-direct 32 x 64 (9)                1024 x 512 x 512: 9.3 us vs 10.2 (l16)
      t64 <  512            k-contiguous LDS tile 32 x 64 (14)         1024 x 1536 x 1024: 35.9 vs 44.0 (9)
      512 <= t64 <= 1024    k-contiguous LDS tile 64 x 64 (13)         2048 x 1024 x 1024: 46.6 vs 54.7 (1); 4096 x 1024 x 1024: 87.6 vs 96.8 (4)
      wide / tall, <= 2048  32 x 64 again (14)                         2048 x 3072 x 1536: 208 vs 254 (1); 15360 x 512 x 512: 91 vs 97 (4)
      >= 448 128x128 tiles  k-contiguous LDS tile 128 x 128 (15)       4096^3: 1059 us = 130 TFLOP/s (83 %) vs 1228 (4); 4096 x 12288 x 5120:
                                                                       3891 (132 TFLOP/s) vs 4299 (12) / 4415 (4); 15360 x 512 x 1536: 224 vs 259 (4)
      otherwise             k-major 32x32x2 tiles, 64 x 64 (1) or 128 x 128 (4) by waves of workgroups."""
    if M <= 32:
        return 2
    if wgrad:
        # big outputs with a long reduction: the 128x128 LDS tile (best flops per L2 byte); everything else
        # the register-direct weight-gradient kernel (K split over workgroups, atomics)
        return 4 if (K >= 4096 and M * N >= 512 * 1024) else 10
    t64 = -(-M // 64) * -(-N // 64)
    if t64 <= 128:
        return _SMALL_TILE
    if _MID_TILE != 11:  # development switch: one tile for every mid-size shape
        return _MID_TILE if t64 <= 512 else _legacy_tile(M, N)
    if -(-M // 128) * -(-N // 128) >= 448:
        return 15
    if t64 < 512:
        return 14
    if t64 <= 1024:
        return 13
    if t64 <= 2048 and (N >= 3072 or N <= 512):
        return 14
    return _legacy_tile(M, N)


# The pipelined update (graph.UpdateRunner.step_pipelined, schedule "lanes") launches every kernel on a queue that owns
# 128 of the 256 compute units (engine.Lanes registers its streams here).  On half of the chip the k-contiguous LDS tiles
# trade places for a few shapes (tools/gemm_bench.py --lane, us per launch on a lane, picked / whole-chip choice):
#   1024 x 1536 x 1024 (GRU of the rollout)  64x96 62.9 / 32x64 74.2      2048 x 3072 x 1536 (cfg 3 GRU)  128x64 325 / 32x64 362
#   2048 x 1024 x 1024 (cfg 3 stacked)        128x64 76.8 / 64x64 84.2    15360 x 255 x 512 (head out)     128x128 76.8 / 64x64 86.9
#   14336 x 1024 x 512 (head dgrad to stoch)  64x64 281.6 / 128x128 297.6
#   1024 x 4096 x 1536 (decoder Linear)       128x128 202 / 64x64 236     14336 x 255 x 512 (head out)     128x128 76.7 / 64x64 82.7
#   1024 x 1024 x 1536 | 4096 (rollout dgrad, encoder Linear dgrad)  64x64 62.3 | 157 / 32x64 65.2 | 164
#   14336 x 512 x 512 (head layers)           64x64 146.7 / 128x128 150.6     (r04 sweep: profiles/r04_gemm_lane_sweep.txt)
# All of these tiles run the same K loop (32-wide K tiles, v_mfma_f32_16x16x4_f32 in ascending k): switching among them
# changes no bit of the result (tests/test_kernels_gpu.py::test_l16_tiles_agree_bit_for_bit), so the lanes schedule still
# computes exactly what the serial update computes.
LANE_STREAMS = {}  # stream handle -> compute units of its queue
_LANE_TILES = {(1024, 1536): 12, (2048, 3072): 16, (2048, 1024): 16, (15360, 255): 15, (14336, 1024): 13,
               (1024, 4096): 15, (14336, 255): 15, (1024, 1024): 13, (14336, 512): 13,
               (2048, 512): 13, (28672, 512): 13, (28672, 255): 13}  # (cfg 3: profiles/r04_gemm_lane_sweep_cfg3.txt)


class gemm_group:
    """`with ops.gemm_group():` -- the weight gradients C += A^T B issued inside (ops.gemm(transA=True, transB=False,
    accumulate=True) with a reduction of at most `max_k` rows) are collected and launched as ONE grid at exit
    (dv3_gemm_tn_grouped_f32): the world model's K = B*T = 1024 weight gradients are 64 ... 1024 output tiles each, one
    by one they need split-K with atomics to cover the chip and pay a launch boundary per 20-60 us.  The caller
    guarantees what deferring a launch to the end of the block needs: nothing inside the block reads these C or
    overwrites these A / B.  Two products into overlapping C (never issued by the engines; checked) end the current
    grid first.  Longer reductions (the behaviour's 14 k-row products) are launched as before: they need split-K."""

    current = None
    MAX = 48

    def __init__(self, max_k: int = 2048, enabled: bool = True):
        self.max_k, self.enabled = max_k, enabled
        self.items, self.stream = [], None
        self.sums = []  # the cluster's bias gradients (ops.colsum(..., accumulate=True)): one more grid at exit

    def __enter__(self):
        self._prev = gemm_group.current
        gemm_group.current = self if self.enabled else None
        return self

    def __exit__(self, et, ev, tb):
        gemm_group.current = self._prev
        if et is None:
            self.flush()
        return False

    def takes(self, M, N, K) -> bool:
        return 32 <= K <= self.max_k and M > 32 and N > 32  # (a K = 1 outer product -- the init state's -- is a launch of its own)

    def add(self, A, lda, B, ldb, C, ldc, M, N, K, s):
        if self.stream is not None and s != self.stream:
            self.flush()
        lo = C.data_ptr()
        if len(self.items) >= self.MAX or any(self._overlap(lo, ldc, M, N, it[4], it[5], it[6], it[7]) for it in self.items):
            self.flush()
        self.stream = s
        self.items.append((A.data_ptr(), lda, B.data_ptr(), ldb, lo, ldc, M, N, K))

    @staticmethod
    def _overlap(p1, ld1, M1, N1, p2, ld2, M2, N2) -> bool:
        """Do two row-strided float matrices share an element?  Column slices of one weight gradient ([:, :k1] and
        [:, k1:] of the same rows: the two K-segments of a Linear over a concatenation) interleave in memory without
        overlapping: with equal row pitch they are rectangles on one grid."""
        if p1 > p2:
            p1, ld1, M1, N1, p2, ld2, M2, N2 = p2, ld2, M2, N2, p1, ld1, M1, N1
        if p2 >= p1 + 4 * ((M1 - 1) * ld1 + N1):
            return False
        d = (p2 - p1) // 4
        r, c = d // ld1, d % ld1
        if ld1 == ld2 and (p2 - p1) % 4 == 0 and N1 <= ld1 and c + N2 <= ld1:
            return r < M1 and c < N1  # (rectangle 2 starts at (r, c) >= (0, 0) of rectangle 1's grid)
        return True

    def add_colsum(self, x, ldx, out, R, N, s):
        if self.stream is not None and s != self.stream:
            self.flush()
        lo = out.data_ptr()
        if len(self.sums) >= self.MAX or any(lo < q + 4 * m and q < lo + 4 * N for _, _, q, _, m in self.sums):
            self.flush()
        self.stream = s
        self.sums.append((x.data_ptr(), ldx, lo, R, N))

    def flush(self):
        import ctypes

        sums, self.sums = self.sums, []
        if sums:
            n = len(sums)
            col = lambda j, ty: (ty * n)(*[it[j] for it in sums])
            _call("dv3_colsum_grouped", n, col(0, ctypes.c_void_p), col(1, ctypes.c_long), col(2, ctypes.c_void_p),
                  col(3, ctypes.c_long), col(4, ctypes.c_int), self.stream,
                  key="dv3_colsum_grouped" + (f"[{n} sums]" if PROFILE.by_shape else ""),
                  nbytes=sum(4.0 * it[3] * it[4] for it in sums))
        items, self.items = self.items, []
        if not items:
            return
        n = len(items)
        col = lambda j, ty: (ty * n)(*[it[j] for it in items])
        flops = sum(2.0 * it[6] * it[7] * it[8] for it in items)
        _call("dv3_gemm_tn_grouped_f32", n, col(0, ctypes.c_void_p), col(1, ctypes.c_long), col(2, ctypes.c_void_p),
              col(3, ctypes.c_long), col(4, ctypes.c_void_p), col(5, ctypes.c_long), col(6, ctypes.c_int),
              col(7, ctypes.c_int), col(8, ctypes.c_int), (ctypes.c_int * n)(*([1] * n)), self.stream,
              key="gemm_tn_grouped_kernel<64x64x32>" + (f"[{n} products]" if PROFILE.by_shape else ""), flops=flops,
              nbytes=sum(4.0 * (it[8] * (it[6] + it[7]) + it[6] * it[7]) for it in items))


# ---------------------------------------------------------------------------------------------
def gemm(A, B, C, *, transA=False, transB=True, A2=None, bias=None, accumulate=False, tile=-1):
    """C (+)= [A|A2] @ op(B) + bias.   op(B)[K,N]: transB -> B is [N,K]; else B is [K,N].

    accumulate: False (overwrite), True (C += ..., fixed summation order on the few-row path) or "atomic"
    (C += ..., the few-row kernel may split K over workgroups and add the partial tiles atomically: order not
    fixed -- used for the gradients of the reverse scan, never for the forward values that feed sampling)."""
    acc_flag = 2 if accumulate == "atomic" else int(bool(accumulate))
    ra, ca, lda = _rows2d(A, "A")
    rb, cb, ldb = _rows2d(B, "B")
    M, N, ldc = _rows2d(C, "C")
    Ka = ra if transA else ca
    Ma = ca if transA else ra
    Kb, Nb = (cb, rb) if transB else (rb, cb)
    K1 = Ka
    K = Ka
    lda2 = 0
    if A2 is not None:
        if transA:
            raise ValueError("A2 requires transA=False")
        r2, c2, lda2 = _rows2d(A2, "A2")
        if r2 != Ma:
            raise ValueError("A2 rows mismatch")
        K = Ka + c2
    if Ma != M or Nb != N or Kb != K:
        raise ValueError(f"gemm shape mismatch: A{'^T' if transA else ''}[{Ma},{K}] B[{Kb},{Nb}] C[{M},{N}]")
    if bias is not None:
        _contig(bias, "bias")
        if bias.numel() != N:
            raise ValueError("bias size mismatch")
    s = _stream()
    grp = gemm_group.current
    if (grp is not None and tile < 0 and transA and not transB and A2 is None and bias is None and accumulate is True
            and grp.takes(M, N, K)):
        grp.add(A, lda, B, ldb, C, ldc, M, N, K, s)
        return C
    if A2 is not None and (K1 % (64 if tile in (5, 6, 8) else 32)) != 0:
        # segment edge not on a K-tile boundary: two passes, the second accumulating
        Bv1 = B[:, :K1] if transB else B[:K1]
        Bv2 = B[:, K1:] if transB else B[K1:]
        gemm(A, Bv1, C, transA=False, transB=transB, bias=bias, accumulate=accumulate, tile=tile)
        gemm(A2, Bv2, C, transA=False, transB=transB, accumulate=accumulate or True, tile=tile)
        return C
    if (tile < 0 and not transA and not transB and A2 is None and 2.0 * M * N * K >= _BT_MIN_FLOPS
            and pick_gemm_tile(M, N, False, K) >= 11 and K % 32 == 0 and lda % 4 == 0 and A.data_ptr() % 16 == 0):
        # big data gradient against a [K][N] weight: a transposed copy (2 K N floats of traffic, < 1 % of the product)
        # puts it in the y = x B^T form of the k-contiguous LDS tiles (14336 x 1024 x 512: 155 + 5 us against 177;
        # 14336 x 512 x 512: 81 + 5 against 100)
        Bt = _bt_scratch(K, N, B.device)
        transpose2d(B, Bt)
        return gemm(A, Bt, C, transA=False, transB=True, bias=bias, accumulate=accumulate)
    if tile < 0:
        tile = pick_gemm_tile(M, N, bool(transA), K)
        if tile in (13, 14, 15) and LANE_STREAMS.get(s, 256) <= 128:
            tile = _LANE_TILES.get((M, N), tile)
        if tile in (4, 10) and transA and not transB and A2 is None and bias is None and LANE_STREAMS.get(s, 256) <= 128:
            # weight gradients on a 128-CU lane (tools/gemm_bench.py --lane --only wgrad, us): long reductions on the k-major
            # 128x128x16 tile -- 512 x 512 x 14336: 151 against 180 (register-direct) / 193 (128x128x32); 512 x 1536 x 15360:
            # 423 against 554 -- and the K = 1024 products of the world model up to 2 M outputs on the 64x64x32 tile
            # (1536 x 1024: 74.8 against 85.5; 1024 x 512: 30.9 against 35.3; 512 x 4608 stays register-direct: 118.9)
            if K >= 4096:
                tile = 0
            elif M * N <= 2 * 1024 * 1024:
                tile = 1
        if M <= 32 and not transA:
            tile = 3
        if 32 < M <= 128 and tile == 9 and not transA and (A2 is None or K1 % 16 == 0):
            tile = 3  # few-row kernel over row blocks: 4x the workgroups of the 32 x 64 direct tile (acting-step encoder)
        if tile in (6, 8) and A2 is not None and (K1 % 64) != 0:
            tile = 1
        if tile >= 11 and not l16_ok(A, A2, B, transA, transB, K, K1, lda, lda2, ldb):
            tile = _legacy_tile(M, N)
        if tile == 9 and (transA or (A2 is not None and (K1 % 16) != 0)):
            tile = (8 if M * N <= 512 * 1024 else 6) if not (A2 is not None and (K1 % 64) != 0) else 1
        if tile == 10 and (transB or A2 is not None or bias is not None):
            tile = 1
        if N <= 32 and M > 32 and not transA and transB and A2 is None:
            tile = 7
    _call("dv3_gemm_f32", int(transA), int(transB), M, N, K, _ptr(A), lda, _ptr(A2), lda2, K1, _ptr(B), ldb,
          _ptr(C), ldc, _ptr(bias), acc_flag, tile, s,
          key=(f"gemm_kernel<{_l16_auto_name(M, N) if tile == 11 else _TILE_NAMES[tile]},tA={int(transA)},"
               f"tB={int(transB)}>" + (f"[{M}x{N}x{K}]" if PROFILE.by_shape else "")), flops=2.0 * M * N * K,
          nbytes=4.0 * (M * K + N * K + M * N))
    return C


def tensorstats(x, out4, *, shift=None, scale=None):
    """out4 [4] = mean, std (unbiased), min, max of (x - shift) / scale (shift / scale: optional 1-element device tensors)."""
    _contig(x, "x"), _contig(out4, "out4")
    if out4.numel() != 4:
        raise ValueError("tensorstats output must have 4 elements")
    for t, nm in ((shift, "shift"), (scale, "scale")):
        if t is not None:
            _contig(t, nm)
    _call("dv3_tensorstats", _ptr(x), x.numel(), _ptr(shift), _ptr(scale), _ptr(out4), _stream())
    return out4


def tensorstats_multi(jobs, out):
    """jobs: up to 6 tuples (x, shift, scale); out [len(jobs), 4] = mean / std / min / max of each, one launch."""
    if not 1 <= len(jobs) <= 6:
        raise ValueError("tensorstats_multi takes 1..6 tensors")
    _contig(out, "out")
    if out.numel() != 4 * len(jobs):
        raise ValueError("tensorstats_multi output must have 4 elements per tensor")
    args = []
    for x, shift, scale in jobs:
        _contig(x, "x")
        if x.numel() == 0:
            raise ValueError("tensorstats_multi: empty tensor")
        for t, nm in ((shift, "shift"), (scale, "scale")):
            if t is not None:
                _contig(t, nm)
        args += [_ptr(x), x.numel(), _ptr(shift), _ptr(scale)]
    args += [None, 0, None, None] * (6 - len(jobs))
    _call("dv3_tensorstats_multi", len(jobs), *args, _ptr(out), _stream())
    return out


def concat_flat(parts, dst):
    """dst[flat] = concatenation of up to 6 contiguous float tensors, one launch (the acting step packs its outputs
    for a single device-to-host hop; dv3hip.graph.PolicyRunner)."""
    if not 1 <= len(parts) <= 6:
        raise ValueError("concat_flat takes 1..6 parts")
    _contig(dst, "dst")
    args, total = [], 0
    for t in parts:
        _contig(t, "part")
        args += [_ptr(t), t.numel()]
        total += t.numel()
    if total != dst.numel():
        raise ValueError("concat_flat size mismatch")
    args += [None, 0] * (6 - len(parts))
    _call("dv3_concat6", *args, _ptr(dst), _stream())
    return dst


def gemm_split_ok(A, B) -> bool:
    """Preconditions of ops.gemm_split (the k-contiguous LDS tile kernel)."""
    if A.dim() != 2 or B.dim() != 2 or A.stride(1) != 1 or B.stride(1) != 1:
        return False
    K = A.shape[1]
    return l16_ok(A, None, B, False, True, K, K, A.stride(0), 0, B.stride(0))


def gemm_split(A, B, C, C2, *, accumulate=False, accumulate2=False):
    """[C | C2] (+)= A @ B^T: the first C.shape[1] output columns go to C, the rest to C2, each with its own
    accumulate flag (one launch for two destinations)."""
    M, K, lda = _rows2d(A, "A")
    N, Kb, ldb = _rows2d(B, "B")
    M1, n1, ldc = _rows2d(C, "C")
    M2, n2, ldc2 = _rows2d(C2, "C2")
    if Kb != K or M1 != M or M2 != M or n1 + n2 != N or n1 % 16:
        raise ValueError("gemm_split shapes mismatch")
    _call("dv3_gemm_split_f32", M, N, K, _ptr(A), lda, _ptr(B), ldb, _ptr(C), ldc, int(bool(accumulate)), _ptr(C2),
          ldc2, n1, int(bool(accumulate2)), _stream(),
          key=f"gemm_kernel<{_l16_auto_name(M, N)},tA=0,tB=1>" + (f"[{M}x{N}x{K}]" if PROFILE.by_shape else ""),
          flops=2.0 * M * N * K, nbytes=4.0 * (M * K + N * K + M * N))


def ln_act_fwd(x, gamma, beta, y, mean=None, rstd=None, *, act=True, chw_group=0):
    R, N, ldx = _rows2d(x, "x")
    _contig(gamma, "gamma"), _contig(beta, "beta")
    if gamma.numel() != N or beta.numel() != N:
        raise ValueError("LN affine size mismatch")
    if chw_group:
        _contig(y, "y")
        if y.numel() != R * N or R % chw_group:
            raise ValueError("chw output size mismatch")
        ldy = N
    else:
        Ry, Ny, ldy = _rows2d(y, "y")
        if (Ry, Ny) != (R, N):
            raise ValueError("y shape mismatch")
    for t, nm in ((mean, "mean"), (rstd, "rstd")):
        if t is not None:
            _contig(t, nm)
            if t.numel() != R:
                raise ValueError(nm + " size mismatch")
    _call("dv3_ln_act_fwd", _ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(y), ldy, _ptr(mean), _ptr(rstd), R, N,
          int(act), int(chw_group), _stream(),
          key="dv3_ln_act_fwd" + (f"[{R}x{N}]" if PROFILE.by_shape else ""), nbytes=8.0 * R * N)
    return y


def ln_act_bwd(dy, x, gamma, beta, mean, rstd, dx, dgamma=None, dbeta=None, *, act=True, chw_group=0,
               accumulate_dx=False):
    R, N, ldx = _rows2d(x, "x")
    if chw_group:
        _contig(dy, "dy")
        if dy.numel() != R * N:
            raise ValueError("dy size mismatch")
        lddy = N
    else:
        Rd, Nd, lddy = _rows2d(dy, "dy")
        if (Rd, Nd) != (R, N):
            raise ValueError("dy shape mismatch")
    Rx, Nx, lddx = _rows2d(dx, "dx")
    if (Rx, Nx) != (R, N):
        raise ValueError("dx shape mismatch")
    _contig(gamma, "gamma"), _contig(beta, "beta"), _contig(mean, "mean"), _contig(rstd, "rstd")
    if gamma.numel() != N or beta.numel() != N or mean.numel() != R or rstd.numel() != R:
        raise ValueError("LN bwd size mismatch")
    if (dgamma is None) != (dbeta is None):
        raise ValueError("dgamma/dbeta: both or neither")
    if dgamma is not None:
        _contig(dgamma, "dgamma"), _contig(dbeta, "dbeta")
        if dgamma.numel() != N or dbeta.numel() != N:
            raise ValueError("dgamma size mismatch")
    key = "dv3_ln_act_bwd" + (f"[{R}x{N}{' +dgamma' if dgamma is not None else ''}]" if PROFILE.by_shape else "")
    if (dgamma is not None and not chw_group and N % 4 == 0 and 16 <= N <= _LN_PART_MAXN and R * N >= _LN_PART_MIN
            and _LN_TWO_STAGE):
        # big activations (the conv stacks' channel LayerNorms, the 14 k-row head layers): parameter gradients through
        # per-block partial rows and a second launch instead of contended atomics (tools/ln_bench.py: 262 k x 64
        # 78 -> 45 us, 1 M x 32 107 -> 93, 65 k x 128 46 -> 27)
        ws = _ln_partials(x.device)
        _call("dv3_ln_act_bwd_ws", _ptr(dy), lddy, _ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dx),
              lddx, _ptr(dgamma), _ptr(dbeta), R, N, int(act), 0, int(accumulate_dx), _ptr(ws), ws.numel(), _stream(),
              key=key, nbytes=12.0 * R * N)
        return dx
    _call("dv3_ln_act_bwd", _ptr(dy), lddy, _ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dx),
          lddx, _ptr(dgamma), _ptr(dbeta), R, N, int(act), int(chw_group), int(accumulate_dx), _stream(), key=key,
          nbytes=12.0 * R * N)
    return dx


# Partial-sum buffers of dv3_ln_act_bwd_ws ([2048 row blocks][2 N] floats, contents irrelevant between launches): one
# per launch stream -- two lanes may run a LayerNorm backward at the same time.
_LN_TWO_STAGE = _dev.flag("DV3_LN_TWO_STAGE", True)
_LN_PART_MAXN, _LN_PART_MIN = 512, 2 * 1024 * 1024
_LN_PART = {}


def _ln_partials(device):
    key = (str(device), _stream())
    t = _LN_PART.get(key)
    if t is None:
        t = torch.empty(2048 * 2 * _LN_PART_MAXN, dtype=F32, device=device)
        _LN_PART[key] = t
    return t


def gru_fwd(p, gamma, beta, h, h_new, mean, rstd, *, next_blend=None):
    """next_blend = (next_first [M], init [De], next_out [M,De]): also write the next observe step's reset blend of
    h_new (fused second output)."""
    M, N3, ldp = _rows2d(p, "p")
    Mh, De, ldh = _rows2d(h, "h")
    Mn, Dn, ldhn = _rows2d(h_new, "h_new")
    if N3 != 3 * De or Mh != M or (Mn, Dn) != (M, De):
        raise ValueError("gru shapes mismatch")
    _contig(gamma, "gamma"), _contig(beta, "beta"), _contig(mean, "mean"), _contig(rstd, "rstd")
    if gamma.numel() != N3 or beta.numel() != N3 or mean.numel() != M or rstd.numel() != M:
        raise ValueError("gru param sizes mismatch")
    if next_blend is not None:
        nf, init, nout = next_blend
        _contig(nf, "next_first"), _contig(init, "init")
        Mo, Do, ldo = _rows2d(nout, "next_out")
        if nf.numel() != M or init.numel() != De or (Mo, Do) != (M, De):
            raise ValueError("next_blend shapes mismatch")
        _call("dv3_gru_fwd_blend", _ptr(p), ldp, _ptr(gamma), _ptr(beta), _ptr(h), ldh, _ptr(h_new), ldhn, _ptr(mean),
              _ptr(rstd), M, De, _ptr(nf), _ptr(init), _ptr(nout), ldo, _stream(), key="dv3_gru_fwd")
        return h_new
    _call("dv3_gru_fwd", _ptr(p), ldp, _ptr(gamma), _ptr(beta), _ptr(h), ldh, _ptr(h_new), ldhn, _ptr(mean),
          _ptr(rstd), M, De, _stream())
    return h_new


def scan_ln_gemm_ok(K, N) -> bool:
    """Shapes dv3_scan_ln_gemm_fwd takes (K = the LayerNorm width)."""
    return K in (256, 512, 1024) and N % 16 == 0


def scan_ln_gemm(x, gamma, beta, y, mean, rstd, W, C, *, bias=None, accumulate=False):
    """ln_act_fwd(act=True) and C (+)= y @ W^T + bias in ONE launch (the observe scan's obs_out LayerNorm + the
    posterior-logit Linear, networks.py:197-200).  y / mean / rstd may be None."""
    M, K, ldx = _rows2d(x, "x")
    N, Kw, ldw = _rows2d(W, "W")
    Mc, Nc, ldc = _rows2d(C, "C")
    if Kw != K or (Mc, Nc) != (M, N):
        raise ValueError("scan_ln_gemm shapes mismatch")
    if not scan_ln_gemm_ok(K, N):
        raise ValueError(f"scan_ln_gemm: unsupported K={K} N={N}")
    _contig(gamma, "gamma"), _contig(beta, "beta")
    if gamma.numel() != K or beta.numel() != K:
        raise ValueError("scan_ln_gemm param sizes mismatch")
    ldy = 0
    if y is not None:
        My, Ky, ldy = _rows2d(y, "y")
        if (My, Ky) != (M, K):
            raise ValueError("scan_ln_gemm y shape mismatch")
    for t, nm in ((mean, "mean"), (rstd, "rstd")):
        if t is not None:
            _contig(t, nm)
            if t.numel() != M:
                raise ValueError(f"scan_ln_gemm {nm} size mismatch")
    if bias is not None:
        _contig(bias, "bias")
        if bias.numel() != N:
            raise ValueError("scan_ln_gemm bias size mismatch")
    _call("dv3_scan_ln_gemm_fwd", _ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(y), ldy, _ptr(mean), _ptr(rstd), _ptr(W),
          ldw, _ptr(bias), _ptr(C), ldc, M, K, N, int(bool(accumulate)), _stream(),
          key="gemm_kernel<skinny16+ln,tA=0,tB=1>", flops=2.0 * M * N * K, nbytes=4.0 * (N * K + M * (2 * K + N)))
    return C


def scan_lnbwd_gemm_ok(K, N) -> bool:
    return K in (256, 512, 1024) and N % 64 == 0


def scan_ln_factors(x, gamma, beta, mean, rstd, xhat, jac):
    """xhat = (x - mean) * rstd, jac = SiLU'(xhat * gamma + beta) for all rows of a LayerNorm + SiLU layer: what its
    backward needs from the forward pass (scan_lnbwd_gemm reads them; once per update, in front of the reverse scan)."""
    R, K, ldx = _rows2d(x, "x")
    _contig(gamma, "gamma"), _contig(beta, "beta"), _contig(mean, "mean"), _contig(rstd, "rstd")
    _contig(xhat, "xhat"), _contig(jac, "jac")
    if gamma.numel() != K or beta.numel() != K or mean.numel() != R or rstd.numel() != R or xhat.numel() != R * K \
            or jac.numel() != R * K or K % 4:
        raise ValueError("scan_ln_factors sizes mismatch")
    _call("dv3_scan_ln_factors", _ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(xhat), _ptr(jac), R, K,
          _stream(), nbytes=12.0 * R * K)


def scan_lnbwd_gemm(dy, xhat, jac, gamma, rstd, dx, W, C, dgamma=None, dbeta=None):
    """ln_act_bwd(act=True) from the precomputed factors (scan_ln_factors) and C += dx @ W (atomic) in ONE launch
    (reverse observe scan).  W [K, N] may be a column slice of a wider [K, .] weight; C must hold the value being
    added to."""
    M, K, lddy = _rows2d(dy, "dy")
    Md, Kd, lddx = _rows2d(dx, "dx")
    Kw, N, ldb = _rows2d(W, "W")
    Mc, Nc, ldc = _rows2d(C, "C")
    if (Md, Kd) != (M, K) or Kw != K or (Mc, Nc) != (M, N):
        raise ValueError("scan_lnbwd_gemm shapes mismatch")
    if not scan_lnbwd_gemm_ok(K, N):
        raise ValueError(f"scan_lnbwd_gemm: unsupported K={K} N={N}")
    _contig(gamma, "gamma"), _contig(rstd, "rstd"), _contig(xhat, "xhat"), _contig(jac, "jac")
    if gamma.numel() != K or rstd.numel() != M or xhat.numel() != M * K or jac.numel() != M * K:
        raise ValueError("scan_lnbwd_gemm param sizes mismatch")
    if (dgamma is None) != (dbeta is None):
        raise ValueError("dgamma/dbeta: both or neither")
    if dgamma is not None:
        _contig(dgamma, "dgamma"), _contig(dbeta, "dbeta")
        if dgamma.numel() != K or dbeta.numel() != K:
            raise ValueError("scan_lnbwd_gemm dgamma size mismatch")
    _call("dv3_scan_lnbwd_gemm", _ptr(dy), lddy, _ptr(xhat), _ptr(jac), _ptr(gamma), _ptr(rstd), _ptr(dx), lddx,
          _ptr(dgamma), _ptr(dbeta), _ptr(W), ldb, _ptr(C), ldc, M, K, N, _stream(),
          key="gemm_kernel<skinny16+lnbwd,tA=0,tB=0>", flops=2.0 * M * N * K, nbytes=4.0 * (N * K + M * (4 * K + 2 * N)))
    return C


def scan_grubwd_gemm_ok(De, N) -> bool:
    return De in (256, 512, 1024) and N % 64 == 0


def scan_gru_factors(p, gamma, beta, h, mean, rstd, xhat, afac, p1, p2, ah):
    """What GRUCell's backward needs from the forward pass, for all rows at once (scan_grubwd_gemm reads it):
    xhat / afac [R, 3 De], p1 / p2 / ah [R, De]."""
    R, N3, ldp = _rows2d(p, "p")
    Rh, De, ldh = _rows2d(h, "h")
    for t, nm, n in ((gamma, "gamma", N3), (beta, "beta", N3), (mean, "mean", R), (rstd, "rstd", R),
                     (xhat, "xhat", R * N3), (afac, "afac", R * N3), (p1, "p1", R * De), (p2, "p2", R * De),
                     (ah, "ah", R * De)):
        _contig(t, nm)
        if t.numel() != n:
            raise ValueError(f"scan_gru_factors: {nm} size mismatch")
    if N3 != 3 * De or Rh != R:
        raise ValueError("scan_gru_factors shapes mismatch")
    _call("dv3_scan_gru_factors", _ptr(p), ldp, _ptr(gamma), _ptr(beta), _ptr(h), ldh, _ptr(mean), _ptr(rstd), _ptr(xhat),
          _ptr(afac), _ptr(p1), _ptr(p2), _ptr(ah), R, De, _stream(), nbytes=4.0 * R * De * 13)


def scan_grubwd_gemm(g, xhat, afac, p1, p2, ah, gamma, rstd, dp, dh, W, C, dgamma=None, dbeta=None):
    """gru_bwd from the precomputed factors and C += dp @ W (atomic) in ONE launch (reverse observe scan).  dh [M, De]
    (may be a column slice of C) receives the direct path g * (1 - u) with atomic adds: it must hold the value being
    added to, like C."""
    M, De, ldg = _rows2d(g, "g")
    Md, Dd, lddh = _rows2d(dh, "dh")
    Kw, N, ldb = _rows2d(W, "W")
    Mc, Nc, ldc = _rows2d(C, "C")
    if (Md, Dd) != (M, De) or Kw != 3 * De or (Mc, Nc) != (M, N):
        raise ValueError("scan_grubwd_gemm shapes mismatch")
    if not scan_grubwd_gemm_ok(De, N):
        raise ValueError(f"scan_grubwd_gemm: unsupported De={De} N={N}")
    for t, nm, n in ((xhat, "xhat", 3 * M * De), (afac, "afac", 3 * M * De), (p1, "p1", M * De), (p2, "p2", M * De),
                     (ah, "ah", M * De), (gamma, "gamma", 3 * De), (rstd, "rstd", M), (dp, "dp", 3 * M * De)):
        _contig(t, nm)
        if t.numel() != n:
            raise ValueError(f"scan_grubwd_gemm: {nm} size mismatch")
    if (dgamma is None) != (dbeta is None):
        raise ValueError("dgamma/dbeta: both or neither")
    if dgamma is not None:
        _contig(dgamma, "dgamma"), _contig(dbeta, "dbeta")
        if dgamma.numel() != 3 * De or dbeta.numel() != 3 * De:
            raise ValueError("scan_grubwd_gemm dgamma size mismatch")
    _call("dv3_scan_grubwd_gemm", _ptr(g), ldg, _ptr(xhat), _ptr(afac), _ptr(p1), _ptr(p2), _ptr(ah), _ptr(gamma), _ptr(rstd),
          _ptr(dp), _ptr(dh), lddh, _ptr(dgamma), _ptr(dbeta), _ptr(W), ldb, _ptr(C), ldc, M, De, N, _stream(),
          key="gemm_kernel<skinny16+grubwd,tA=0,tB=0>", flops=2.0 * M * N * 3 * De,
          nbytes=4.0 * (N * 3 * De + M * (12 * De + 2 * N)))
    return C


def scan_carry_st_gemm_ok(S, D, N) -> bool:
    return D == 32 and S % 8 == 0 and N % 64 == 0


def scan_carry_st_gemm(gs, logit, dlogit, dlogit_out, W, C, *, unimix, carry=None):
    """Reverse observe scan, first launch of a step: carry (optional) + straight-through gradient of the posterior
    sample + C += dlogit_out @ W (atomic).  gs [B, S*D], logit / dlogit / dlogit_out [B, S, D] contiguous, W [S*D, N].
    carry = (dsin [B, S*D], ddin [B, De], first [B], gd [B, De], dstoch0 [S*D], ddeter0 [De]) of the NEXT step."""
    _contig(logit, "logit"), _contig(dlogit, "dlogit"), _contig(dlogit_out, "dlogit_out"), _contig(gs, "gs")
    if logit.dim() != 3 or dlogit.shape != logit.shape or dlogit_out.shape != logit.shape:
        raise ValueError("scan_carry_st_gemm: logit / dlogit [B,S,D]")
    B, S, D = logit.shape
    Kw, N, ldb = _rows2d(W, "W")
    Mc, Nc, ldc = _rows2d(C, "C")
    if Kw != S * D or (Mc, Nc) != (B, N) or gs.numel() != B * S * D:
        raise ValueError("scan_carry_st_gemm shapes mismatch")
    if not scan_carry_st_gemm_ok(S, D, N) or dlogit_out.data_ptr() == dlogit.data_ptr():
        raise ValueError(f"scan_carry_st_gemm: unsupported S={S} D={D} N={N} (or aliased output)")
    dsin = ddin = first = gd = ds0 = dd0 = None
    ld_dsin = ld_ddin = De = 0
    if carry is not None:
        dsin, ddin, first, gd, ds0, dd0 = carry
        Bs, SDs, ld_dsin = _rows2d(dsin, "dsin")
        Bd, De, ld_ddin = _rows2d(ddin, "ddin")
        _contig(first, "first"), _contig(gd, "gd"), _contig(ds0, "dstoch0"), _contig(dd0, "ddeter0")
        if (Bs, SDs) != (B, S * D) or Bd != B or first.numel() != B or gd.numel() != B * De or ds0.numel() != S * D \
                or dd0.numel() != De:
            raise ValueError("scan_carry_st_gemm carry shapes mismatch")
    _call("dv3_scan_carry_st_gemm", _ptr(dsin), ld_dsin, _ptr(ddin), ld_ddin, _ptr(first), _ptr(gs), _ptr(gd), _ptr(ds0),
          _ptr(dd0), _ptr(logit), _ptr(dlogit), _ptr(dlogit_out), _ptr(W), ldb, _ptr(C), ldc, B, S, D, De, N,
          float(unimix), _stream(), key="gemm_kernel<skinny16+carry_st,tA=0,tB=0>", flops=2.0 * B * N * S * D,
          nbytes=4.0 * (N * S * D + B * (5 * S * D + 2 * N)))
    return C


def gru_bwd(dh_new, p, gamma, beta, h, mean, rstd, dp, dh, dgamma=None, dbeta=None, *, accumulate_dh=False):
    M, N3, ldp = _rows2d(p, "p")
    Mh, De, ldh = _rows2d(h, "h")
    Mg, Dg, lddhn = _rows2d(dh_new, "dh_new")
    Mp, Np, lddp = _rows2d(dp, "dp")
    Md, Dd, lddh = _rows2d(dh, "dh")
    if N3 != 3 * De or Mh != M or (Mg, Dg) != (M, De) or (Mp, Np) != (M, N3) or (Md, Dd) != (M, De):
        raise ValueError("gru bwd shapes mismatch")
    _contig(gamma, "gamma"), _contig(beta, "beta"), _contig(mean, "mean"), _contig(rstd, "rstd")
    if gamma.numel() != N3 or mean.numel() != M or rstd.numel() != M:
        raise ValueError("gru bwd sizes mismatch")
    if (dgamma is None) != (dbeta is None):
        raise ValueError("dgamma/dbeta: both or neither")
    if dgamma is not None:
        _contig(dgamma, "dgamma"), _contig(dbeta, "dbeta")
        if dgamma.numel() != N3 or dbeta.numel() != N3:
            raise ValueError("dgamma size mismatch")
    _call("dv3_gru_bwd", _ptr(dh_new), lddhn, _ptr(p), ldp, _ptr(gamma), _ptr(beta), _ptr(h), ldh, _ptr(mean),
          _ptr(rstd), _ptr(dp), lddp, _ptr(dh), lddh, _ptr(dgamma), _ptr(dbeta), M, De, int(accumulate_dh), _stream())


def _groups(t, name, D):
    _contig(t, name)
    if t.shape[-1] != D or t.numel() % D:
        raise ValueError(f"{name}: last dim must be {D}")
    return t.numel() // D


class RngStream:
    """Device-resident Philox state {seed, offset} plus a host cursor.  Samplers `take()` disjoint counter
    ranges (a launch argument, so the sequence is static under hipGraph replay); `commit()` advances the
    device offset once for everything taken since the last commit."""

    def __init__(self, device, seed=0):
        self.state = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        self.cursor = 0

    def take(self, n_elems: int) -> int:
        off = self.cursor
        self.cursor += n_elems // 4 + 1
        return off

    def commit(self):
        if self.cursor:
            rng_advance(self.state, self.cursor)
            self.cursor = 0

    def reseed(self, seed: int):
        self.state[0] = seed
        self.state[1] = 0
        self.cursor = 0


class PhasedRng(RngStream):
    """The Philox state of ONE phase of the update (world model or behaviour) while the phases of two different updates
    run side by side (graph.UpdateRunner.step_pipelined).  The serial stream hands out counter ranges in launch order
    -- world model k takes [S, S + w), behaviour k [S + w, S + w + b), world model k+1 starts at S + w + b -- and the
    two phases cannot share one device offset once behaviour k runs beside world model k+1.  Each phase therefore
    carries its own {seed, offset}; a phase's draws use launch-argument offsets from its own state as before, and the
    phase ends with `offset += stride`, stride = w + b (a device scalar the runner fills once both phases have been
    captured): the numbers drawn are exactly those of the serial stream."""

    def __init__(self, device):
        super().__init__(device, 0)
        self.stride = torch.zeros(1, dtype=torch.int64, device=device)
        self.taken = 0  # counters one phase takes (known once the phase has been traced)

    def commit(self):
        pass  # the phase's captured sequence ends with finish_phase() instead

    def finish_phase(self):
        """Last launch of the phase's captured sequence: advance the state past this update's draws of BOTH phases."""
        self.taken, self.cursor = self.cursor, 0
        self.state[1:2].add_(self.stride)


def onehot_sample(logit, out, *, noise=None, rng=None, idx=None, unimix=0.01, mode=False, next_blend=None,
                  forced=None, flips=None):
    """next_blend = (next_first [B], init [S*D], next_out [B,S,D]) for logit [B,S,D]: also write the next observe
    step's reset blend of the sample (fused second output).  idx (int32 [R]) receives the class indices.
    forced (int32 [R], parity tests): emit these classes instead of the kernel's own draw and count the draws
    that differ into flips (int32 [1], incremented)."""
    D = logit.shape[-1]
    R = _groups(logit, "logit", D)
    if _groups(out, "out", D) != R:
        raise ValueError("out size mismatch")
    if noise is not None and _groups(noise, "noise", D) != R:
        raise ValueError("noise size mismatch")
    rng_state, rng_off = None, 0
    if not mode and noise is None:
        if rng is None:
            raise ValueError("sampling needs noise or an RngStream")
        rng_state, rng_off = rng.state, rng.take(R * D)
    for t, nm in ((idx, "idx"), (forced, "forced")):
        if t is not None:
            _contig(t, nm, torch.int32)
            if t.numel() != R:
                raise ValueError(nm + " size mismatch")
    if flips is not None:
        _contig(flips, "flips", torch.int32)
        if flips.numel() != 1:
            raise ValueError("flips: one int32 counter")
    nf = init = nout = init_idx = next_idx = None
    groups = 1
    if next_blend is not None:
        nf, init, nout = next_blend[:3]
        _contig(nf, "next_first"), _contig(init, "init"), _contig(nout, "next_out")
        if R % nf.numel() or init.numel() * nf.numel() != R * D or nout.numel() != R * D:
            raise ValueError("next_blend shapes mismatch")
        groups = R // nf.numel()
        if len(next_blend) > 3:  # (..., init_idx [groups], next_idx [R]): class indices of the blended state
            init_idx, next_idx = next_blend[3], next_blend[4]
            _contig(init_idx, "init_idx", torch.int32), _contig(next_idx, "next_idx", torch.int32)
            if init_idx.numel() != groups or next_idx.numel() != R:
                raise ValueError("next_blend index shapes mismatch")
    _call("dv3_onehot_sample_fwd_ex", _ptr(logit), _ptr(noise), _ptr(rng_state), int(rng_off), _ptr(out), _ptr(idx),
          _ptr(forced), _ptr(flips), R, D, float(unimix), int(mode), _ptr(nf), _ptr(init), _ptr(nout), groups,
          _ptr(init_idx), _ptr(next_idx), _stream(), key="dv3_onehot_sample_fwd")
    return out


def onehot_st_bwd(logit, dstoch, dlogit, *, unimix=0.01, mode=False, accumulate=False):
    D = logit.shape[-1]
    R = _groups(logit, "logit", D)
    if _groups(dstoch, "dstoch", D) != R or _groups(dlogit, "dlogit", D) != R:
        raise ValueError("size mismatch")
    _call("dv3_onehot_st_bwd", _ptr(logit), _ptr(dstoch), _ptr(dlogit), R, D, float(unimix), int(mode),
          int(accumulate), _stream())
    return dlogit


def onehot_ent_logp_fwd(logit, x=None, ent=None, logp=None, *, unimix=0.01):
    D = logit.shape[-1]
    R = _groups(logit, "logit", D)
    if x is not None and _groups(x, "x", D) != R:
        raise ValueError("x size mismatch")
    for t, nm in ((ent, "ent"), (logp, "logp")):
        if t is not None:
            _contig(t, nm)
            if t.numel() != R:
                raise ValueError(nm + " size mismatch")
    _call("dv3_onehot_ent_logp_fwd", _ptr(logit), _ptr(x), _ptr(ent), _ptr(logp), R, D, float(unimix), _stream())


def onehot_ent_logp_bwd(logit, x, dent, dlogp, dlogit, *, unimix=0.01, accumulate=False):
    D = logit.shape[-1]
    R = _groups(logit, "logit", D)
    if _groups(dlogit, "dlogit", D) != R:
        raise ValueError("dlogit size mismatch")
    for t, nm in ((dent, "dent"), (dlogp, "dlogp")):
        if t is not None:
            _contig(t, nm)
            if t.numel() != R:
                raise ValueError(nm + " size mismatch")
    if x is not None and _groups(x, "x", D) != R:
        raise ValueError("x size mismatch")
    _call("dv3_onehot_ent_logp_bwd", _ptr(logit), _ptr(x), _ptr(dent), _ptr(dlogp), _ptr(dlogit), R, D,
          float(unimix), int(accumulate), _stream())


def kl_fwd(post_logit, prior_logit, kl, ent_post=None, ent_prior=None, *, unimix=0.01):
    S, D = post_logit.shape[-2], post_logit.shape[-1]
    _contig(post_logit, "post_logit"), _contig(prior_logit, "prior_logit"), _contig(kl, "kl")
    rows = post_logit.numel() // (S * D)
    if prior_logit.shape != post_logit.shape or kl.numel() != rows:
        raise ValueError("kl shapes mismatch")
    for t, nm in ((ent_post, "ent_post"), (ent_prior, "ent_prior")):
        if t is not None:
            _contig(t, nm)
            if t.numel() != rows:
                raise ValueError(nm + " size mismatch")
    _call("dv3_kl_fwd", _ptr(post_logit), _ptr(prior_logit), _ptr(kl), _ptr(ent_post), _ptr(ent_prior), rows, S, D,
          float(unimix), _stream())


def kl_bwd(post_logit, prior_logit, kl, dpost, dprior, *, unimix, free, dyn_scale, rep_scale, upstream,
           acc_post=False, acc_prior=False):
    S, D = post_logit.shape[-2], post_logit.shape[-1]
    for t, nm in ((post_logit, "post"), (prior_logit, "prior"), (kl, "kl"), (dpost, "dpost"), (dprior, "dprior")):
        _contig(t, nm)
    rows = post_logit.numel() // (S * D)
    if (prior_logit.shape != post_logit.shape or dpost.numel() != post_logit.numel()
            or dprior.numel() != post_logit.numel() or kl.numel() != rows):
        raise ValueError("kl bwd shapes mismatch")
    _call("dv3_kl_bwd", _ptr(post_logit), _ptr(prior_logit), _ptr(kl), _ptr(dpost), _ptr(dprior), rows, S, D,
          float(unimix), float(free), float(dyn_scale), float(rep_scale), float(upstream), int(acc_post),
          int(acc_prior), _stream())


def _disc_rows(logits):
    _contig(logits, "logits")
    if logits.shape[-1] != 255:
        raise ValueError("DiscDist heads have 255 buckets")
    return logits.numel() // 255


def disc_mode_fwd(logits, out):
    R = _disc_rows(logits)
    _contig(out, "out")
    if out.numel() != R:
        raise ValueError("out size mismatch")
    _call("dv3_disc_mode_fwd", _ptr(logits), _ptr(out), R, _stream())
    return out


def disc_mode_bwd(logits, up, dlogits, *, accumulate=False):
    R = _disc_rows(logits)
    _contig(up, "up"), _contig(dlogits, "dlogits")
    if up.numel() != R or dlogits.numel() != R * 255:
        raise ValueError("size mismatch")
    _call("dv3_disc_mode_bwd", _ptr(logits), _ptr(up), _ptr(dlogits), R, int(accumulate), _stream())


def disc_logprob_fwd(logits, x, out):
    R = _disc_rows(logits)
    _contig(x, "x"), _contig(out, "out")
    if x.numel() != R or out.numel() != R:
        raise ValueError("size mismatch")
    _call("dv3_disc_logprob_fwd", _ptr(logits), _ptr(x), _ptr(out), R, _stream())
    return out


def disc_logprob_bwd(logits, x, up, dlogits, *, accumulate=False):
    R = _disc_rows(logits)
    _contig(x, "x"), _contig(up, "up"), _contig(dlogits, "dlogits")
    if x.numel() != R or up.numel() != R or dlogits.numel() != R * 255:
        raise ValueError("size mismatch")
    _call("dv3_disc_logprob_bwd", _ptr(logits), _ptr(x), _ptr(up), _ptr(dlogits), R, int(accumulate), _stream())


def bernoulli_logprob_fwd(logit, x, out):
    for t, nm in ((logit, "logit"), (x, "x"), (out, "out")):
        _contig(t, nm)
    n = logit.numel()
    if x.numel() != n or out.numel() != n:
        raise ValueError("size mismatch")
    _call("dv3_bernoulli_logprob_fwd", _ptr(logit), _ptr(x), _ptr(out), n, _stream())
    return out


def bernoulli_logprob_bwd(logit, x, up, dlogit, *, accumulate=False):
    for t, nm in ((logit, "logit"), (x, "x"), (up, "up"), (dlogit, "dlogit")):
        _contig(t, nm)
    n = logit.numel()
    if x.numel() != n or up.numel() != n or dlogit.numel() != n:
        raise ValueError("size mismatch")
    _call("dv3_bernoulli_logprob_bwd", _ptr(logit), _ptr(x), _ptr(up), _ptr(dlogit), n, int(accumulate), _stream())


def _perm(perm):
    if perm is None:
        return 0, 0
    B, T = perm
    return int(B), int(T)


def image_to_f32(image_u8, out, *, n_images, perm=None):
    """u8 -> u8/255 - 0.5.  perm=(B,T): output image t*B+b reads input image b*T+t."""
    _contig(image_u8, "image", torch.uint8), _contig(out, "out")
    if out.numel() != image_u8.numel() or image_u8.numel() % n_images:
        raise ValueError("size mismatch")
    pixels = image_u8.numel() // n_images
    B, T = _perm(perm)
    if perm is not None and B * T != n_images:
        raise ValueError("perm does not match n_images")
    _call("dv3_image_to_f32", _ptr(image_u8), _ptr(out), n_images, pixels, B, T, _stream())
    return out


def mse_image(recon, image_u8, loss, drecon=None, *, upstream=0.0, perm=None):
    _contig(recon, "recon"), _contig(image_u8, "image", torch.uint8), _contig(loss, "loss")
    n_img = loss.numel()
    if recon.numel() != image_u8.numel() or recon.numel() % max(n_img, 1):
        raise ValueError("size mismatch")
    pixels = recon.numel() // n_img
    if drecon is not None:
        _contig(drecon, "drecon")
        if drecon.numel() != recon.numel():
            raise ValueError("drecon size mismatch")
    B, T = _perm(perm)
    if perm is not None and B * T != n_img:
        raise ValueError("perm does not match n_images")
    _call("dv3_mse_image", _ptr(recon), _ptr(image_u8), _ptr(loss), _ptr(drecon), n_img, pixels, float(upstream),
          B, T, _stream())


def transpose01(x, y):
    """[B,T,...] -> [T,B,...] (contiguous copies)."""
    _contig(x, "x"), _contig(y, "y")
    B, T = x.shape[0], x.shape[1]
    k = x.numel() // (B * T)
    if y.numel() != x.numel() or y.shape[0] != T or y.shape[1] != B:
        raise ValueError("transpose01 shape mismatch")
    _call("dv3_transpose01", _ptr(x), _ptr(y), B, T, k, _stream())
    return y


def colsum(x, out, *, accumulate=False):
    R, N, ldx = _rows2d(x, "x")
    _contig(out, "out")
    if out.numel() != N:
        raise ValueError("out size mismatch")
    grp = gemm_group.current
    if grp is not None and accumulate and N > 32 and R >= 64:
        grp.add_colsum(x, ldx, out, R, N, _stream())
        return out
    _call("dv3_colsum", _ptr(x), ldx, _ptr(out), R, N, int(accumulate), _stream())
    return out


def tanh_fwd(x, y):
    _contig(x, "x"), _contig(y, "y")
    if x.numel() != y.numel():
        raise ValueError("size mismatch")
    _call("dv3_tanh_fwd", _ptr(x), _ptr(y), x.numel(), _stream())
    return y


def tanh_bwd(y, dy, dx, *, accumulate=False):
    _contig(y, "y"), _contig(dy, "dy"), _contig(dx, "dx")
    if dy.numel() != y.numel() or dx.numel() != y.numel():
        raise ValueError("size mismatch")
    _call("dv3_tanh_bwd", _ptr(y), _ptr(dy), _ptr(dx), y.numel(), int(accumulate), _stream())


def pack_conv_weight(w, wp, *, transposed):
    """w: Conv2d [Co,Ci,4,4] (transposed=False) or ConvTranspose2d [Ci,Co,4,4] (transposed=True)."""
    _contig(w, "w"), _contig(wp, "wp")
    if w.dim() != 4 or w.shape[2] != 4 or w.shape[3] != 4 or wp.numel() != w.numel():
        raise ValueError("conv weight must be [*,*,4,4]")
    if transposed:
        Ci, Co = w.shape[0], w.shape[1]
    else:
        Co, Ci = w.shape[0], w.shape[1]
    _call("dv3_pack_conv_weight", _ptr(w), _ptr(wp), Co, Ci, int(transposed), _stream())
    return wp


def im2col_s2(x, cols):
    """x [N,H,W,C] NHWC -> cols [N*(H/2)*(W/2), 16*C], columns in the Conv2d weight's (ci, ky, kx) order."""
    _contig(x, "x"), _contig(cols, "cols")
    if x.dim() != 4 or x.shape[1] % 2 or x.shape[2] % 2:
        raise ValueError("im2col input must be [N,H,W,C] with even H,W")
    N, H, W, C = x.shape
    if cols.numel() != N * (H // 2) * (W // 2) * 16 * C:
        raise ValueError("im2col output size mismatch")
    _call("dv3_im2col_s2", _ptr(x), _ptr(cols), N, H, W, C, _stream())
    return cols


def conv_s2_fwd(x, wp, y, *, Ci, Co, accumulate=False):
    """x [N,H,W,Ci] NHWC, wp packed [Co,16*Ci], y [N,H/2,W/2,Co]."""
    _contig(x, "x"), _contig(wp, "wp"), _contig(y, "y")
    if x.dim() != 4 or x.shape[3] != Ci or x.shape[1] % 2 or x.shape[2] % 2:
        raise ValueError(f"conv input must be [N,H,W,{Ci}] with even H,W")
    N, H, W = x.shape[0], x.shape[1], x.shape[2]
    if wp.numel() != 16 * Ci * Co or y.numel() != N * (H // 2) * (W // 2) * Co:
        raise ValueError("conv shapes mismatch")
    _call("dv3_conv_s2_fwd", _ptr(x), _ptr(wp), _ptr(y), N, H, W, Ci, Co, int(accumulate), _stream(),
          key=f"conv_s2_kernel<Co={Co}>" + (f"[N{N} {H}x{W} Ci{Ci}]" if PROFILE.by_shape else ""), flops=2.0 * N * (H // 2) * (W // 2) * 16 * Ci * Co,
          nbytes=4.0 * (x.numel() + y.numel() + wp.numel()))
    return y


def convT_s2_fwd(x, wp, y, *, Ci, Co, bias=None, out_add=0.0, accumulate=False):
    """x [N,IH,IW,Ci] NHWC, wp packed [4,Co,4*Ci], y [N,2IH,2IW,Co]."""
    _contig(x, "x"), _contig(wp, "wp"), _contig(y, "y")
    if x.dim() != 4 or x.shape[3] != Ci:
        raise ValueError(f"convT input must be [N,IH,IW,{Ci}]")
    N, IH, IW = x.shape[0], x.shape[1], x.shape[2]
    if wp.numel() != 16 * Ci * Co or y.numel() != N * 4 * IH * IW * Co:
        raise ValueError("convT shapes mismatch")
    if bias is not None:
        _contig(bias, "bias")
        if bias.numel() != Co:
            raise ValueError("bias size mismatch")
    _call("dv3_convT_s2_fwd", _ptr(x), _ptr(wp), _ptr(bias), float(out_add), _ptr(y), N, IH, IW, Ci, Co,
          int(accumulate), _stream(), key=f"convT_s2_kernel<Co={Co}>" + (f"[N{N} {IH}x{IW} Ci{Ci}]" if PROFILE.by_shape else ""), flops=2.0 * N * IH * IW * 16 * Ci * Co,
          nbytes=4.0 * (x.numel() + y.numel() + wp.numel()))
    return y


_WGRAD_SCRATCH = {}
_WGRAD_TILE = _dev.flag("DV3_WGRAD_TILE", True)


def conv_s2_wgrad(coarse, fine, dw):
    """dw[Cc,Cf,4,4] += grad.  coarse [N,H/2,W/2,Cc], fine [N,H,W,Cf] (NHWC)."""
    _contig(coarse, "coarse"), _contig(fine, "fine"), _contig(dw, "dw")
    if fine.dim() != 4 or coarse.dim() != 4:
        raise ValueError("wgrad operands must be NHWC 4-D")
    N, H, W, Cf = fine.shape
    Cc = coarse.shape[3]
    if coarse.shape[0] != N or coarse.shape[1] * 2 != H or coarse.shape[2] * 2 != W or dw.numel() != 16 * Cc * Cf:
        raise ValueError("wgrad shapes mismatch")
    flops = 2.0 * N * (H // 2) * (W // 2) * 16 * Cf * Cc
    n_part = _lib.load().dv3_conv_s2_wgrad_tile_scratch(N, H, W, Cf, Cc) if _WGRAD_TILE else 0
    if n_part > 0:
        # narrow layer pair next to the image layers: tile walk + two-stage reduction (csrc/conv.hip)
        key = ("tile", str(dw.device), n_part, _stream())  # per launch stream: graph branches must not share it
        part = _WGRAD_SCRATCH.get(key)
        if part is None:
            part = torch.empty(n_part, dtype=F32, device=dw.device)
            _WGRAD_SCRATCH[key] = part
        _call("dv3_conv_s2_wgrad_tile", _ptr(coarse), _ptr(fine), _ptr(part), _ptr(dw), N, H, W, Cf, Cc, _stream(),
              key="conv_wgrad_tile_kernel<32,64>" + (f"[N{N} {H}x{W} Cf{Cf} Cc{Cc}]" if PROFILE.by_shape else ""),
              flops=flops, nbytes=4.0 * (coarse.numel() + fine.numel() + dw.numel()))
        return dw
    key = (str(dw.device), dw.data_ptr())  # one zero-initialised packed scratch per weight tensor
    scratch = _WGRAD_SCRATCH.get(key)
    if scratch is None or scratch.numel() != dw.numel():
        scratch = torch.zeros(dw.numel(), dtype=F32, device=dw.device)
        _WGRAD_SCRATCH[key] = scratch
    c3_tile = Cf == 3 and Cc in (32, 96) and (H // 2) % 16 == 0 and (W // 2) % 16 == 0  # csrc/conv.hip dv3_conv_s2_wgrad
    name = (f"conv_wgrad_c3_kernel<{Cc}>" if c3_tile else
            f"conv_wgrad_kernel<{'128x128x16' if Cc >= 128 else '32x64x64s2' if Cc <= 32 else '64x64x32'}"
            f"{',c3' if Cf == 3 else ''}>")
    _call("dv3_conv_s2_wgrad", _ptr(coarse), _ptr(fine), _ptr(scratch), _ptr(dw), N, H, W, Cf, Cc, _stream(),
          key=name + (f"[N{N} {H}x{W} Cf{Cf} Cc{Cc}]" if PROFILE.by_shape else ""),
          flops=2.0 * N * (H // 2) * (W // 2) * 16 * Cf * Cc,
          nbytes=4.0 * (coarse.numel() + fine.numel() + dw.numel()))
    return dw


def sumsq_accumulate(x, out):
    _contig(x, "x"), _contig(out, "out")
    _call("dv3_sumsq_accumulate", _ptr(x), x.numel(), _ptr(out), _stream())


def sumsq_ordered(x, out, partial):
    """out[0] += sum(x^2), summed in a fixed order (partial: float32 scratch of >= 1 element, up to 1024 are used)."""
    _contig(x, "x"), _contig(out, "out"), _contig(partial, "partial")
    _call("dv3_sumsq_ordered", _ptr(x), x.numel(), _ptr(out), _ptr(partial), int(partial.numel()), _stream())


def adam_step(param, grad, exp_avg, exp_avg_sq, state, *, lr, beta1=0.9, beta2=0.999, eps, clip, weight_decay=0.0,
              grad_scale=1.0):
    for t, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"),
                  (state, "state")):
        _contig(t, nm)
    n = param.numel()
    if grad.numel() != n or exp_avg.numel() != n or exp_avg_sq.numel() != n or state.numel() < 3:
        raise ValueError("size mismatch")
    _call("dv3_adam_step", _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), n, _ptr(state), float(lr),
          float(beta1), float(beta2), float(eps), float(clip or 0.0), float(weight_decay or 0.0), float(grad_scale),
          _stream())


def axpby(x, y, a, b):
    _contig(x, "x"), _contig(y, "y")
    if x.numel() != y.numel():
        raise ValueError("size mismatch")
    _call("dv3_axpby", _ptr(x), _ptr(y), x.numel(), float(a), float(b), _stream())
    return y


def rng_advance(rng_state, increment):
    _contig(rng_state, "rng_state", torch.int64)
    _call("dv3_rng_advance", _ptr(rng_state), int(increment), _stream())


def symlog(x, y):
    _contig(x, "x"), _contig(y, "y")
    if x.numel() != y.numel():
        raise ValueError("size mismatch")
    _call("dv3_symlog", _ptr(x), _ptr(y), x.numel(), _stream())
    return y


def symlog_mse(mode, x, loss, dmode=None, *, upstream=0.0):
    _contig(mode, "mode"), _contig(x, "x"), _contig(loss, "loss")
    W = mode.shape[-1]
    R = mode.numel() // W
    if x.shape != mode.shape or loss.numel() != R:
        raise ValueError("size mismatch")
    if dmode is not None:
        _contig(dmode, "dmode")
        if dmode.numel() != mode.numel():
            raise ValueError("dmode size mismatch")
    _call("dv3_symlog_mse", _ptr(mode), _ptr(x), _ptr(loss), _ptr(dmode), R, W, float(upstream), _stream())


def actor_normal_fwd(mean_raw, std_raw, eps=None, action=None, entropy=None, *, min_std=0.1, max_std=1.0):
    _contig(mean_raw, "mean_raw"), _contig(std_raw, "std_raw")
    A = mean_raw.shape[-1]
    M = mean_raw.numel() // A
    if std_raw.shape != mean_raw.shape:
        raise ValueError("std shape mismatch")
    for t, nm in ((eps, "eps"), (action, "action")):
        if t is not None:
            _contig(t, nm)
            if t.numel() != M * A:
                raise ValueError(nm + " size mismatch")
    if entropy is not None:
        _contig(entropy, "entropy")
        if entropy.numel() != M:
            raise ValueError("entropy size mismatch")
    _call("dv3_actor_normal_fwd", _ptr(mean_raw), _ptr(std_raw), _ptr(eps), _ptr(action), _ptr(entropy), M, A,
          float(min_std), float(max_std), _stream())


def actor_normal_logp(mean_raw, std_raw, action, logp, *, min_std=0.1, max_std=1.0):
    for t, nm in ((mean_raw, "mean_raw"), (std_raw, "std_raw"), (action, "action"), (logp, "logp")):
        _contig(t, nm)
    A = mean_raw.shape[-1]
    M = mean_raw.numel() // A
    if std_raw.numel() != M * A or action.numel() != M * A or logp.numel() != M:
        raise ValueError("size mismatch")
    _call("dv3_actor_normal_logp", _ptr(mean_raw), _ptr(std_raw), _ptr(action), _ptr(logp), M, A, float(min_std),
          float(max_std), _stream())


def actor_normal_bwd(mean_raw, std_raw, dmean_raw, dstd_raw, *, eps=None, action=None, daction=None, dent=None,
                     dlogp=None, min_std=0.1, max_std=1.0, logp_of_sample=False):
    """logp_of_sample: dlogp is on the log-prob of the action rsampled from this (mean, std) through eps (the path
    through the action is differentiated, models.py:667); False: the action is a constant."""
    for t, nm in ((mean_raw, "mean_raw"), (std_raw, "std_raw"), (dmean_raw, "dmean_raw"), (dstd_raw, "dstd_raw")):
        _contig(t, nm)
    A = mean_raw.shape[-1]
    M = mean_raw.numel() // A
    for t, nm, n in ((eps, "eps", M * A), (action, "action", M * A), (daction, "daction", M * A), (dent, "dent", M),
                     (dlogp, "dlogp", M)):
        if t is not None:
            _contig(t, nm)
            if t.numel() != n:
                raise ValueError(nm + " size mismatch")
    if std_raw.numel() != M * A or dmean_raw.numel() != M * A or dstd_raw.numel() != M * A:
        raise ValueError("size mismatch")
    _call("dv3_actor_normal_bwd", _ptr(mean_raw), _ptr(std_raw), _ptr(eps), _ptr(action), _ptr(daction), _ptr(dent),
          _ptr(dlogp), _ptr(dmean_raw), _ptr(dstd_raw), M, A, float(min_std), float(max_std), int(logp_of_sample),
          _stream())


def lambda_return_fwd(reward, value, cont_logit, target, weights, disc=None, *, gamma, lam):
    for t, nm in ((reward, "reward"), (value, "value"), (cont_logit, "cont_logit"), (target, "target"),
                  (weights, "weights")):
        _contig(t, nm)
    H = reward.shape[0]
    N = reward.numel() // H
    if value.numel() != H * N or cont_logit.numel() != H * N or weights.numel() != H * N or target.numel() != (H - 1) * N:
        raise ValueError("size mismatch")
    if disc is not None:
        _contig(disc, "disc")
        if disc.numel() != H * N:
            raise ValueError("disc size mismatch")
    _call("dv3_lambda_return_fwd", _ptr(reward), _ptr(value), _ptr(cont_logit), _ptr(target), _ptr(weights),
          _ptr(disc), H, N, float(gamma), float(lam), _stream())


def lambda_return_bwd(dtarget, value, cont_logit, target, dreward, dcont_logit, *, gamma, lam):
    for t, nm in ((dtarget, "dtarget"), (value, "value"), (cont_logit, "cont_logit"), (target, "target"),
                  (dreward, "dreward"), (dcont_logit, "dcont_logit")):
        _contig(t, nm)
    H = value.shape[0]
    N = value.numel() // H
    if (dtarget.numel() != (H - 1) * N or target.numel() != (H - 1) * N or cont_logit.numel() != H * N
            or dreward.numel() != H * N or dcont_logit.numel() != H * N):
        raise ValueError("size mismatch")
    _call("dv3_lambda_return_bwd", _ptr(dtarget), _ptr(value), _ptr(cont_logit), _ptr(target), _ptr(dreward),
          _ptr(dcont_logit), H, N, float(gamma), float(lam), _stream())


def reset_blend(x, init, is_first, out):
    B, n, ldo = _rows2d(out, "out")
    ldx = 0
    if x is not None:
        Bx, nx, ldx = _rows2d(x, "x")
        if (Bx, nx) != (B, n):
            raise ValueError("x shape mismatch")
    if init is not None:
        _contig(init, "init")
        if init.numel() != n:
            raise ValueError("init size mismatch")
    _contig(is_first, "is_first")
    if is_first.numel() != B:
        raise ValueError("is_first size mismatch")
    _call("dv3_reset_blend", _ptr(x), ldx, _ptr(init), _ptr(is_first), _ptr(out), ldo, B, n, _stream())
    return out


def reset_blend_bwd(dout, is_first, dx=None, dinit=None):
    B, n, ldo = _rows2d(dout, "dout")
    ldx = 0
    if dx is not None:
        Bx, nx, ldx = _rows2d(dx, "dx")
        if (Bx, nx) != (B, n):
            raise ValueError("dx shape mismatch")
    if dinit is not None:
        _contig(dinit, "dinit")
        if dinit.numel() != n:
            raise ValueError("dinit size mismatch")
    _contig(is_first, "is_first")
    if is_first.numel() != B:
        raise ValueError("is_first size mismatch")
    _call("dv3_reset_blend_bwd", _ptr(dout), ldo, _ptr(is_first), _ptr(dx), ldx, _ptr(dinit), B, n, _stream())


def fill_normal(out, rng: RngStream):
    _contig(out, "out")
    _call("dv3_fill_normal", _ptr(out), out.numel(), _ptr(rng.state), int(rng.take(out.numel())), _stream())
    return out


def obs_blend(prev_stoch, init_stoch, prev_deter, init_deter, action, is_first, out_stoch, out_deter, out_action):
    B, SD = out_stoch.shape
    De, A = out_deter.shape[1], out_action.shape[1]
    for t, nm, n in ((prev_stoch, "prev_stoch", B * SD), (init_stoch, "init_stoch", SD), (prev_deter, "prev_deter", B * De),
                     (init_deter, "init_deter", De), (action, "action", B * A), (is_first, "is_first", B),
                     (out_stoch, "out_stoch", B * SD), (out_deter, "out_deter", B * De), (out_action, "out_action", B * A)):
        if t is None:
            continue
        _contig(t, nm)
        if t.numel() != n:
            raise ValueError(nm + " size mismatch")
    _call("dv3_obs_blend", _ptr(prev_stoch), _ptr(init_stoch), _ptr(prev_deter), _ptr(init_deter), _ptr(action),
          _ptr(is_first), _ptr(out_stoch), _ptr(out_deter), _ptr(out_action), B, SD, De, A, _stream())


def obs_blend_bwd(dsin, ddin, is_first, gs_prev, gd_prev, dstoch0, ddeter0):
    B, SD, ld_s = _rows2d(dsin, "dsin")
    Bd, De, ld_d = _rows2d(ddin, "ddin")
    if Bd != B:
        raise ValueError("dsin/ddin row mismatch")
    for t, nm, n in ((is_first, "is_first", B), (gs_prev, "gs_prev", B * SD), (gd_prev, "gd_prev", B * De),
                     (dstoch0, "dstoch0", SD), (ddeter0, "ddeter0", De)):
        if t is None:
            continue
        _contig(t, nm)
        if t.numel() != n:
            raise ValueError(nm + " size mismatch")
    _call("dv3_obs_blend_bwd", _ptr(dsin), ld_s, _ptr(ddin), ld_d, _ptr(is_first), _ptr(gs_prev), _ptr(gd_prev),
          _ptr(dstoch0), _ptr(ddeter0), B, SD, De, _stream())


def obs_carry_st_bwd(dsin, ddin, is_first, gs_prev, gd_prev, dstoch0, ddeter0, logit_prev, dlogit_prev, *, unimix=0.01):
    """obs_blend_bwd (step t) + onehot_st_bwd(accumulate=True) (step t-1) in one launch; logit_prev [B,S,D]."""
    B, SD, ld_s = _rows2d(dsin, "dsin")
    Bd, De, ld_d = _rows2d(ddin, "ddin")
    if logit_prev.dim() != 3 or logit_prev.shape[0] != B or logit_prev.shape[1] * logit_prev.shape[2] != SD or Bd != B:
        raise ValueError("obs_carry_st_bwd shapes mismatch")
    S, D = logit_prev.shape[1], logit_prev.shape[2]
    for t, nm, n in ((is_first, "is_first", B), (gs_prev, "gs_prev", B * SD), (gd_prev, "gd_prev", B * De),
                     (dstoch0, "dstoch0", SD), (ddeter0, "ddeter0", De), (logit_prev, "logit_prev", B * SD),
                     (dlogit_prev, "dlogit_prev", B * SD)):
        _contig(t, nm)
        if t.numel() != n:
            raise ValueError(nm + " size mismatch")
    _call("dv3_obs_carry_st_bwd", _ptr(dsin), ld_s, _ptr(ddin), ld_d, _ptr(is_first), _ptr(gs_prev), _ptr(gd_prev),
          _ptr(dstoch0), _ptr(ddeter0), _ptr(logit_prev), _ptr(dlogit_prev), B, S, D, De, float(unimix), _stream())


def dot_accumulate(x, out, *, w=None, clip_min=None, scale=1.0):
    """out[0] += scale * sum(max(x, clip_min) * w)."""
    _contig(x, "x"), _contig(out, "out")
    if w is not None:
        _contig(w, "w")
        if w.numel() != x.numel():
            raise ValueError("w size mismatch")
    _call("dv3_dot_accumulate", _ptr(x), _ptr(w), x.numel(), _ptr(out), int(clip_min is not None),
          float(clip_min or 0.0), float(scale), _stream())


def actor_loss(target, value, weights, entropy, ema_vals, loss_out, dent, *, dtarget=None, logp=None, dlogp=None,
               entropy_coef, reinforce=False, mode=None, mix=0.0):
    """mode: 0 'dynamics', 1 'reinforce', 2 'both' (default: reinforce flag)."""
    mode = int(bool(reinforce)) if mode is None else int(mode)
    H = value.shape[0]
    N = value.numel() // H
    for t, nm, n in ((target, "target", (H - 1) * N), (value, "value", H * N), (weights, "weights", H * N),
                     (entropy, "entropy", H * N), (dent, "dent", H * N), (ema_vals, "ema_vals", 2),
                     (loss_out, "loss_out", None), (dtarget, "dtarget", (H - 1) * N), (logp, "logp", H * N),
                     (dlogp, "dlogp", H * N)):
        if t is None:
            continue
        _contig(t, nm)
        if n is not None and t.numel() != n:
            raise ValueError(f"{nm} size mismatch")
    _call("dv3_actor_loss", _ptr(target), _ptr(value), _ptr(weights), _ptr(entropy), _ptr(logp), _ptr(ema_vals),
          _ptr(loss_out), _ptr(dtarget), _ptr(dlogp), _ptr(dent), H, N, float(entropy_coef), mode, float(mix), _stream())


def scale_neg(w, out, s):
    _contig(w, "w"), _contig(out, "out")
    if w.numel() != out.numel():
        raise ValueError("size mismatch")
    _call("dv3_scale_neg", _ptr(w), _ptr(out), w.numel(), float(s), _stream())
    return out


C3_WIDTHS = (32, 96)


def conv_s2_c3_fwd(x, w, y, *, CW, accumulate=False):
    """x [N,H,W,3] -> y [N,H/2,W/2,CW]; w = Conv2d weight [CW,3,4,4] (reference layout, unpacked)."""
    _contig(x, "x"), _contig(w, "w"), _contig(y, "y")
    if x.dim() != 4 or x.shape[3] != 3 or tuple(w.shape) != (CW, 3, 4, 4) or CW not in C3_WIDTHS:
        raise ValueError("conv_s2_c3 shapes")
    N, H, W = x.shape[0], x.shape[1], x.shape[2]
    if H % 2 or W % 2 or y.numel() != N * (H // 2) * (W // 2) * CW:
        raise ValueError("conv_s2_c3 output size")
    _call("dv3_conv_s2_c3_fwd", _ptr(x), _ptr(w), _ptr(y), N, H, W, CW, int(accumulate), _stream(),
          key="conv_s2_c3_kernel", nbytes=4.0 * (x.numel() + y.numel()))
    return y


def convT_s2_c3_fwd(x, w, y, *, CW, bias=None, out_add=0.0, accumulate=False):
    """x [N,IH,IW,CW] -> y [N,2IH,2IW,3]; w = ConvTranspose2d weight [CW,3,4,4] (reference layout)."""
    _contig(x, "x"), _contig(w, "w"), _contig(y, "y")
    if x.dim() != 4 or x.shape[3] != CW or tuple(w.shape) != (CW, 3, 4, 4) or CW not in C3_WIDTHS:
        raise ValueError("convT_s2_c3 shapes")
    N, IH, IW = x.shape[0], x.shape[1], x.shape[2]
    if y.numel() != N * 4 * IH * IW * 3:
        raise ValueError("convT_s2_c3 output size")
    if bias is not None:
        _contig(bias, "bias")
        if bias.numel() != 3:
            raise ValueError("bias size")
    _call("dv3_convT_s2_c3_fwd", _ptr(x), _ptr(w), _ptr(bias), float(out_add), _ptr(y), N, IH, IW, CW,
          int(accumulate), _stream(), key="convT_s2_c3_kernel", nbytes=4.0 * (x.numel() + y.numel()))
    return y


# ---------------------------------------------------------------------------------------------
# row-fused layers of the imagination step (csrc/fusedops.hip)
# ---------------------------------------------------------------------------------------------
def transpose2d_many(pairs):
    """dst[c][r] = src[r][c] for every (src, dst) pair, 12 per launch (weights re-packed once per update)."""
    import numpy as np

    pairs = list(pairs)
    for k0 in range(0, len(pairs), 12):
        chunk = pairs[k0:k0 + 12]
        jobs = np.empty((len(chunk), 6), dtype=np.uint64)
        for k, (src, dst) in enumerate(chunk):
            R, C, lds = _rows2d(src, "src")
            Cd, Rd, ldd = _rows2d(dst, "dst")
            if (Rd, Cd) != (R, C):
                raise ValueError(f"transpose2d_many: src {tuple(src.shape)} dst {tuple(dst.shape)}")
            jobs[k] = (src.data_ptr(), dst.data_ptr(), lds, ldd, R, C)
        _call("dv3_transpose2d_many", len(chunk), jobs.ctypes.data, _stream(), key="dv3_transpose2d")


def transpose2d(src, dst):
    """dst [C,R] = src [R,C]^T (src may be a column slice of a wider matrix)."""
    R, C, lds = _rows2d(src, "src")
    Rd, Cd, ldd = _rows2d(dst, "dst")
    if (Rd, Cd) != (C, R):
        raise ValueError(f"transpose2d: src {tuple(src.shape)} dst {tuple(dst.shape)}")
    _call("dv3_transpose2d", _ptr(src), lds, R, C, _ptr(dst), ldd, _stream())
    return dst


def onehot_to_idx(onehot, idx):
    D = onehot.shape[-1]
    R = _groups(onehot, "onehot", D)
    _contig(idx, "idx", torch.int32)
    if idx.numel() != R:
        raise ValueError("idx size mismatch")
    _call("dv3_onehot_to_idx", _ptr(onehot), _ptr(idx), R, D, _stream())
    return idx


def onehot_linear_ln(idx, D, WT, pre, *, x2=None, base=None, gamma=None, beta=None, y=None, mean=None, rstd=None,
                     act=True):
    """pre [M,N] = base + gather of WT rows by idx [M,S] (+ x2 [M,A2] @ WT[S*D:]); y = act(LN(pre)) when y is given.
    WT [S*D + A2, N] is the transposed Linear weight."""
    _contig(idx, "idx", torch.int32)
    if idx.dim() != 2:
        raise ValueError("idx must be [M,S]")
    M, S = idx.shape
    Kw, N, ldw = _rows2d(WT, "WT")
    Mp, Np, ldpre = _rows2d(pre, "pre")
    A2, ldx2 = 0, 0
    if x2 is not None:
        Mx, A2, ldx2 = _rows2d(x2, "x2")
        if Mx != M:
            raise ValueError("x2 rows mismatch")
    if (Mp, Np) != (M, N) or Kw != S * D + A2 or S > 64:
        raise ValueError(f"onehot_linear_ln: idx {tuple(idx.shape)} D {D} A2 {A2} WT {tuple(WT.shape)} pre {tuple(pre.shape)}")
    ldbase = 0
    if base is not None:
        Mb, Nb, ldbase = _rows2d(base, "base")
        if (Mb, Nb) != (M, N):
            raise ValueError("base shape mismatch")
    ldy = 0
    if y is not None:
        My, Ny, ldy = _rows2d(y, "y")
        _contig(gamma, "gamma"), _contig(beta, "beta")
        if (My, Ny) != (M, N) or gamma.numel() != N or beta.numel() != N:
            raise ValueError("y / LN affine shape mismatch")
        for t, nm in ((mean, "mean"), (rstd, "rstd")):
            if t is not None:
                _contig(t, nm)
                if t.numel() != M:
                    raise ValueError(nm + " size mismatch")
    _call("dv3_onehot_linear_ln_fwd", _ptr(idx), S, int(D), _ptr(x2), ldx2, A2, _ptr(WT), ldw, _ptr(base), ldbase,
          _ptr(pre), ldpre, _ptr(gamma), _ptr(beta), _ptr(y), ldy, _ptr(mean), _ptr(rstd), M, N, int(act), _stream(),
          key="dv3_onehot_linear_ln_fwd" + (f"[{M}x{N},S={S}]" if PROFILE.by_shape else ""),
          flops=2.0 * M * N * (S * D + A2), nbytes=4.0 * M * N * (S + 2))
    return y if y is not None else pre


def sample_linear_ln_ok(S, D, N) -> bool:
    """Shapes the fused posterior-sample + img_in launch takes (ops.onehot_sample_linear_ln)."""
    return D == 32 and S <= 32 and N % 256 == 0 and N <= 1024


def onehot_sample_linear_ln(logit, out, *, next_first, init, init_idx, next_out, next_idx, WT, x2, pre, gamma, beta, y,
                            mean, rstd, noise=None, rng=None, idx=None, forced=None, flips=None, unimix=0.01,
                            mode=False, act=True):
    """ops.onehot_sample(logit [M,S,32], out, next_blend=(next_first, init, next_out, init_idx, next_idx)) followed by
    ops.onehot_linear_ln(next_idx, 32, WT, pre, x2=x2, ..., y=y) in one launch (observe scan: the sample of step t and
    the img_in layer of step t+1)."""
    if logit.dim() != 3 or logit.shape[2] != 32:
        raise ValueError("logit must be [M,S,32]")
    M, S, D = logit.shape
    Kw, N, ldw = _rows2d(WT, "WT")
    Mp, Np, ldpre = _rows2d(pre, "pre")
    My, Ny, ldy = _rows2d(y, "y")
    Mx, A2, ldx2 = _rows2d(x2, "x2") if x2 is not None else (M, 0, 0)
    if not sample_linear_ln_ok(S, D, N) or (Mp, Np) != (M, N) or (My, Ny) != (M, N) or Mx != M or Kw != S * D + A2:
        raise ValueError("onehot_sample_linear_ln shapes mismatch")
    R = M * S
    for t, nm, n, dt in ((logit, "logit", R * D, F32), (out, "out", R * D, F32), (noise, "noise", R * D, F32),
                         (idx, "idx", R, torch.int32), (forced, "forced", R, torch.int32),
                         (flips, "flips", 1, torch.int32), (next_first, "next_first", M, F32),
                         (init, "init", S * D, F32), (init_idx, "init_idx", S, torch.int32),
                         (next_out, "next_out", R * D, F32), (next_idx, "next_idx", R, torch.int32),
                         (gamma, "gamma", N, F32), (beta, "beta", N, F32), (mean, "mean", M, F32),
                         (rstd, "rstd", M, F32)):
        if t is None:
            continue
        _contig(t, nm, dt)
        if t.numel() != n:
            raise ValueError(nm + " size mismatch")
    rng_state, rng_off = None, 0
    if not mode and noise is None:
        if rng is None:
            raise ValueError("sampling needs noise or an RngStream")
        rng_state, rng_off = rng.state, rng.take(R * D)
    _call("dv3_onehot_sample_linear_ln_fwd", _ptr(logit), _ptr(noise), _ptr(rng_state), int(rng_off), _ptr(out),
          _ptr(idx), _ptr(forced), _ptr(flips), float(unimix), int(mode), _ptr(next_first), _ptr(init), _ptr(init_idx),
          _ptr(next_out), _ptr(next_idx), S, _ptr(x2), ldx2, A2, _ptr(WT), ldw, _ptr(pre), ldpre, _ptr(gamma),
          _ptr(beta), _ptr(y), ldy, _ptr(mean), _ptr(rstd), M, N, int(act), _stream(),
          key="dv3_onehot_sample_linear_ln_fwd" + (f"[{M}x{N},S={S}]" if PROFILE.by_shape else ""),
          flops=2.0 * M * N * (S * D + A2), nbytes=4.0 * M * N * (S + 2))
    return out


def actor_head(pre, gamma, beta, y, mean, rstd, Wm, bm, Ws, bs, out_m, out_s, action, entropy, *, noise=None,
               rng=None, eps_out=None, act_idx=None, forced=None, flips=None, min_std=0.1, max_std=1.0, unimix=0.01,
               onehot=False):
    """Last trunk LayerNorm+SiLU, the heads, the action sample and the entropy of the actor in one launch."""
    M, U, ldpre = _rows2d(pre, "pre")
    My, Uy, ldy = _rows2d(y, "y")
    A = Wm.shape[0]
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (Wm, "Wm"), (bm, "bm"), (out_m, "out_m"), (action, "action")):
        _contig(t, nm)
    if (My, Uy) != (M, U) or gamma.numel() != U or tuple(Wm.shape) != (A, U) or bm.numel() != A \
            or out_m.numel() != M * A or action.numel() != M * A or U > 1024 or A > 64:
        raise ValueError("actor_head shapes mismatch")
    if not onehot:
        for t, nm in ((Ws, "Ws"), (bs, "bs"), (out_s, "out_s")):
            _contig(t, nm)
        if tuple(Ws.shape) != (A, U) or bs.numel() != A or out_s.numel() != M * A:
            raise ValueError("actor_head std head shapes mismatch")
    else:
        Ws = bs = out_s = None
    for t, nm, n in ((mean, "mean", M), (rstd, "rstd", M), (entropy, "entropy", M), (noise, "noise", M * A),
                     (eps_out, "eps_out", M * A)):
        if t is not None:
            _contig(t, nm)
            if t.numel() != n:
                raise ValueError(nm + " size mismatch")
    for t, nm in ((act_idx, "act_idx"), (forced, "forced")):
        if t is not None:
            _contig(t, nm, torch.int32)
            if t.numel() != M:
                raise ValueError(nm + " size mismatch")
    if flips is not None:
        _contig(flips, "flips", torch.int32)
    rng_state, rng_off = None, 0
    if noise is None:
        if rng is None:
            raise ValueError("actor_head needs noise or an RngStream")
        rng_state, rng_off = rng.state, rng.take(M * A)
    _call("dv3_actor_head_fwd", _ptr(pre), ldpre, _ptr(gamma), _ptr(beta), _ptr(y), ldy, _ptr(mean), _ptr(rstd), _ptr(Wm),
          _ptr(bm), _ptr(Ws), _ptr(bs), _ptr(out_m), _ptr(out_s), _ptr(noise), _ptr(rng_state), int(rng_off),
          _ptr(eps_out), _ptr(action), _ptr(entropy), _ptr(act_idx), _ptr(forced), _ptr(flips), M, U, A,
          float(min_std), float(max_std), float(unimix), int(onehot), _stream(),
          flops=2.0 * M * U * A * (1 if onehot else 2))


def gemm_sample_ok(M, N, D, A=None) -> bool:
    """True when ops.gemm_sample applies: groups of 32 classes, whole groups per 64-column tile, and an output
    size for which dv3_gemm_f32 would pick the register-direct kernel anyway."""
    return D == 32 and N % 64 == 0 and M > 32 and -(-M // 64) * (N // 64) <= 512


def gemm_sample(A, B, logit, onehot, *, bias=None, noise=None, rng=None, idx=None, forced=None, flips=None,
                unimix=0.01, mode=False, ln=None):
    """logit [M,N] = A @ B^T + bias and, in the same launch, the one-hot sample of every group of 32 logits
    (onehot [M,N]; idx/forced int32 [M*N/32]) -- ops.gemm followed by ops.onehot_sample, fused.  ln: the
    LayerNorm + SiLU of the layer that produced A, applied on the fly (A = its pre-activations)."""
    M, K, lda = _rows2d(A, "A")
    N, Kb, ldb = _rows2d(B, "B")
    Mc, Nc, ldc = _rows2d(logit, "logit")
    _contig(onehot, "onehot")
    if Kb != K or (Mc, Nc) != (M, N) or onehot.numel() != M * N or N % 64:
        raise ValueError("gemm_sample shapes mismatch")
    R = M * N // 32
    if bias is not None:
        _contig(bias, "bias")
        if bias.numel() != N:
            raise ValueError("bias size mismatch")
    if noise is not None:
        _contig(noise, "noise")
        if noise.numel() != M * N:
            raise ValueError("noise size mismatch")
    for t, nm in ((idx, "idx"), (forced, "forced")):
        if t is not None:
            _contig(t, nm, torch.int32)
            if t.numel() != R:
                raise ValueError(nm + " size mismatch")
    if flips is not None:
        _contig(flips, "flips", torch.int32)
    rng_state, rng_off = None, 0
    if not mode and noise is None:
        if rng is None:
            raise ValueError("sampling needs noise or an RngStream")
        rng_state, rng_off = rng.state, rng.take(M * N)
    lg = lb = lm = lr = None
    if ln is not None:  # (gamma [K], beta [K], mean_out [M] | None, rstd_out [M] | None): A = pre-activations
        lg, lb, lm, lr = ln
        _contig(lg, "ln gamma"), _contig(lb, "ln beta")
        if lg.numel() != K or lb.numel() != K or K % 4 or lda % 4:
            raise ValueError("gemm_sample ln shapes mismatch")
        for t, nm in ((lm, "ln mean"), (lr, "ln rstd")):
            if t is not None:
                _contig(t, nm)
                if t.numel() != M:
                    raise ValueError(nm + " size mismatch")
    l16 = (ln is None and pick_gemm_tile(M, N) >= 11
           and l16_ok(A, None, B, False, True, K, K, lda, 0, ldb))  # profile key only: the library decides the same way
    _call("dv3_gemm_sample_f32", M, N, K, _ptr(A), lda, 0, 0, 0, _ptr(B), ldb, _ptr(logit), ldc, _ptr(bias), _ptr(noise),
          _ptr(rng_state), int(rng_off), _ptr(onehot), _ptr(idx), _ptr(forced), _ptr(flips), float(unimix), int(mode),
          _ptr(lg), _ptr(lb), _ptr(lm), _ptr(lr),
          _stream(), key=("gemm_kernel<l16+sample,tA=0,tB=1>" if l16 else "gemm_kernel<direct32x64+sample,tA=0,tB=1>")
          + (f"[{M}x{N}x{K}]" if PROFILE.by_shape else ""),
          flops=2.0 * M * N * K, nbytes=4.0 * (M * K + N * K + 2 * M * N))
    return onehot


def quantile2_ema(x, q0, q1, ema=None, alpha=0.0, out_q=None):
    """Quantiles q0, q1 of all elements of x (torch.quantile, linear interpolation) by exact radix selection; with
    ema [2]: ema <- alpha*q + (1-alpha)*ema in the same launch (models.RewardEMA)."""
    _contig(x, "x")
    for t, nm in ((ema, "ema"), (out_q, "out_q")):
        if t is not None:
            _contig(t, nm)
            if t.numel() != 2:
                raise ValueError(nm + " must hold 2 floats")
    if ema is None and out_q is None:
        raise ValueError("quantile2_ema: nothing to write")
    _call("dv3_quantile2_ema", _ptr(x), x.numel(), float(q0), float(q1), _ptr(ema), float(alpha), _ptr(out_q), _stream())
    return out_q if out_q is not None else ema

