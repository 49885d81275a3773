// fp32 MFMA tile engine for gfx950 (CDNA4), shared by the dense GEMM and the implicit-GEMM
// convolution kernels.
//
// Why fp32 MFMA: the parity bar is fp32 1e-4 through 64 sampled recurrent steps, so inputs
// cannot be rounded to bf16.  v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain at the
// fp32 vector peak (157 TFLOP/s chip-wide) while leaving the VALU free for epilogues.
//
// Structure (one workgroup = 4 waves = one BM x BN tile of C):
//   * LDS image is k-major: As[k][m], Bs[k][n] (row stride +4 floats).  A 32x32x2 MFMA wants
//     A[i][k0+h] / B[k0+h][j] for lane (i = l&31, h = l>>5): one conflict-free ds_read_b32 each
//     (32 consecutive floats per half-wave).
//   * global -> registers -> LDS staging, software-pipelined: the loads of K-tile t+1 are issued
//     before the MFMAs of tile t and written to the other LDS buffer after them; one barrier per
//     K-tile.
//   * Loaders are functors, so "A" can be a dense matrix in either orientation, two K-segments
//     (the reference's torch.cat([...], -1) in front of every Linear), or an im2col gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dv3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// 4-byte aligned float4: gfx950 global loads only need dword alignment
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kThreads = 256;

// Bijective XCD-aware remap of a linear workgroup id (blocks b and b+8 share an XCD / L2, so
// give each XCD a contiguous chunk of tiles).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ---------------------------------------------------------------------------------------------
// Dense operand loader.  The operand is a logical [R x K] matrix (R = rows of the tile
// dimension, i.e. M for A and N for B).
//   KCONTIG = true : element (r,k) at p[r*ld + k]        (x[M][K], W[N][K])
//   KCONTIG = false: element (r,k) at p[k*ld + r]        (dY^T for wgrad, W as [K][N] for dgrad)
// Optional second K-segment (KCONTIG only): k >= K1 reads p2[r*ld2 + (k-K1)]; K1 % BK == 0.
// ---------------------------------------------------------------------------------------------
template <bool KCONTIG>
struct DenseOperand {
  const float* p;
  const float* p2;
  long ld, ld2;
  int R, K, K1;
  bool vec_ok;  // ld % 4 == 0-independent: only dword alignment is needed; false forces scalar
};

template <int ROWS, int BK, bool KCONTIG>
struct DenseTile {
  // registers held per thread for one staged tile
  static constexpr int kVecs = ROWS * BK / 4 / kThreads;
  static_assert(ROWS * BK % (4 * kThreads) == 0, "tile must split into float4 per thread");
  // LDS row stride.  k-contiguous sources are transposed on the way in with ds_write_b32; the pad
  // spreads the BK/4 column chunks of a 32-lane group over the 32 banks (pad = 8/CH).  Row-contiguous
  // sources are written with ds_write_b128 and need a 16-byte multiple.
  static constexpr int LD = ROWS + (KCONTIG ? (BK == 16 ? 2 : 1) : 4);  // BK 16 -> +2, BK 32/64 -> +1 (<= 2-way)
  f32x4 v[kVecs];
  float mask[kVecs];  // 1 for rows inside the matrix, 0 past its edge (KCONTIG tiles; loop-invariant)

  __device__ __forceinline__ void init(const DenseOperand<KCONTIG>& op, int r0, int tid) {
    if constexpr (KCONTIG) {
      constexpr int RPP = kThreads / (BK / 4);
#pragma unroll
      for (int p = 0; p < kVecs; ++p) mask[p] = (r0 + p * RPP + tid / (BK / 4) < op.R) ? 1.f : 0.f;
    }
  }

  // number of leading K-tiles of [kbeg, kend) that need no k guard (segment edges sit on tile edges)
  __device__ __forceinline__ static int full_tiles(const DenseOperand<KCONTIG>& op, int kbeg, int kend) {
    if (kend > op.K) kend = op.K;
    return (kend > kbeg) ? (kend - kbeg) / BK : 0;
  }

  // Steady-state load: the tile's k range is entirely valid.  Rows past the edge are handled without
  // control flow (load from a safe address, select zero) so that the loop body stays straight-line and
  // the compiler keeps every global load of the tile in flight across the MFMAs.
  __device__ __forceinline__ void load_full(const DenseOperand<KCONTIG>& op, int r0, int k0, int tid) {
    if constexpr (KCONTIG) {
      constexpr int CH = BK / 4;
      constexpr int RPP = kThreads / CH;
      const float* base = op.p;
      long ld = op.ld;
      int kk0 = k0;
      if (k0 >= op.K1) { base = op.p2; ld = op.ld2; kk0 = k0 - op.K1; }
      const int c = tid % CH, rr = tid / CH;
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const int r = r0 + p * RPP + rr;
        // rows past the edge read a safe address; store() multiplies them by mask = 0.  (A multiply, not
        // a select or a branch: the load stays unconditional and nothing consumes it before the MFMAs,
        // so all of the tile's global loads stay in flight across them.)
        const float* src = (mask[p] != 0.f) ? base + (long)r * ld + (kk0 + 4 * c) : base;
        v[p] = *reinterpret_cast<const f32x4u*>(src);
      }
    } else {
      constexpr int CH = ROWS / 4;
      constexpr int KPP = kThreads / CH;
      static_assert(kThreads % CH == 0 && BK % KPP == 0, "bad tile");
      const int c = tid % CH, kr = tid / CH;
      const int r = r0 + 4 * c;
      const bool ok4 = r + 3 < op.R;
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const int k = k0 + p * KPP + kr;
        const float* q = op.p + (long)k * op.ld + r;
        if (ok4) {
          v[p] = *reinterpret_cast<const f32x4u*>(q);
        } else {
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (r + e < op.R) t[e] = q[e];
          v[p] = t;
        }
      }
    }
  }

  // Fully guarded load (the one partial K-tile at the end of a segment).
  __device__ __forceinline__ void load_tail(const DenseOperand<KCONTIG>& op, int r0, int k0, int tid) {
    if constexpr (KCONTIG) {
      constexpr int CH = BK / 4;
      constexpr int RPP = kThreads / CH;
      const float* base = op.p;
      long ld = op.ld;
      int kk0 = k0, kend = op.K1;
      if (k0 >= op.K1) { base = op.p2; ld = op.ld2; kk0 = k0 - op.K1; kend = op.K - op.K1; }
      const int c = tid % CH, rr = tid / CH;
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const int r = r0 + p * RPP + rr;
        const int k = kk0 + 4 * c;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (r < op.R) {
          const float* q = base + (long)r * ld + k;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < kend) t[e] = q[e];
        }
        v[p] = t;
      }
    } else {
      constexpr int CH = ROWS / 4;
      constexpr int KPP = kThreads / CH;
      const int c = tid % CH, kr = tid / CH;
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const int k = k0 + p * KPP + kr;
        const int r = r0 + 4 * c;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (k < op.K) {
          const float* q = op.p + (long)k * op.ld + r;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (r + e < op.R) t[e] = q[e];
        }
        v[p] = t;
      }
    }
  }

  // LDS image: s[k][row], row stride LD
  __device__ __forceinline__ void store(float* s, int tid) const {
    if constexpr (KCONTIG) {
      constexpr int CH = BK / 4;
      constexpr int RPP = kThreads / CH;
      const int c = tid % CH, rr = tid / CH;
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const int r = p * RPP + rr;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[(4 * c + e) * LD + r] = v[p][e] * mask[p];
      }
    } else {
      constexpr int CH = ROWS / 4;
      constexpr int KPP = kThreads / CH;
      const int c = tid % CH, kr = tid / CH;
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const int k = p * KPP + kr;
        *reinterpret_cast<f32x4*>(&s[k * LD + 4 * c]) = v[p];
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Tile shape: 4 waves as WM x WN, each wave TM x TN MFMA tiles of 32x32.
// ---------------------------------------------------------------------------------------------
// WK > 1: the waves also split K -- wave-group wk takes k-steps [wk*BK/2/WK, (wk+1)*BK/2/WK) of every staged
// tile and the partial accumulators are reduced through LDS at the end.  Twice the workgroups for the same
// output: fills the chip when M*N is small (M = 1024 rows of imagination against N = 512 columns).
template <int WM_, int WN_, int TM_, int TN_, int BK_, int WK_ = 1>
struct TileShape {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, BK = BK_, WK = WK_;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");
  static_assert((BK / 2) % WK == 0, "k-steps must split evenly over the wave-groups");
  static constexpr int lds_floats = 2 * BK * (BM + 4) + 2 * BK * (BN + 4);  // upper bound on any pad
};

// The K loop.  ATile/BTile: staged-tile types with init / full_tiles / load_full / load_tail / store.
// acc[tm][tn] on exit holds C(m0 + wm*TM*32 + tm*32 + row(reg,lane), n0 + ... + (lane&31)),
// row(reg, lane) = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//
// Shape of the loop: all K-tiles that need no k guard run in a branch-free steady state (global loads of
// tile t+1 issued, every LDS fragment of tile t read up front, MFMAs, LDS store, one barrier); the at most
// one guarded tail tile is peeled off behind it.
// A wave with a single 32x32 output tile would issue every MFMA into ONE accumulator: a dependent chain
// that exposes the MFMA result latency.  Such shapes alternate k-steps between two accumulators (summed
// once at the end), so consecutive MFMAs are independent.
template <class TS>
struct AccSplit {
  static constexpr int N = (TS::TM * TS::TN == 1) ? 2 : 1;
};

template <class TS, class ATile, class BTile>
__device__ __forceinline__ void mfma_tile_compute(const float* as, const float* bs, int h,
                                                  f32x16 (&acc)[AccSplit<TS>::N][TS::TM][TS::TN]) {
  constexpr int BK = TS::BK, LDA = ATile::LD, LDB = BTile::LD;
  constexpr int NS = BK / 2 / TS::WK;  // k-steps of this wave-group (as/bs already point at its first one)
  float av[NS][TS::TM], bv[NS][TS::TN];
  // every LDS fragment of the tile is requested before the first MFMA (pinned with sched_barrier: left
  // alone, hipcc sinks each pair of ds_reads next to its MFMAs and waits lgkmcnt(0) 8x per tile)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int a = 0; a < TS::TM; ++a) av[s][a] = as[(2 * s + h) * LDA + a * 32];
#pragma unroll
    for (int b = 0; b < TS::TN; ++b) bv[s][b] = bs[(2 * s + h) * LDB + b * 32];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int a = 0; a < TS::TM; ++a)
#pragma unroll
      for (int b = 0; b < TS::TN; ++b)
        acc[s % AccSplit<TS>::N][a][b] =
            __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][a], bv[s][b], acc[s % AccSplit<TS>::N][a][b], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}

template <class TS, class ATile, class BTile, class AOp, class BOp>
__device__ __forceinline__ void mfma_mainloop(const AOp& aop, const BOp& bop, int m0, int n0, int kbeg,
                                              int kend, float* lds, f32x16 (&out)[TS::TM][TS::TN],
                                              bool& owner) {
  f32x16 acc[AccSplit<TS>::N][TS::TM][TS::TN];
  owner = true;
  constexpr int BM = TS::BM, BK = TS::BK;
  constexpr int LDA = ATile::LD, LDB = BTile::LD;
  float* As = lds;
  float* Bs = lds + 2 * BK * (BM + 4);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wk = wave / (TS::WM * TS::WN);
  const int wmn = wave % (TS::WM * TS::WN);
  const int wm = wmn / TS::WN, wn = wmn % TS::WN;
  const int i = lane & 31, h = lane >> 5;

#pragma unroll
  for (int a = 0; a < TS::TM; ++a)
#pragma unroll
    for (int b = 0; b < TS::TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        out[a][b][r] = 0.f;
#pragma unroll
        for (int z = 0; z < AccSplit<TS>::N; ++z) acc[z][a][b][r] = 0.f;
      }

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk <= 0) return;
  int n_full = ATile::full_tiles(aop, kbeg, kend);
  const int nfb = BTile::full_tiles(bop, kbeg, kend);
  if (nfb < n_full) n_full = nfb;
  if (n_full > nk) n_full = nk;

  ATile at;
  BTile bt;
  at.init(aop, m0, tid);
  bt.init(bop, n0, tid);
  if (n_full > 0) {
    at.load_full(aop, m0, kbeg, tid);
    bt.load_full(bop, n0, kbeg, tid);
  } else {
    at.load_tail(aop, m0, kbeg, tid);
    bt.load_tail(bop, n0, kbeg, tid);
  }
  at.store(As, tid);
  bt.store(Bs, tid);
  __syncthreads();
  int cur = 0;
  // + the first k-step of this wave-group inside a staged tile
  const int aoff = wm * TS::TM * 32 + i + wk * (BK / TS::WK) * LDA;
  const int boff = wn * TS::TN * 32 + i + wk * (BK / TS::WK) * LDB;
  for (int t = 0; t + 1 < n_full; ++t) {
    at.load_full(aop, m0, kbeg + (t + 1) * BK, tid);
    bt.load_full(bop, n0, kbeg + (t + 1) * BK, tid);
    mfma_tile_compute<TS, ATile, BTile>(As + cur * BK * LDA + aoff, Bs + cur * BK * LDB + boff, h, acc);
    at.store(As + (cur ^ 1) * BK * LDA, tid);
    bt.store(Bs + (cur ^ 1) * BK * LDB, tid);
    __syncthreads();
    cur ^= 1;
  }
  // guarded tiles behind the steady state (one for dense operands; all of them for loaders that report no
  // unguarded tiles, e.g. the 3-channel first conv layer)
  for (int u = (n_full > 0 ? n_full : 1); u < nk; ++u) {
    at.load_tail(aop, m0, kbeg + u * BK, tid);
    bt.load_tail(bop, n0, kbeg + u * BK, tid);
    mfma_tile_compute<TS, ATile, BTile>(As + cur * BK * LDA + aoff, Bs + cur * BK * LDB + boff, h, acc);
    at.store(As + (cur ^ 1) * BK * LDA, tid);
    bt.store(Bs + (cur ^ 1) * BK * LDB, tid);
    __syncthreads();
    cur ^= 1;
  }
  mfma_tile_compute<TS, ATile, BTile>(As + cur * BK * LDA + aoff, Bs + cur * BK * LDB + boff, h, acc);
#pragma unroll
  for (int a = 0; a < TS::TM; ++a)
#pragma unroll
    for (int b = 0; b < TS::TN; ++b) {
      out[a][b] = acc[0][a][b];
#pragma unroll
      for (int z = 1; z < AccSplit<TS>::N; ++z) out[a][b] += acc[z][a][b];
    }
  if constexpr (TS::WK > 1) {
    // reduce the wave-groups' partial tiles through LDS (the staging buffers are free now): wave-group wk > 0
    // parks its tile in slot (wk-1, wmn); wave-group 0 adds them in wk order
    constexpr int kTile = TS::TM * TS::TN * 16 * 64;
    static_assert((TS::WK - 1) * TS::WM * TS::WN * kTile <= TS::lds_floats, "reduction must fit the staging LDS");
    __syncthreads();
    if (wk > 0) {
      float* red = lds + (long)((wk - 1) * TS::WM * TS::WN + wmn) * kTile;
#pragma unroll
      for (int a = 0; a < TS::TM; ++a)
#pragma unroll
        for (int b = 0; b < TS::TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((a * TS::TN + b) * 16 + r) * 64 + lane] = out[a][b][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int g = 1; g < TS::WK; ++g) {
        const float* red = lds + (long)((g - 1) * TS::WM * TS::WN + wmn) * kTile;
#pragma unroll
        for (int a = 0; a < TS::TM; ++a)
#pragma unroll
          for (int b = 0; b < TS::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[a][b][r] += red[((a * TS::TN + b) * 16 + r) * 64 + lane];
      }
    }
    owner = (wk == 0);
  }
}

}  // namespace dv3
