// 64x64 CNN encoder / decoder as implicit GEMMs on the fp32 MFMA tile engine (mfma_gemm.h).
//
// Activations are NHWC (channel-last) end to end: the replay image arrives [B,T,64,64,3], the
// channel LayerNorm reduces over the contiguous dim, the decoder's Linear output is already
// (h,w,c) (networks.py:571-573) and its mean is returned NHWC (networks.py:580).  Only the
// encoder's final flatten is (C,H,W) (networks.py:494); dv3_ln_act_fwd(chw_group=16) does that.
//
// Three compute kernels cover forward and backward of both stacks:
//   conv_s2    : Conv2d k4 s2 "same" pad (networks.py:771-798)        = encoder fwd = decoder dgrad
//   convT_s2   : ConvTranspose2d k4 s2 p1 (networks.py:540-550)       = decoder fwd = encoder dgrad
//   conv_wgrad : weight gradient of either (split-K, fp32 atomics straight into the reference
//                [out|in][in|out][4][4] layout)
// A Conv2d weight [Co][Ci][4][4] read as a ConvTranspose2d weight [in=Co][out=Ci][4][4] is its
// adjoint, so the same packed images serve forward and backward.
#include <stdlib.h>

#include "mfma_gemm.h"
#include "dv3_common.h"

namespace dv3 {

// x / d and x % d without an integer division when d is a power of two (every spatial size and most
// channel counts here are): runtime integer division is ~40 VALU instructions on CDNA and the im2col
// address decode would otherwise out-cost the MFMAs it feeds.
struct FastDiv {
  int d, shift;  // shift >= 0: d == 1 << shift
  __host__ __device__ static FastDiv make(int d) {
    FastDiv f{d, -1};
    if (d > 0 && (d & (d - 1)) == 0) {
      int s = 0;
      while ((1 << s) < d) ++s;
      f.shift = s;
    }
    return f;
  }
  __device__ __forceinline__ long quot(long x) const { return shift >= 0 ? (x >> shift) : x / d; }
  __device__ __forceinline__ int rem(long x) const { return shift >= 0 ? (int)(x & (d - 1)) : (int)(x % d); }
  __device__ __forceinline__ int quot(int x) const { return shift >= 0 ? (x >> shift) : x / d; }
  __device__ __forceinline__ int rem(int x) const { return shift >= 0 ? (x & (d - 1)) : x % d; }
};

// ------------------------------------------------------------------------------------------------
// A loader for conv_s2: row m = (n, oy, ox) of the output grid, k = (ky, kx, ci).
// ------------------------------------------------------------------------------------------------
struct ConvA {
  const float* x;  // [Nimg][H][W][C]
  int H, W, C, OH, OW;
  long M;  // Nimg*OH*OW
  int K;   // 16*C
  FastDiv dC, dC4;
};

template <int ROWS, int BK>
struct ConvATile {
  static constexpr int kVecs = ROWS * BK / 4 / kThreads;
  static constexpr int CH = BK / 4, RPP = kThreads / CH;
  static constexpr int LD = ROWS + (BK == 16 ? 2 : 1);
  f32x4 v[kVecs];
  float msk[kVecs];  // 0 for padding taps / rows past the edge of the staged tile, applied in store()
  long base[kVecs];  // n*H*W*C, or -1 when the row is out of range
  int iy0[kVecs], ix0[kVecs];

  __device__ __forceinline__ void init(const ConvA& op, int r0, int tid) {
    const int rr = tid / CH;
#pragma unroll
    for (int p = 0; p < kVecs; ++p) {
      const long r = (long)r0 + p * RPP + rr;
      if (r < op.M) {
        const int ox = (int)(r % op.OW);
        const long t = r / op.OW;
        const int oy = (int)(t % op.OH);
        const long n = t / op.OH;
        base[p] = n * op.H * op.W * op.C;
        iy0[p] = 2 * oy - 1;
        ix0[p] = 2 * ox - 1;
      } else {
        base[p] = -1;
        iy0[p] = ix0[p] = 0;
      }
    }
  }
  __device__ __forceinline__ float at(const ConvA& op, int p, int k) const {
    if (base[p] < 0 || k >= op.K) return 0.f;
    const int c4 = 4 * op.C;
    const int ky = k / c4, j = k - ky * c4;
    const int kx = j / op.C, ci = j - kx * op.C;
    const int iy = iy0[p] + ky, ix = ix0[p] + kx;
    if (iy < 0 || iy >= op.H || ix < 0 || ix >= op.W) return 0.f;
    return op.x[base[p] + ((long)iy * op.W + ix) * op.C + ci];
  }
  // every tile goes through the same predicated gather (padding taps make guards inherent); the guard
  // selects the ADDRESS, not the control flow, so the loads stay in flight across the MFMAs
  __device__ __forceinline__ static int full_tiles(const ConvA& op, int kbeg, int kend) {
    return ((op.C & 3) == 0) ? (kend - kbeg + BK - 1) / BK : 0;
  }
  __device__ __forceinline__ void load_full(const ConvA& op, int r0, int k0, int tid) { load(op, r0, k0, tid); }
  __device__ __forceinline__ void load_tail(const ConvA& op, int r0, int k0, int tid) { load(op, r0, k0, tid); }
  __device__ __forceinline__ void load(const ConvA& op, int, int k0, int tid) {
    const int c = tid % CH;
    const int k = k0 + 4 * c;
    if ((op.C & 3) == 0) {
      // 4 consecutive k share (ky,kx): one 16-byte gather or zeros
      const int ky = op.dC4.quot(k), j = op.dC4.rem(k);
      const int kx = op.dC.quot(j), ci = op.dC.rem(j);
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const int iy = iy0[p] + ky, ix = ix0[p] + kx;
        const bool ok = base[p] >= 0 && k < op.K && iy >= 0 && iy < op.H && ix >= 0 && ix < op.W;
        const float* src = ok ? op.x + base[p] + ((long)iy * op.W + ix) * op.C + ci : op.x;
        const f32x4 t = *reinterpret_cast<const f32x4u*>(src);
        v[p] = t;                 // consumed only in store(): the gather stays in flight across the MFMAs
        msk[p] = ok ? 1.f : 0.f;  // (a multiply here would pin a vmcnt(0) wait in front of them)
      }
    } else {
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = at(op, p, k + e);
        v[p] = t;
        msk[p] = 1.f;
      }
    }
  }
  __device__ __forceinline__ void store(float* s, int tid) const {
    const int c = tid % CH, rr = tid / CH;
#pragma unroll
    for (int p = 0; p < kVecs; ++p) {
      const int r = p * RPP + rr;
#pragma unroll
      for (int e = 0; e < 4; ++e) s[(4 * c + e) * LD + r] = v[p][e] * msk[p];
    }
  }
};

// ------------------------------------------------------------------------------------------------
// A loader for convT_s2, one output-parity class (py,px) per launch slice: row m = (n, y2, x2) of
// the INPUT grid, k = (a, b, ci), tap input pixel (y2 + py - a, x2 + px - b), kernel tap
// (ky, kx) = (1 - py + 2a, 1 - px + 2b).
// ------------------------------------------------------------------------------------------------
struct ConvTA {
  const float* x;  // [Nimg][IH][IW][C]
  int IH, IW, C, py, px;
  long M;  // Nimg*IH*IW
  int K;   // 4*C
  FastDiv dC, dC2;
};

template <int ROWS, int BK>
struct ConvTATile {
  static constexpr int kVecs = ROWS * BK / 4 / kThreads;
  static constexpr int CH = BK / 4, RPP = kThreads / CH;
  static constexpr int LD = ROWS + (BK == 16 ? 2 : 1);
  f32x4 v[kVecs];
  float msk[kVecs];
  long base[kVecs];
  int y0[kVecs], x0[kVecs];

  __device__ __forceinline__ void init(const ConvTA& op, int r0, int tid) {
    const int rr = tid / CH;
#pragma unroll
    for (int p = 0; p < kVecs; ++p) {
      const long r = (long)r0 + p * RPP + rr;
      if (r < op.M) {
        const int x2 = (int)(r % op.IW);
        const long t = r / op.IW;
        const int y2 = (int)(t % op.IH);
        const long n = t / op.IH;
        base[p] = n * op.IH * op.IW * op.C;
        y0[p] = y2 + op.py;
        x0[p] = x2 + op.px;
      } else {
        base[p] = -1;
        y0[p] = x0[p] = 0;
      }
    }
  }
  __device__ __forceinline__ float at(const ConvTA& op, int p, int k) const {
    if (base[p] < 0 || k >= op.K) return 0.f;
    const int c2 = 2 * op.C;
    const int a = k / c2, j = k - a * c2;
    const int b = j / op.C, ci = j - b * op.C;
    const int iy = y0[p] - a, ix = x0[p] - b;
    if (iy < 0 || iy >= op.IH || ix < 0 || ix >= op.IW) return 0.f;
    return op.x[base[p] + ((long)iy * op.IW + ix) * op.C + ci];
  }
  __device__ __forceinline__ static int full_tiles(const ConvTA& op, int kbeg, int kend) {
    return ((op.C & 3) == 0) ? (kend - kbeg + BK - 1) / BK : 0;
  }
  __device__ __forceinline__ void load_full(const ConvTA& op, int r0, int k0, int tid) { load(op, r0, k0, tid); }
  __device__ __forceinline__ void load_tail(const ConvTA& op, int r0, int k0, int tid) { load(op, r0, k0, tid); }
  __device__ __forceinline__ void load(const ConvTA& op, int, int k0, int tid) {
    const int c = tid % CH;
    const int k = k0 + 4 * c;
    if ((op.C & 3) == 0) {
      const int a = op.dC2.quot(k), j = op.dC2.rem(k);
      const int b = op.dC.quot(j), ci = op.dC.rem(j);
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const int iy = y0[p] - a, ix = x0[p] - b;
        const bool ok = base[p] >= 0 && k < op.K && iy >= 0 && iy < op.IH && ix >= 0 && ix < op.IW;
        const float* src = ok ? op.x + base[p] + ((long)iy * op.IW + ix) * op.C + ci : op.x;
        const f32x4 t = *reinterpret_cast<const f32x4u*>(src);
        v[p] = t;                 // consumed only in store(): the gather stays in flight across the MFMAs
        msk[p] = ok ? 1.f : 0.f;  // (a multiply here would pin a vmcnt(0) wait in front of them)
      }
    } else {
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = at(op, p, k + e);
        v[p] = t;
        msk[p] = 1.f;
      }
    }
  }
  __device__ __forceinline__ void store(float* s, int tid) const {
    const int c = tid % CH, rr = tid / CH;
#pragma unroll
    for (int p = 0; p < kVecs; ++p) {
      const int r = p * RPP + rr;
#pragma unroll
      for (int e = 0; e < 4; ++e) s[(4 * c + e) * LD + r] = v[p][e] * msk[p];
    }
  }
};

// ------------------------------------------------------------------------------------------------
// B loader for conv_wgrad: reduction index k = output-grid row m = (n, oy, ox); tile column
// j = (ky, kx, ci) gathers x[n, 2oy+ky-1, 2ox+kx-1, ci].  LDS image Bs[k][j], written 16 B at a time.
// ------------------------------------------------------------------------------------------------
struct WgradB {
  const float* x;  // [Nimg][H][W][C]  (the stride-2-sampled, "fine" tensor)
  int H, W, C, OH, OW;
  long Mrows;  // Nimg*OH*OW  (the reduction length)
  int Ncols;   // 16*C
  FastDiv dOW, dOH, dC, dC4;
};

template <int ROWS, int BK, bool C3 = false>  // C3: the fine tensor has 3 channels (image side)
struct WgradBTile {
  static constexpr int kVecs = ROWS * BK / 4 / kThreads;
  static constexpr int CH = ROWS / 4, KPP = kThreads / CH;
  static constexpr int LD = ROWS + 4;
  static_assert(kThreads % CH == 0 && BK % KPP == 0, "bad tile");
  f32x4 v[kVecs];
  float msk[kVecs];
  int edge[kVecs];  // C == 3 only: 1 = left image edge (keep loaded[0] as element 3), 2 = right edge (loaded[3] as element 0)
  int jcol;

  __device__ __forceinline__ void init(const WgradB&, int n0, int tid) { jcol = n0 + 4 * (tid % CH); }
  __device__ __forceinline__ float at(const WgradB& op, long m, int j) const {
    if (m >= op.Mrows || j >= op.Ncols) return 0.f;
    const int c4 = 4 * op.C;
    const int ky = j / c4, jj = j - ky * c4;
    const int kx = jj / op.C, ci = jj - kx * op.C;
    const int ox = (int)(m % op.OW);
    const long t = m / op.OW;
    const int oy = (int)(t % op.OH);
    const long n = t / op.OH;
    const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
    if (iy < 0 || iy >= op.H || ix < 0 || ix >= op.W) return 0.f;
    return op.x[((n * op.H + iy) * op.W + ix) * op.C + ci];
  }
  __device__ __forceinline__ static int full_tiles(const WgradB& op, int kbeg, int kend) {
    return (C3 || (op.C & 3) == 0) ? (kend - kbeg + BK - 1) / BK : 0;
  }
  __device__ __forceinline__ void load_full(const WgradB& op, int r0, int k0, int tid) { load(op, r0, k0, tid); }
  __device__ __forceinline__ void load_tail(const WgradB& op, int r0, int k0, int tid) { load(op, r0, k0, tid); }
  __device__ __forceinline__ void load(const WgradB& op, int, int k0, int tid) {
    const int kr = tid / CH;
    if constexpr (C3) {
      // image-side layer: the 12 floats (kx, ci) of one kernel row are CONTIGUOUS in NHWC, so this lane's four
      // columns j = jcol..jcol+3 are one 16-byte load at x[n][iy][2ox-1][0] + jcol % 12.  Only the first and
      // last output column of an image row reach outside it: there the load is shifted by one pixel to stay
      // inside the row and store() keeps the single float that belongs to this lane (edge = 1 / 2).
      const int ky = jcol / 12, off = jcol - 12 * ky;
      const bool colok = jcol < 48;
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const long m = (long)k0 + p * KPP + kr;
        const long mc = (m < op.Mrows) ? m : 0;
        const int ox = op.dOW.rem(mc);
        const long q = op.dOW.quot(mc);
        const int oy = op.dOH.rem(q);
        const long n = op.dOH.quot(q);
        const int iy = 2 * oy - 1 + ky;
        const bool ok = colok && m < op.Mrows && iy >= 0 && iy < op.H;
        const int e1 = (ox == 0 && off == 0) ? 1 : (ox == op.OW - 1 && off == 8) ? 2 : 0;
        const long f = ((n * op.H + iy) * op.W + (2 * ox - 1)) * 3 + off + (e1 == 1 ? 3 : e1 == 2 ? -3 : 0);
        v[p] = *reinterpret_cast<const f32x4u*>(ok ? op.x + f : op.x);
        msk[p] = ok ? 1.f : 0.f;
        edge[p] = e1;
      }
      return;
    }
    if ((op.C & 3) == 0) {
      const int ky = op.dC4.quot(jcol), jj = op.dC4.rem(jcol);
      const int kx = op.dC.quot(jj), ci = op.dC.rem(jj);
      const bool colok = jcol < op.Ncols;
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const long m = (long)k0 + p * KPP + kr;
        const long mc = (m < op.Mrows) ? m : 0;
        const int ox = op.dOW.rem(mc);
        const long q = op.dOW.quot(mc);
        const int oy = op.dOH.rem(q);
        const long n = op.dOH.quot(q);
        const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
        const bool ok = colok && m < op.Mrows && iy >= 0 && iy < op.H && ix >= 0 && ix < op.W;
        const float* src = ok ? op.x + ((n * op.H + iy) * op.W + ix) * op.C + ci : op.x;
        const f32x4 t = *reinterpret_cast<const f32x4u*>(src);
        v[p] = t;                 // consumed only in store(): the gather stays in flight across the MFMAs
        msk[p] = ok ? 1.f : 0.f;  // (a multiply here would pin a vmcnt(0) wait in front of them)
      }
    } else {
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        const long m = (long)k0 + p * KPP + kr;
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = at(op, m, jcol + e);
        v[p] = t;
        msk[p] = 1.f;
      }
    }
  }
  __device__ __forceinline__ void store(float* s, int tid) const {
    const int c = tid % CH, kr = tid / CH;
#pragma unroll
    for (int p = 0; p < kVecs; ++p) {
      f32x4 t = v[p];
      if constexpr (C3) {
        if (edge[p] == 1) t = (f32x4){0.f, 0.f, 0.f, t[0]};
        else if (edge[p] == 2) t = (f32x4){t[3], 0.f, 0.f, 0.f};
      }
      *reinterpret_cast<f32x4*>(&s[(p * KPP + kr) * LD + 4 * c]) = t * msk[p];
    }
  }
};

// ------------------------------------------------------------------------------------------------
struct ConvParams {
  const float* x;
  const float* wp;  // packed weights
  float* y;
  const float* bias;
  float out_add;
  int Nimg, H, W, Ci, Co;  // H, W: spatial size of the FINE tensor for conv (input), COARSE for convT (input)
  int tiles_m, tiles_n;
  int accumulate;
};

// y[n,oy,ox,co] = sum x[n,2oy+ky-1,2ox+kx-1,ci] * wp[co][(ky,kx,ci)]
template <class TS>
__global__ __launch_bounds__(kThreads) void conv_s2_kernel(ConvParams p) {
  __shared__ __attribute__((aligned(16))) float lds[TS::lds_floats];
  using ATile = ConvATile<TS::BM, TS::BK>;
  using BTile = DenseTile<TS::BN, TS::BK, true>;
  const int OH = p.H / 2, OW = p.W / 2;
  ConvA aop{p.x, p.H, p.W, p.Ci, OH, OW, (long)p.Nimg * OH * OW, 16 * p.Ci, FastDiv::make(p.Ci),
            FastDiv::make(4 * p.Ci)};
  DenseOperand<true> bop{p.wp, nullptr, 16L * p.Ci, 0, p.Co, 16 * p.Ci, 16 * p.Ci, true};
  const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int m0 = (wg / p.tiles_n) * TS::BM, n0 = (wg % p.tiles_n) * TS::BN;
  f32x16 acc[TS::TM][TS::TN];
  bool owner;
  mfma_mainloop<TS, ATile, BTile>(aop, bop, m0, n0, 0, aop.K, lds, acc, owner);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / TS::WN, wn = wave % TS::WN, col_l = lane & 31, h = lane >> 5;
#pragma unroll
  for (int a = 0; a < TS::TM; ++a)
#pragma unroll
    for (int b = 0; b < TS::TN; ++b) {
      const int n = n0 + (wn * TS::TN + b) * 32 + col_l;
      if (n >= p.Co) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = (long)m0 + (wm * TS::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < aop.M) {
          float* o = p.y + m * p.Co + n;
          *o = p.accumulate ? (*o + acc[a][b][r]) : acc[a][b][r];
        }
      }
    }
}

// y[n,2y2+py,2x2+px,co] = sum_{a,b,ci} x[n,y2+py-a,x2+px-b,ci] * wp[cls][co][(a,b,ci)] + bias[co] + out_add
template <class TS>
__global__ __launch_bounds__(kThreads) void convT_s2_kernel(ConvParams p) {
  __shared__ __attribute__((aligned(16))) float lds[TS::lds_floats];
  using ATile = ConvTATile<TS::BM, TS::BK>;
  using BTile = DenseTile<TS::BN, TS::BK, true>;
  const int cls = blockIdx.y, py = cls >> 1, px = cls & 1;
  ConvTA aop{p.x, p.H, p.W, p.Ci, py, px, (long)p.Nimg * p.H * p.W, 4 * p.Ci, FastDiv::make(p.Ci),
             FastDiv::make(2 * p.Ci)};
  DenseOperand<true> bop{p.wp + (long)cls * p.Co * 4 * p.Ci, nullptr, 4L * p.Ci, 0, p.Co, 4 * p.Ci, 4 * p.Ci, true};
  const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int m0 = (wg / p.tiles_n) * TS::BM, n0 = (wg % p.tiles_n) * TS::BN;
  f32x16 acc[TS::TM][TS::TN];
  bool owner;
  mfma_mainloop<TS, ATile, BTile>(aop, bop, m0, n0, 0, aop.K, lds, acc, owner);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / TS::WN, wn = wave % TS::WN, col_l = lane & 31, h = lane >> 5;
  const int OH = 2 * p.H, OW = 2 * p.W;
#pragma unroll
  for (int a = 0; a < TS::TM; ++a)
#pragma unroll
    for (int b = 0; b < TS::TN; ++b) {
      const int n = n0 + (wn * TS::TN + b) * 32 + col_l;
      if (n >= p.Co) continue;
      const float add = (p.bias ? p.bias[n] : 0.f) + p.out_add;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = (long)m0 + (wm * TS::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < aop.M) {
          const int x2 = (int)(m % p.W);
          const long t = m / p.W;
          const int y2 = (int)(t % p.H);
          const long img = t / p.H;
          float* o = p.y + ((img * OH + 2 * y2 + py) * OW + 2 * x2 + px) * p.Co + n;
          const float val = acc[a][b][r] + add;
          *o = p.accumulate ? (*o + val) : val;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Narrow transposed convolution (Co <= 32: the decoder layer in front of the image layer, and the encoder's
// second layer read backwards), register-direct: on the 128x32 LDS tile every gathered A element is staged in LDS
// to be read exactly once (46 TF/s).  Here a WAVE owns 32 rows (input pixels of one parity class) x 32 output
// channels over the whole K = 4*Ci: per 16-k chunk two 16-byte gathers (one per 16-row tile) and two 16-byte
// weight loads (32 KB of weights per class, cache resident) feed 16 v_mfma_f32_16x16x4_f32; no LDS, no barriers,
// ~80 registers, so many waves per SIMD hide the gather latency.  Ci % 4 == 0.
// grid = (ceil(M / 128), 4 parity classes), 4 waves per workgroup (4 consecutive 32-row groups).
// ------------------------------------------------------------------------------------------------
// The four parity classes of a transposed convolution read the SAME input pixels.  With the class as grid.y the chip
// sweeps the whole input once per class (4 x the input through the fabric: convT_s2_direct<2> 656 MB per launch against
// 200 MB algorithmic).  1-D grid instead: workgroup ids go round-robin to the 8 XCDs, so ids id, id+8, id+16, id+24
// land on the same XCD back to back -- they are made the four classes of ONE pixel tile, which then comes out of that
// XCD's L2 three times out of four.  grid.x = ceil8(tiles) * 4; returns false for the padding workgroups.
__device__ __forceinline__ bool convT_tile_class(int tiles, int& tile, int& cls) {
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  cls = j & 3;
  tile = (j >> 2) * 8 + xcd;
  return tile < tiles;
}
static unsigned convT_grid(long tiles) { return (unsigned)((tiles + 7) / 8 * 8 * 4); }

typedef float f32x4n __attribute__((ext_vector_type(4)));
template <int RN>  // 16*RN output channels per wave
__global__ __launch_bounds__(256) void convT_s2_direct_kernel(ConvParams p) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const long M = (long)p.Nimg * p.H * p.W;
  int tile, cls;
  if (!convT_tile_class((int)((M + 127) / 128), tile, cls)) return;
  const int py = cls >> 1, px = cls & 1;
  const int K = 4 * p.Ci;
  const long row0 = (long)tile * 128 + wave * 32;
  if (row0 >= M) return;
  const FastDiv dW = FastDiv::make(p.W), dH = FastDiv::make(p.H), dC = FastDiv::make(p.Ci);
  // the two rows (input pixels) this lane gathers for: row0 + 16 t + i
  long base[2];
  int y0[2], x0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long r = row0 + 16 * t + i;
    const long rc = r < M ? r : 0;
    const int x2 = dW.rem(rc);
    const long tq = dW.quot(rc);
    const int y2 = dH.rem(tq);
    const long n = dH.quot(tq);
    base[t] = r < M ? n * p.H * p.W * p.Ci : -1;
    y0[t] = y2 + py;
    x0[t] = x2 + px;
  }
  const float* wrow[RN];
  float bmask[RN];
#pragma unroll
  for (int c = 0; c < RN; ++c) {
    const int co = 16 * c + i;
    bmask[c] = co < p.Co ? 1.f : 0.f;
    wrow[c] = p.wp + ((long)cls * p.Co + (co < p.Co ? co : 0)) * K;
  }
  f32x4n acc[2][RN];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < RN; ++c) acc[t][c] = (f32x4n){0.f, 0.f, 0.f, 0.f};
  const int chunks = (K + 15) >> 4;
  f32x4n a0[2], b0[RN], a1[2], b1[RN];
  float m0[2], m1[2];
  auto load = [&](f32x4n (&a)[2], f32x4n (&b)[RN], float (&msk)[2], int c) {
    if (c >= chunks) return;
    const int k = (c << 4) + 4 * q;
    const bool kok = k < K;  // K % 4 == 0: a lane's four k values are all in or all out
    const int kk = kok ? k : 0;
    const int tap = dC.quot(kk), ci = dC.rem(kk);
    const int ta = tap >> 1, tb = tap & 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int iy = y0[t] - ta, ix = x0[t] - tb;
      const bool ok = kok && base[t] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const float* src = ok ? p.x + base[t] + ((long)iy * p.W + ix) * p.Ci + ci : p.x;
      a[t] = *reinterpret_cast<const f32x4u*>(src);
      msk[t] = ok ? 1.f : 0.f;
    }
#pragma unroll
    for (int cc = 0; cc < RN; ++cc) b[cc] = *reinterpret_cast<const f32x4u*>(wrow[cc] + kk);
  };
  auto compute = [&](const f32x4n (&a)[2], const f32x4n (&b)[RN], const float (&msk)[2], int c) {
    if (c >= chunks) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float av = a[t][g] * msk[t];
#pragma unroll
        for (int cc = 0; cc < RN; ++cc)
          acc[t][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[cc][g] * bmask[cc], acc[t][cc], 0, 0, 0);
      }
  };
  load(a0, b0, m0, 0);
  for (int c = 0; c < chunks; c += 2) {
    load(a1, b1, m1, c + 1);
    compute(a0, b0, m0, c);
    load(a0, b0, m0, c + 2);
    compute(a1, b1, m1, c + 1);
  }
  // acc[t][cc][r]: row 16 t + 4 (lane >> 4) + r, channel 16 cc + (lane & 15)
  const int OH = 2 * p.H, OW = 2 * p.W;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = row0 + 16 * t + 4 * q + r;
      if (m >= M) continue;
      const int x2 = dW.rem(m);
      const long tq = dW.quot(m);
      const int y2 = dH.rem(tq);
      const long img = dH.quot(tq);
      float* orow = p.y + ((img * OH + 2 * y2 + py) * OW + 2 * x2 + px) * p.Co;
#pragma unroll
      for (int cc = 0; cc < RN; ++cc) {
        const int co = 16 * cc + i;
        if (co < p.Co) {
          const float val = acc[t][cc][r] + (p.bias ? p.bias[co] : 0.f) + p.out_add;
          orow[co] = p.accumulate ? (orow[co] + val) : val;
        }
      }
    }
}

// The stride-2 forward convolution in the same register-direct form (Co <= 64, where it measured faster than the
// LDS tiles: encoder layer 2 and the decoder's data gradients; DV3_CONV_DIRECT / DV3_CONVT_DIRECT = 0 switch both
// off for A/B runs): row = output pixel (n, oy, ox), k = (ky, kx, ci) over K = 16*Ci, input pixel
// (2oy + ky - 1, 2ox + kx - 1).  grid = ceil(M / 128) workgroups of 4 waves, 32 rows each.
template <int RN>
__global__ __launch_bounds__(256) void conv_s2_direct_kernel(ConvParams p) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int OH = p.H / 2, OW = p.W / 2;
  const long M = (long)p.Nimg * OH * OW;
  const int K = 16 * p.Ci;
  const long row0 = (long)blockIdx.x * 128 + wave * 32;
  if (row0 >= M) return;
  const FastDiv dW = FastDiv::make(OW), dH = FastDiv::make(OH), dC = FastDiv::make(p.Ci);
  long base[2];
  int y0[2], x0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long r = row0 + 16 * t + i;
    const long rc = r < M ? r : 0;
    const int ox = dW.rem(rc);
    const long tq = dW.quot(rc);
    const int oy = dH.rem(tq);
    const long n = dH.quot(tq);
    base[t] = r < M ? n * p.H * p.W * p.Ci : -1;
    y0[t] = 2 * oy - 1;
    x0[t] = 2 * ox - 1;
  }
  const float* wrow[RN];
  float bmask[RN];
#pragma unroll
  for (int c = 0; c < RN; ++c) {
    const int co = 16 * c + i;
    bmask[c] = co < p.Co ? 1.f : 0.f;
    wrow[c] = p.wp + (long)(co < p.Co ? co : 0) * K;
  }
  f32x4n acc[2][RN];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < RN; ++c) acc[t][c] = (f32x4n){0.f, 0.f, 0.f, 0.f};
  const int chunks = (K + 15) >> 4;
  f32x4n a0[2], b0[RN], a1[2], b1[RN];
  float m0[2], m1[2];
  auto load = [&](f32x4n (&a)[2], f32x4n (&b)[RN], float (&msk)[2], int c) {
    if (c >= chunks) return;
    const int k = (c << 4) + 4 * q;  // K % 16 == 0
    const int tap = dC.quot(k), ci = dC.rem(k);
    const int ky = tap >> 2, kx = tap & 3;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int iy = y0[t] + ky, ix = x0[t] + kx;
      const bool ok = base[t] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const float* src = ok ? p.x + base[t] + ((long)iy * p.W + ix) * p.Ci + ci : p.x;
      a[t] = *reinterpret_cast<const f32x4u*>(src);
      msk[t] = ok ? 1.f : 0.f;
    }
#pragma unroll
    for (int cc = 0; cc < RN; ++cc) b[cc] = *reinterpret_cast<const f32x4u*>(wrow[cc] + k);
  };
  auto compute = [&](const f32x4n (&a)[2], const f32x4n (&b)[RN], const float (&msk)[2], int c) {
    if (c >= chunks) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float av = a[t][g] * msk[t];
#pragma unroll
        for (int cc = 0; cc < RN; ++cc)
          acc[t][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[cc][g] * bmask[cc], acc[t][cc], 0, 0, 0);
      }
  };
  load(a0, b0, m0, 0);
  for (int c = 0; c < chunks; c += 2) {
    load(a1, b1, m1, c + 1);
    compute(a0, b0, m0, c);
    load(a0, b0, m0, c + 2);
    compute(a1, b1, m1, c + 1);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = row0 + 16 * t + 4 * q + r;
      if (m >= M) continue;
      float* orow = p.y + m * p.Co;
#pragma unroll
      for (int cc = 0; cc < RN; ++cc) {
        const int co = 16 * cc + i;
        if (co < p.Co) orow[co] = p.accumulate ? (orow[co] + acc[t][cc][r]) : acc[t][cc][r];
      }
    }
}

struct WgradParams {
  const float* dy;  // [rows][Co]   coarse-grid tensor (conv: dY; convT: the layer input)
  const float* x;   // [Nimg][H][W][Ci]  fine-grid tensor (conv: the layer input; convT: dOut)
  float* dw;        // packed [Co][(ky,kx,ci)] scratch, accumulated with atomics
  int Nimg, H, W, Ci, Co;
  int tiles_m, tiles_n, splits, chunk;  // chunk: reduction rows per split (multiple of BK)
  int xcd_group;  // grid = tiles * ceil8(splits): all output tiles of one K-split run on ONE XCD (see the kernel)
};

// dw_packed[co][(ky,kx,ci)] += sum_m dy[m][co] * x[n,2oy+ky-1,2ox+kx-1,ci]
template <class TS, bool C3 = false>
__global__ __launch_bounds__(kThreads) void conv_wgrad_kernel(WgradParams p) {
  __shared__ __attribute__((aligned(16))) float lds[TS::lds_floats];
  using ATile = DenseTile<TS::BM, TS::BK, false>;  // A[k=m][row=co], co-contiguous
  using BTile = WgradBTile<TS::BN, TS::BK, C3>;
  const int OH = p.H / 2, OW = p.W / 2;
  const long rows = (long)p.Nimg * OH * OW;
  DenseOperand<false> aop{p.dy, nullptr, (long)p.Co, 0, p.Co, (int)rows, (int)rows, true};
  WgradB bop{p.x, p.H, p.W, p.Ci, OH, OW, rows, 16 * p.Ci, FastDiv::make(OW), FastDiv::make(OH),
             FastDiv::make(p.Ci), FastDiv::make(4 * p.Ci)};
  const int tiles = p.tiles_m * p.tiles_n;
  // Every output tile of a K-split reads the same dy rows and the same x pixels.  Workgroup ids go round-robin to the
  // 8 XCDs, so with tile-major ids each XCD's L2 pulls EVERY split's operands (8 copies through the fabric, and at
  // crafter widths gigabytes from HBM: 12 column tiles x 2 row tiles re-read each chunk).  xcd_group: the tiles of
  // split s all get ids = s mod 8 (mod 8), dispatched back to back -- they run together on one XCD and share the
  // chunk through its L2.
  int tile, split;
  if (p.xcd_group) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    split = (j / tiles) * 8 + xcd;
    tile = j % tiles;
    if (split >= p.splits) return;  // padding workgroups of the last round (whole workgroup, before any barrier)
  } else {
    tile = blockIdx.x % tiles;
    split = blockIdx.x / tiles;
  }
  const int m0 = (tile / p.tiles_n) * TS::BM, n0 = (tile % p.tiles_n) * TS::BN;
  const long kb = (long)split * p.chunk;
  long ke = kb + p.chunk;
  if (ke > rows) ke = rows;
  f32x16 acc[TS::TM][TS::TN];
  bool owner;
  mfma_mainloop<TS, ATile, BTile>(aop, bop, m0, n0, (int)kb, (int)ke, lds, acc, owner);
  if (kb >= ke || !owner) return;
  const int tid = threadIdx.x, wave = (tid >> 6) % (TS::WM * TS::WN), lane = tid & 63;
  const int wm = wave / TS::WN, wn = wave % TS::WN, col_l = lane & 31, h = lane >> 5;
#pragma unroll
  for (int a = 0; a < TS::TM; ++a)
#pragma unroll
    for (int b = 0; b < TS::TN; ++b) {
      const int j = n0 + (wn * TS::TN + b) * 32 + col_l;
      if (j >= 16 * p.Ci) continue;
      // packed layout [co][(ky,kx,ci)]: a half-wave adds 32 consecutive floats (128 B) per instruction --
      // the shape fp32 atomics run at full rate on; dv3_unpack_conv_wgrad moves it to [co][ci][ky][kx]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + (wm * TS::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < p.Co) atomicAdd(p.dw + (long)co * (16 * p.Ci) + j, acc[a][b][r]);
      }
    }
}

// dw[co][ci][ky][kx] += packed[co][(ky,kx,ci)]; packed is cleared for the next update
__global__ void unpack_conv_wgrad_kernel(float* __restrict__ packed, float* __restrict__ dw, int Co, int Ci) {
  const long total = (long)Co * Ci * 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    long t = i / Ci;
    const int kx = (int)(t % 4);
    t /= 4;
    const int ky = (int)(t % 4);
    const long co = t / 4;
    dw[((co * Ci + ci) * 4 + ky) * 4 + kx] += packed[i];
    packed[i] = 0.f;
  }
}

// Conv2d weight [Co][Ci][4][4] -> [Co][(ky,kx,ci)]
__global__ void pack_conv_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci) {
  const long total = (long)Co * Ci * 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    long t = i / Ci;
    const int kx = (int)(t % 4);
    t /= 4;
    const int ky = (int)(t % 4);
    const long co = t / 4;
    wp[i] = w[((co * Ci + ci) * 4 + ky) * 4 + kx];
  }
}
// ConvTranspose2d weight [Ci][Co][4][4] -> [cls=(py,px)][Co][(a,b,ci)], (ky,kx) = (1-py+2a, 1-px+2b)
__global__ void pack_convT_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int Ci, int Co) {
  const long total = 4L * Co * 4 * Ci;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    long t = i / Ci;
    const int b = (int)(t % 2);
    t /= 2;
    const int a = (int)(t % 2);
    t /= 2;
    const int co = (int)(t % Co);
    const int cls = (int)(t / Co);
    const int py = cls >> 1, px = cls & 1;
    const int ky = 1 - py + 2 * a, kx = 1 - px + 2 * b;
    wp[i] = w[(((long)ci * Co + co) * 4 + ky) * 4 + kx];
  }
}


// ------------------------------------------------------------------------------------------------
// The 3-channel image layers (encoder's first conv, decoder's last transposed conv) have K = 48 or
// N = 3: on MFMA tiles they run at 3-14 TFLOP/s of padding.  They are HBM-bound (134 MB of fp32
// activations on the 32-channel side) with 1536 FMAs per pixel, so: one thread per coarse-grid pixel, the
// whole receptive field in registers, weights read with wave-uniform indices (scalar loads, they
// never touch a VGPR), 16-byte global accesses.  CW = channels on the wide side (multiple of 4).
// ------------------------------------------------------------------------------------------------
typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));

// y[n,oy,ox,0..CW) = sum_{ky,kx,c<3} x[n,2oy+ky-1,2ox+kx-1,c] * w[co][c][ky][kx]      (Conv2d 3 -> CW)
template <int CW>
__global__ __launch_bounds__(256) void conv_s2_c3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ y, int Nimg, int H, int W,
                                                         int accumulate) {
  const int OH = H / 2, OW = W / 2;
  const long total = (long)Nimg * OH * OW;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(p % OW);
    const long t = p / OW;
    const int oy = (int)(t % OH);
    const long n = t / OH;
    float in[4][12];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      const bool rowok = iy >= 0 && iy < H;
      const float* row = x + ((n * H + (rowok ? iy : 0)) * W) * 3;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        const bool ok = rowok && ix >= 0 && ix < W;
        const float* q = row + (ok ? ix : 0) * 3;
        const float m = ok ? 1.f : 0.f;
        in[ky][kx * 3 + 0] = q[0] * m;
        in[ky][kx * 3 + 1] = q[1] * m;
        in[ky][kx * 3 + 2] = q[2] * m;
      }
    }
    float* out = y + p * CW;
#pragma unroll 1
    for (int c0 = 0; c0 < CW; c0 += 4) {
      f4a acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wc = w + (long)(c0 + e) * 48;  // [co][c][ky][kx]
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) a += in[ky][kx * 3 + c] * wc[(c * 4 + ky) * 4 + kx];
        acc[e] = a;
      }
      f4a* o = reinterpret_cast<f4a*>(out + c0);
      if (accumulate) acc += *o;
      *o = acc;
    }
  }
}

// The same layer on the matrix cores: K = 48 = 4 kernel rows x 12 floats, and the 12 floats (kx, c) of one kernel row are
// CONTIGUOUS in the NHWC image, so a pixel's A row is four 48-byte segments.  A workgroup stages 128 output pixels
// x 48 (k-contiguous, stride 52) and the CW x 48 weights (reordered to (ky, kx, c) on the way in) in LDS and runs the
// ds_read_b128 / v_mfma_f32_16x16x4_f32 scheme of gemm_l16_kernel: 3 chunks of 16 k, 48 MFMAs per wave at CW = 32.
// One thread per (pixel, kernel-row pair); only the first / last output column of an image row needs the scalar path.
template <int CW>
__global__ __launch_bounds__(256) void conv_s2_c3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ y, int Nimg, int H, int W,
                                                              int accumulate) {
  constexpr int LD = 52, TN = CW / 16;
  __shared__ __attribute__((aligned(16))) float As[128 * LD];
  __shared__ __attribute__((aligned(16))) float Bs[CW * LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int OH = H / 2, OW = W / 2;
  const long total = (long)Nimg * OH * OW;
  const long p0 = (long)blockIdx.x * 128;
  for (int e = tid; e < CW * 48; e += 256) {  // w[co][c][ky][kx] -> Bs[co][ky*12 + kx*3 + c]
    const int co = e / 48, r = e - co * 48;
    const int c = r >> 4, ky = (r >> 2) & 3, kx = r & 3;
    Bs[co * LD + ky * 12 + kx * 3 + c] = w[e];
  }
  {
    const int pl = tid >> 1, kyb = (tid & 1) * 2;
    const long p = p0 + pl;
    const bool pv = p < total;
    const long pc = pv ? p : 0;
    const int ox = (int)(pc % OW);
    const long t = pc / OW;
    const int oy = (int)(t % OH);
    const long n = t / OH;
    const bool edge = (ox == 0) || (ox == OW - 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ky = kyb + kk;
      const int iy = 2 * oy - 1 + ky;
      const bool rowok = pv && iy >= 0 && iy < H;
      float* dst = &As[pl * LD + ky * 12];
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0;
      if (rowok) {
        const float* row = x + ((n * H + iy) * W) * 3;
        if (!edge) {
          const float* src = row + (2 * ox - 1) * 3;
          v0 = *reinterpret_cast<const f32x4u*>(src);
          v1 = *reinterpret_cast<const f32x4u*>(src + 4);
          v2 = *reinterpret_cast<const f32x4u*>(src + 8);
        } else {
          float tmp[12];
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            const int ix = 2 * ox - 1 + j / 3;
            tmp[j] = (ix >= 0 && ix < W) ? row[ix * 3 + j % 3] : 0.f;
          }
          v0 = (f32x4){tmp[0], tmp[1], tmp[2], tmp[3]};
          v1 = (f32x4){tmp[4], tmp[5], tmp[6], tmp[7]};
          v2 = (f32x4){tmp[8], tmp[9], tmp[10], tmp[11]};
        }
      }
      *reinterpret_cast<f32x4*>(dst) = v0;
      *reinterpret_cast<f32x4*>(dst + 4) = v1;
      *reinterpret_cast<f32x4*>(dst + 8) = v2;
    }
  }
  __syncthreads();
  f32x4 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x4 af[2], bf[TN];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const f32x4*>(&As[(wave * 32 + 16 * a + i) * LD + 16 * c + 4 * q]);
#pragma unroll
    for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const f32x4*>(&Bs[(16 * b + i) * LD + 16 * c + 4 * q]);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][g], bf[b][g], acc[a][b], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long p = p0 + wave * 32 + 16 * a + 4 * q + r;
        if (p < total) {
          float* o = y + p * CW + 16 * b + i;
          const float v = acc[a][b][r];
          *o = accumulate ? (*o + v) : v;
        }
      }
}

// y[n,2y2+py,2x2+px,co<3] = sum_{a,b,ci<CW} x[n,y2+py-a,x2+px-b,ci] * w[ci][co][1-py+2a][1-px+2b] + bias + add
// (ConvTranspose2d CW -> 3).  One thread per input-grid pixel computes its 2x2 output block.
template <int CW>
__global__ __launch_bounds__(256) void convT_s2_c3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float out_add,
                                                          float* __restrict__ y, int Nimg, int IH, int IW,
                                                          int accumulate) {
  const long total = (long)Nimg * IH * IW;
  const int OW = 2 * IW, OH = 2 * IH;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int x2 = (int)(p % IW);
    const long t = p / IW;
    const int y2 = (int)(t % IH);
    const long n = t / IH;
    float acc[2][2][3];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int co = 0; co < 3; ++co) acc[py][px][co] = 0.f;
#pragma unroll 1
    for (int c0 = 0; c0 < CW; c0 += 4) {
      f4a in[3][3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int iy = y2 + dy - 1, ix = x2 + dx - 1;
          const bool ok = iy >= 0 && iy < IH && ix >= 0 && ix < IW;
          const float* q = x + (((n * IH + (ok ? iy : 0)) * IW + (ok ? ix : 0)) * CW) + c0;
          in[dy][dx] = *reinterpret_cast<const f4a*>(q) * (ok ? 1.f : 0.f);
        }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wc = w + (long)(c0 + e) * 48;  // [ci][co][ky][kx]
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const float v = in[py - a + 1][px - b + 1][e];
                const int ky = 1 - py + 2 * a, kx = 1 - px + 2 * b;
#pragma unroll
                for (int co = 0; co < 3; ++co) acc[py][px][co] += v * wc[(co * 4 + ky) * 4 + kx];
              }
      }
    }
    const float b0 = (bias ? bias[0] : 0.f) + out_add, b1 = (bias ? bias[1] : 0.f) + out_add,
                b2 = (bias ? bias[2] : 0.f) + out_add;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float* o = y + (((n * OH + 2 * y2 + py) * OW) + 2 * x2) * 3;  // 6 contiguous floats: px = 0, 1
      float v[6] = {acc[py][0][0] + b0, acc[py][0][1] + b1, acc[py][0][2] + b2,
                    acc[py][1][0] + b0, acc[py][1][1] + b1, acc[py][1][2] + b2};
#pragma unroll
      for (int e = 0; e < 6; ++e) o[e] = accumulate ? (o[e] + v[e]) : v[e];
    }
  }
}

// The same layer on the matrix cores: an input pixel (iy, ix) contributes in[iy, ix, :] (CW) x Wm[CW][48] to the
// 4 x 4 x 3 output patch at (2 iy - 1, 2 ix - 1) -- Wm[ci][(co, ky, kx)] is the ConvTranspose2d weight as stored.  That is
// a dense GEMM M = pixels, K = CW, N = 48 (three 16-column MFMA blocks, no padding of the 3 output channels to 16) whose
// A rows are the NHWC pixels themselves (one 16-byte global load per lane per 16-k chunk, all of a wave's row blocks
// in flight together, no staging), followed by the overlap-add of the patches: a workgroup owns a 16 x 16 input tile
// (+ one pixel of halo: 324 GEMM rows in 21 row blocks over its 4 waves), writes the products P[pixel][48] to LDS
// and every thread then GATHERS the four terms of each of its 12 outputs of the 32 x 32 x 3 tile, stored as 384-byte
// rows.  (Adding the patches into an LDS tile with ds_add_f32 instead measured 266 us: LDS float atomics retire about
// two lanes per clock; the thread-per-pixel VALU form above: 178 us for 1024 frames; HBM floor of the layer: ~25 us.)
template <int CW>
__global__ __launch_bounds__(256) void convT_s2_c3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float out_add,
                                                               float* __restrict__ y, int Nimg, int IH, int IW,
                                                               int accumulate) {
  constexpr int NCH = CW / 16;
  constexpr int RB = (18 * 18 + 15) / 16;  // 21 row blocks of 16 tile pixels (halo included)
  constexpr int LDP = 52;                  // row stride of P: the four 4-row groups of a store land 16 banks apart
  __shared__ float P[RB * 16 * LDP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int tx_n = IW / 16, ty_n = IH / 16;
  const int img = blockIdx.x / (tx_n * ty_n), trem = blockIdx.x % (tx_n * ty_n);
  const int ty = trem / tx_n, tx = trem % tx_n;
  const float* xin = x + (long)img * IH * IW * CW;
  // B fragments: Wm[k = 16 c + 4 q + g][n = 16 b + i]
  f32x4 bf[NCH][3];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) bf[c][b][g] = w[(long)(16 * c + 4 * q + g) * 48 + 16 * b + i];
  constexpr int RPW = (RB + 3) / 4;        // row blocks per wave (6), loaded JB at a time: ONE memory round trip per batch
  constexpr int JB = NCH <= 2 ? RPW : 2;
  for (int j0 = 0; j0 < RPW; j0 += JB) {
    f32x4 af[JB][NCH];
    float msk[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int pl = (wave + 4 * (j0 + j)) * 16 + i;  // the pixel this lane loads
      const int ly = pl / 18, lx = pl % 18;
      const int gy = ty * 16 + ly - 1, gx = tx * 16 + lx - 1;
      const bool ok = pl < 18 * 18 && gy >= 0 && gy < IH && gx >= 0 && gx < IW;
      const float* src = xin + ((long)(ok ? gy : 0) * IW + (ok ? gx : 0)) * CW + 4 * q;
#pragma unroll
      for (int c = 0; c < NCH; ++c) af[j][c] = *reinterpret_cast<const f32x4u*>(src + 16 * c);
      msk[j] = ok ? 1.f : 0.f;
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int rbk = wave + 4 * (j0 + j);
      if (rbk >= RB) continue;  // wave-uniform
      f32x4 acc[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int b = 0; b < 3; ++b)
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][c][g] * msk[j], bf[c][b][g], acc[b], 0, 0, 0);
      // acc[b][r]: tile pixel rbk * 16 + 4 q + r, column 16 b + i
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < 3; ++b) P[(rbk * 16 + 4 * q + r) * LDP + 16 * b + i] = acc[b][r];
    }
  }
  __syncthreads();
  // output (oy, ox, c) of the tile = sum over its two (ky, ly) x two (kx, lx) source pairs: oy = 2 ly - 3 + ky
  const int OH = 2 * IH, OW = 2 * IW;
  const float bb[3] = {(bias ? bias[0] : 0.f) + out_add, (bias ? bias[1] : 0.f) + out_add, (bias ? bias[2] : 0.f) + out_add};
#pragma unroll 4
  for (int f = tid; f < 32 * 32 * 3; f += 256) {
    const int oy = f / 96, col = f % 96;
    const int ox = col / 3, c = col % 3;
    const int ky0 = (oy + 1) & 1, kx0 = (ox + 1) & 1;
    float v = bb[c];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int ky = ky0 + 2 * dy, kx = kx0 + 2 * dx;
        const int ly = (oy + 3 - ky) >> 1, lx = (ox + 3 - kx) >> 1;
        v += P[(ly * 18 + lx) * LDP + c * 16 + ky * 4 + kx];
      }
    float* o = y + (((long)img * OH + 32 * ty + oy) * OW + 32 * tx) * 3 + col;
    *o = accumulate ? (*o + v) : v;
  }
}

// Weight gradient of the two image-side layers (encoder Conv2d 3 -> CW: coarse = dY, fine = image; decoder
// ConvTranspose2d CW -> 3: coarse = layer input, fine = dOut), straight into the reference layout dw[CW][3][4][4]:
//   dw[cc][(c, ky, kx)] += sum over coarse pixels (n, cy, cx) of coarse[n, cy, cx, cc] * fine[n, 2 cy - 1 + ky, 2 cx - 1 + kx, c]
// = a GEMM M = CW, N = 48, K = coarse pixels.  HBM-side (184 MB per launch at 1024 frames against 3.2 GFLOP): a
// workgroup walks 16 x 16 coarse-pixel tiles (grid-stride), stages the tile [256][CW] (row stride CW + 16: the four
// k-lanes of a ds_read_b32 fragment read land in four different 16-bank windows) and the 34 x 34 x 3 fine window in LDS,
// its 4 waves split the 256 pixels (16 k-steps x CW/16 x 3 MFMAs each), the NEXT tile's loads are in flight in
// registers meanwhile, and the accumulators live across all the tiles of the workgroup: one LDS reduction over the
// waves and CW x 48 atomics per workgroup at the end (no packed scratch, no unpack launch).
template <int CW>
__global__ __launch_bounds__(256) void conv_wgrad_c3_kernel(const float* __restrict__ coarse,
                                                            const float* __restrict__ fine, float* __restrict__ dw,
                                                            int Nimg, int H, int W) {
  constexpr int MA = CW / 16, LDC = CW + 16;
  constexpr int CV = 256 * (CW / 4) / 256;            // float4 of the coarse tile per thread
  constexpr int FN = 34 * 34 * 3, FV = (FN + 255) / 256;  // floats of the fine window (per thread)
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  float* cs = wlds;               // [256][LDC]
  float* fs = wlds + 256 * LDC;   // [34 * 34 * 3]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int OH = H / 2, OW = W / 2;
  const int tx_n = OW / 16, ty_n = OH / 16;
  const int tiles = Nimg * tx_n * ty_n;
  f32x4 cv[CV];
  float fv[FV];
  auto gload = [&](int tile) {
    const int img = tile / (tx_n * ty_n), trem = tile % (tx_n * ty_n);
    const int ty = trem / tx_n, tx = trem % tx_n;
    const float* cbase = coarse + (((long)img * OH + ty * 16) * OW + tx * 16) * CW;
#pragma unroll
    for (int j = 0; j < CV; ++j) {
      const int f = tid + 256 * j;
      const int pix = f / (CW / 4), c4 = (f % (CW / 4)) * 4;
      cv[j] = *reinterpret_cast<const f32x4u*>(cbase + ((long)(pix >> 4) * OW + (pix & 15)) * CW + c4);
    }
    const float* fbase = fine + (long)img * H * W * 3;
#pragma unroll
    for (int j = 0; j < FV; ++j) {
      const int f = tid + 256 * j;
      const int fy = f / 102, fr = f % 102;  // 102 = 34 * 3 floats per window row
      const int gy = 32 * ty - 1 + fy, gxc = (32 * tx - 1) * 3 + fr;
      const bool ok = f < FN && gy >= 0 && gy < H && gxc >= 0 && gxc < 3 * W;
      fv[j] = ok ? fbase[(long)gy * W * 3 + gxc] : 0.f;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < CV; ++j) {
      const int f = tid + 256 * j;
      *reinterpret_cast<f32x4*>(&cs[(f / (CW / 4)) * LDC + (f % (CW / 4)) * 4]) = cv[j];
    }
#pragma unroll
    for (int j = 0; j < FV; ++j) {
      const int f = tid + 256 * j;
      if (f < FN) fs[f] = fv[j];
    }
  };
  f32x4 acc[MA][3];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // this lane's B columns n = 16 b + i -> (c, ky, kx): offset of the patch element inside the fine window
  int boff[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int n = 16 * b + i;
    boff[b] = (((n >> 2) & 3) * 34 + (n & 3)) * 3 + (n >> 4);
  }
  int tile = blockIdx.x;
  if (tile < tiles) gload(tile);
  for (; tile < tiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's readers are done
    lstore();
    __syncthreads();
    if (tile + (int)gridDim.x < tiles) gload(tile + gridDim.x);  // in flight during the MFMAs below
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const int pk = wave * 64 + 4 * s + q;  // this lane's k (pixel of the tile) for the step
      const int cy = pk >> 4, cx = pk & 15;
      float av[MA], bv[3];
#pragma unroll
      for (int a = 0; a < MA; ++a) av[a] = cs[pk * LDC + 16 * a + i];
      const int fo = (2 * cy * 34 + 2 * cx) * 3;
#pragma unroll
      for (int b = 0; b < 3; ++b) bv[b] = fs[fo + boff[b]];
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  // reduce the four waves' CW x 48 partial sums through LDS (reusing cs), then one atomic per element
  __syncthreads();
  float* red = cs;  // [4][MA * 3 * 256]
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * MA * 3 + a * 3 + b) * 256 + r * 64 + lane] = acc[a][b][r];
  __syncthreads();
  for (int e = tid; e < MA * 3 * 256; e += 256) {
    const int blk = e >> 8, x = e & 255;
    const int a = blk / 3, b = blk % 3;
    const int r = x >> 6, l = x & 63;
    const int cc = 16 * a + 4 * (l >> 4) + r, n = 16 * b + (l & 15);
    const float v = red[e] + red[MA * 3 * 256 + e] + red[2 * MA * 3 * 256 + e] + red[3 * MA * 3 * 256 + e];
    atomicAdd(dw + (long)cc * 48 + n, v);
  }
}

// Weight gradient of the narrow layer pair next to the image layers (fine 32 channels, coarse 64: encoder layer 2 /
// decoder layer 3 at cnn_depth 32), the same tile walk with the WHOLE 64 x 512 gradient in the accumulators of one
// workgroup: on the split-K 64 x 64 tiles every column tile re-reads the coarse rows and re-gathers its two taps of
// the fine rows from L2 for 262 KFLOP per 16 KB (56 TFLOP/s, 39 % MFMA-busy, 307 us at 1024 frames).  Here a workgroup
// (8 waves, two per SIMD) walks 8 x 8 coarse-pixel tiles: the tile [64][CC] (row stride CC + 16) and its 18 x 18 x CF
// fine window (row stride CF + 8: the four k-lanes of a fragment read are two fine pixels apart = 16 banks) are
// staged once and serve all 16 taps; wave w owns taps 2 w, 2 w + 1 (4 column blocks) x all CC rows: per k-step of
// 4 pixels 4 + 4 ds_read_b32 feed 16 MFMAs; the next tile's loads are in flight in registers.  The per-workgroup
// partial sums go to partial[workgroup][CC][16 CF] and conv_wgrad_tile_reduce_kernel adds them into the reference
// layout dw[CC][CF][4][4] (two-stage: 32 K atomics per workgroup would serialise at the fabric).
template <int CF, int CC>
__global__ __launch_bounds__(512) void conv_wgrad_tile_kernel(const float* __restrict__ coarse,
                                                              const float* __restrict__ fine, float* __restrict__ partial,
                                                              int Nimg, int H, int W) {
  constexpr int NT = 512, MA = CC / 16, NH = CF / 16, LDC = CC + 16, LDF = CF + 8;
  constexpr int CVN = 64 * (CC / 4), CV = (CVN + NT - 1) / NT;        // float4 of the coarse tile (per thread)
  constexpr int FVN = 18 * 18 * (CF / 4), FV = (FVN + NT - 1) / NT;   // float4 of the fine window (per thread)
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  float* cs = wlds;             // [64][LDC]
  float* fs = wlds + 64 * LDC;  // [18 * 18][LDF]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int OH = H / 2, OW = W / 2;
  const int tx_n = OW / 8, ty_n = OH / 8;
  const int tiles = Nimg * tx_n * ty_n;
  f32x4 cv[CV], fv[FV];
  auto gload = [&](int tile) {
    const int img = tile / (tx_n * ty_n), trem = tile % (tx_n * ty_n);
    const int ty = trem / tx_n, tx = trem % tx_n;
    const float* cbase = coarse + (((long)img * OH + ty * 8) * OW + tx * 8) * CC;
#pragma unroll
    for (int j = 0; j < CV; ++j) {
      const int f = tid + NT * j;
      const int pix = (f < CVN ? f : 0) / (CC / 4), c4 = (f % (CC / 4)) * 4;
      cv[j] = *reinterpret_cast<const f32x4u*>(cbase + ((long)(pix >> 3) * OW + (pix & 7)) * CC + c4);
    }
    const float* fbase = fine + (long)img * H * W * CF;
#pragma unroll
    for (int j = 0; j < FV; ++j) {
      const int f = tid + NT * j;
      const int pix = f / (CF / 4), c4 = (f % (CF / 4)) * 4;
      const int fy = pix / 18, fx = pix % 18;
      const int gy = 16 * ty - 1 + fy, gx = 16 * tx - 1 + fx;
      const bool ok = f < FVN && gy >= 0 && gy < H && gx >= 0 && gx < W;
      fv[j] = *reinterpret_cast<const f32x4u*>(fbase + ((long)(ok ? gy : 0) * W + (ok ? gx : 0)) * CF + c4);
      if (!ok) fv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < CV; ++j) {
      const int f = tid + NT * j;
      if (f < CVN) *reinterpret_cast<f32x4*>(&cs[(f / (CC / 4)) * LDC + (f % (CC / 4)) * 4]) = cv[j];
    }
#pragma unroll
    for (int j = 0; j < FV; ++j) {
      const int f = tid + NT * j;
      if (f < FVN) *reinterpret_cast<f32x4*>(&fs[(f / (CF / 4)) * LDF + (f % (CF / 4)) * 4]) = fv[j];
    }
  };
  f32x4 acc[MA][2 * NH];  // column block nb = 2-tap index * NH + channel half
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < 2 * NH; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // this wave's taps: 2 wave + tt -> (ky, kx); fine-window offset of the tap for column block b
  int boff[2 * NH];
#pragma unroll
  for (int b = 0; b < 2 * NH; ++b) {
    const int tap = 2 * wave + b / NH;
    boff[b] = ((tap >> 2) * 18 + (tap & 3)) * LDF + 16 * (b % NH) + i;
  }
  int tile = blockIdx.x;
  if (tile < tiles) gload(tile);
  for (; tile < tiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's readers are done
    lstore();
    __syncthreads();
    if (tile + (int)gridDim.x < tiles) gload(tile + gridDim.x);  // in flight during the MFMAs below
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const int pk = 4 * s + q;  // this lane's k (coarse pixel of the tile) for the step
      const int cy = pk >> 3, cx = pk & 7;
      float av[MA], bv[2 * NH];
#pragma unroll
      for (int a = 0; a < MA; ++a) av[a] = cs[pk * LDC + 16 * a + i];
      const int fo = (2 * cy * 18 + 2 * cx) * LDF;
#pragma unroll
      for (int b = 0; b < 2 * NH; ++b) bv[b] = fs[fo + boff[b]];
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < 2 * NH; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  // partial[blockIdx.x][cc][tap * CF + cf]
  float* out = partial + (long)blockIdx.x * CC * 16 * CF;
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < 2 * NH; ++b) {
      const int tap = 2 * wave + b / NH, cf = 16 * (b % NH) + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(16 * a + 4 * q + r) * (16 * CF) + tap * CF + cf] = acc[a][b][r];
    }
}

// dw[cc][cf][ky][kx] += sum_g partial[g][cc][(ky, kx, cf)]
__global__ __launch_bounds__(256) void conv_wgrad_tile_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                     int G, int CC, int CF) {
  const int total = CC * 16 * CF;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int g = 0;
  for (; g + 3 < G; g += 4) {
    s0 += partial[(long)g * total + e];
    s1 += partial[(long)(g + 1) * total + e];
    s2 += partial[(long)(g + 2) * total + e];
    s3 += partial[(long)(g + 3) * total + e];
  }
  for (; g < G; ++g) s0 += partial[(long)g * total + e];
  const int cf = e % CF, tap = (e / CF) % 16, cc = e / (16 * CF);
  dw[(((long)cc * CF + cf) * 4 + (tap >> 2)) * 4 + (tap & 3)] += (s0 + s1) + (s2 + s3);
}

using C128 = TileShape<2, 2, 2, 2, 16>;    // 128 x 128
using C128K32 = TileShape<2, 2, 2, 2, 32>;  // 128 x 128, BK 32 (twice the loads in flight per barrier)
using C64 = TileShape<2, 2, 1, 1, 32>;     // 64 x 64
using C128x32 = TileShape<4, 1, 1, 1, 32>;  // 128 x 32 (narrow channel counts)
using C32x64S = TileShape<1, 2, 1, 1, 64, 2>;  // 32 x 64, K split over two wave-groups (first/last layer wgrad)

}  // namespace dv3

using namespace dv3;

static bool pow2_spatial(int H, int W) { return H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0; }

extern "C" int dv3_pack_conv_weight(const float* w, float* wp, int Co, int Ci, int transposed, void* stream) {
  if (!w || !wp || Co <= 0 || Ci <= 0) return DV3_ERR_ARG;
  const long total = 16L * Co * Ci;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (transposed) hipLaunchKernelGGL(pack_convT_w_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Ci, Co);
  else hipLaunchKernelGGL(pack_conv_w_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Co, Ci);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// conv_s2 as an implicit GEMM on the k-contiguous LDS tile scheme of gemm_l16_kernel (gemm.hip): M = output pixels,
// N = Co, K = (ky, kx, ci).  With Ci % 32 == 0 a 32-wide K-tile is 32 consecutive channels of ONE tap, i.e. 128
// contiguous bytes of the NHWC input per output pixel: the A staging is the dense loader with a per-tile row pointer
// (tap offset, zero for padding taps) and both operands sit k-contiguous in LDS (stride 40 floats), so an MFMA
// fragment for four k-steps is one ds_read_b128 -- against the k-major image of the 32x32x2 tile engine above that is
// a quarter of the LDS instructions and no transposing ds_write_b32.  4 waves (2 x 2), v_mfma_f32_16x16x4_f32,
// double-buffered LDS, global loads of tile t+1 in flight during the MFMAs of tile t; padding is zeroed when the
// tile is written to LDS (not when it is loaded: that would make the wave wait for its own prefetch).
// ------------------------------------------------------------------------------------------------
// TR = false: conv_s2 (rows = output pixels, K = 16 Ci, taps (ky, kx), input at (2oy-1+ky, 2ox-1+kx)).
// TR = true : convT_s2, one parity class (py, px) = blockIdx.y per grid row (rows = INPUT pixels (y, x), K = 4 Ci, taps
//             (a, b), input at (y+py-a, x+px-b), weights wp[cls][Co][4 Ci], output pixel (2y+py, 2x+px); + bias, + out_add).
template <int BM, int BN, bool TR>
__global__ __launch_bounds__(256) void conv_s2_l16_kernel(ConvParams p) {
  constexpr int BK = 32, LD = 40;
  constexpr int WM = 2, WN = 2;                // waves: 2 x 2
  constexpr int TM = BM / 32, TN = BN / 32;    // 16 x 16 blocks per wave
  constexpr int NA = BM * (BK / 4) / 256, NB = BN * (BK / 4) / 256;
  static_assert(BM % 32 == 0 && BN % 32 == 0 && NA >= 1 && NB >= 1, "tile");
  __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  static_assert(WM * WN == 4, "four waves");
  int cls = 0, lin = blockIdx.x;
  if constexpr (TR) {
    if (!convT_tile_class(p.tiles_m * p.tiles_n, lin, cls)) return;  // whole workgroup, before any barrier
  }
  const int py = cls >> 1, px = cls & 1;
  const int RH = TR ? p.H : (p.H >> 1), RW = TR ? p.W : (p.W >> 1);  // the grid of pixels the rows enumerate
  const long M = (long)p.Nimg * RH * RW;
  const int K = (TR ? 4 : 16) * p.Ci;
  // conv: an XCD owns a contiguous range of pixel tiles (its share of x is private to its L2; the weights are small);
  // convT: the four classes of a tile share an XCD (convT_tile_class)
  if constexpr (!TR) {
    const int tiles = p.tiles_m * p.tiles_n;
    if ((p.tiles_m & 7) == 0) lin = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
  }
  const int tm = lin / p.tiles_n, tn = lin % p.tiles_n;
  const long m0 = (long)tm * BM;
  const int n0 = tn * BN;
  const int c4 = (tid & 7) * 4;
  const float* abase[NA];
  int iy0[NA], ix0[NA];
  bool rowok[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const long m = m0 + ((tid + 256 * j) >> 3);
    rowok[j] = m < M;
    const long mm = rowok[j] ? m : 0;
    const int rx = (int)(mm % RW);
    const long t = mm / RW;
    const int ry = (int)(t % RH);
    const long n = t / RH;
    iy0[j] = TR ? ry + py : 2 * ry - 1;
    ix0[j] = TR ? rx + px : 2 * rx - 1;
    abase[j] = p.x + n * p.H * p.W * p.Ci + c4;
  }
  const float* bsrc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int col = n0 + ((tid + 256 * j) >> 3);
    bsrc[j] = p.wp + (long)cls * p.Co * K + (long)(col < p.Co ? col : 0) * K + c4;
  }
  f32x4 ra[NA], rb[NB];
  bool aok[NA];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int k0) {
    const int tap = k0 / p.Ci, ci0 = k0 - tap * p.Ci;
    const int ty = TR ? -(tap >> 1) : (tap >> 2), tx = TR ? -(tap & 1) : (tap & 3);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int iy = iy0[j] + ty, ix = ix0[j] + tx;
      aok[j] = rowok[j] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const float* src = aok[j] ? abase[j] + ((long)iy * p.W + ix) * p.Ci + ci0 : p.x;
      ra[j] = *reinterpret_cast<const f32x4*>(src);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = *reinterpret_cast<const f32x4*>(bsrc[j] + k0);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      *reinterpret_cast<f32x4*>(&As[buf][((tid + 256 * j) >> 3) * LD + c4]) = aok[j] ? ra[j] : zero4;
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(&Bs[buf][((tid + 256 * j) >> 3) * LD + c4]) = rb[j];
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = zero4;
  const int nk = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int aoff = (wm * (16 * TM) + i) * LD + 4 * q;
  const int boff = (wn * (16 * TN) + i) * LD + 4 * q;
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) gload((t + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const f32x4*>(&As[cur][aoff + 16 * a * LD + 16 * kk]);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const f32x4*>(&Bs[cur][boff + 16 * b * LD + 16 * kk]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][g], bf[b][g], acc[a][b], 0, 0, 0);
    }
    if (t + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wn * (16 * TN) + 16 * b + i;
      if (col >= p.Co) continue;
      const float add = TR ? (p.bias ? p.bias[col] : 0.f) + p.out_add : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * (16 * TM) + 16 * a + 4 * q + r;
        if (row < M) {
          float* o;
          if (TR) {
            const int x2 = (int)(row % p.W);
            const long t = row / p.W;
            const int y2 = (int)(t % p.H);
            const long img = t / p.H;
            o = p.y + ((img * (2 * p.H) + 2 * y2 + py) * (2L * p.W) + 2 * x2 + px) * p.Co + col;
          } else {
            o = p.y + row * p.Co + col;
          }
          float v = acc[a][b][r] + add;
          if (p.accumulate) v += *o;
          *o = v;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Narrow transposed convolution, one 16 x 16 INPUT-pixel tile of one image per workgroup, all four parity classes:
// the decoder layer in front of the image layer (64 -> 32 channels at 16 x 16 -> 32 x 32) and the encoder's second
// layer read backwards.  With 32 output channels the register-direct kernel above re-reads every gathered input
// element and the whole 32 KB weight image per wave from L1 / L2 (8 flop per cache byte: 63 TFLOP/s, 45 % MFMA-busy).
// Here the tile and its one-pixel halo (18 x 18 x Ci floats, channel-contiguous, row stride Ci + 8: the
// ds_read_b128 fragment reads of 16 pixels x 4 k-quads cover the 64 banks evenly) are staged in LDS ONCE and serve the
// 4 taps x 4 classes = 16 gathers of every pixel; a class's weights [32][4 Ci] (stride 4 Ci + 8) sit beside it, the
// next class's are prefetched into registers while the current class multiplies.  A wave owns 2 input rows (2 x 16
// pixels) x 32 channels: per 16-k chunk 2 + 2 ds_read_b128 feed 16 v_mfma_f32_16x16x4_f32.  ~80 flop per byte
// entering the CU.  CI = 16 .. 64 (multiple of 16), Co <= 32; IH, IW multiples of 16.
// grid = Nimg * (IH / 16) * (IW / 16) workgroups of 8 waves (two per SIMD: one wave's LDS waits and stores are covered
// by the other's MFMAs; the LDS footprint allows one workgroup per CU).
// ------------------------------------------------------------------------------------------------
template <int CI>
__global__ __launch_bounds__(512) void convT_s2_tile_kernel(ConvParams p) {
  constexpr int NT = 512, RPW = 2;  // 8 waves (two per SIMD), 2 input rows each
  constexpr int LDX = CI + 8, K = 4 * CI, LDW = K + 8, NCH = CI / 16;
  constexpr int XV = 18 * 18 * (CI / 4);     // float4 of the input tile
  constexpr int XJ = (XV + NT - 1) / NT;     // ... per thread
  constexpr int WV = 32 * (K / 4) / NT;      // float4 of one class's weights per thread
  extern __shared__ __attribute__((aligned(16))) float tlds[];
  float* xs = tlds;                  // [18 * 18][LDX]
  float* ws = tlds + 18 * 18 * LDX;  // [32][LDW]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int tx_n = p.W / 16, ty_n = p.H / 16;
  const int img = blockIdx.x / (tx_n * ty_n), trem = blockIdx.x % (tx_n * ty_n);
  const int ty = trem / tx_n, tx = trem % tx_n;
  const float* xin = p.x + (long)img * p.H * p.W * CI;
  // weights of class `cls` for this thread: float4 f = tid + NT j -> channel f / (K / 4), k offset 4 (f % (K / 4));
  // channels >= Co read channel 0 (their outputs are never stored)
  f32x4 wr[WV];
  auto wload = [&](int cls) {
#pragma unroll
    for (int j = 0; j < WV; ++j) {
      const int f = tid + NT * j;
      const int co = f / (K / 4), k4 = (f % (K / 4)) * 4;
      wr[j] = *reinterpret_cast<const f32x4u*>(p.wp + ((long)cls * p.Co + (co < p.Co ? co : 0)) * K + k4);
    }
  };
  auto wstore = [&]() {
#pragma unroll
    for (int j = 0; j < WV; ++j) {
      const int f = tid + NT * j;
      *reinterpret_cast<f32x4*>(&ws[(f / (K / 4)) * LDW + (f % (K / 4)) * 4]) = wr[j];
    }
  };
  wload(0);
  // ---- stage the input tile + halo (zero outside the image): every load of the thread in flight before the first
  // LDS write (a load-store-load-store loop pays the memory latency XJ times)
  {
    f32x4 xv[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int f = tid + NT * j;
      const int pix = f / (CI / 4), c4 = (f % (CI / 4)) * 4;
      const int ly = pix / 18, lx = pix % 18;
      const int gy = ty * 16 + ly - 1, gx = tx * 16 + lx - 1;
      const bool ok = f < XV && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      xv[j] = *reinterpret_cast<const f32x4u*>(xin + ((long)(ok ? gy : 0) * p.W + (ok ? gx : 0)) * CI + c4);
      if (!ok) xv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int f = tid + NT * j;
      if (f < XV) *reinterpret_cast<f32x4*>(&xs[(f / (CI / 4)) * LDX + (f % (CI / 4)) * 4]) = xv[j];
    }
  }
  wstore();
  __syncthreads();
  const int OH = 2 * p.H, OW = 2 * p.W;
  const float bias0 = (p.bias && i < p.Co) ? p.bias[i] : 0.f;
  const float bias1 = (p.bias && 16 + i < p.Co) ? p.bias[16 + i] : 0.f;
  for (int cls = 0; cls < 4; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    if (cls < 3) wload(cls + 1);  // in flight while this class multiplies
    f32x4 acc[RPW][2];
#pragma unroll
    for (int a = 0; a < RPW; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      const int ta = tap >> 1, tb = tap & 1;
      // input pixel of (row y = RPW wave + a, x = i): (y + py - ta, x + px - tb), + 1 for the halo
      const int xoff = ((RPW * wave + py - ta + 1) * 18 + (i + px - tb + 1)) * LDX + 4 * q;
      const int woff = i * LDW + tap * CI + 4 * q;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        f32x4 af[RPW], bf[2];
#pragma unroll
        for (int a = 0; a < RPW; ++a) af[a] = *reinterpret_cast<const f32x4*>(&xs[xoff + a * 18 * LDX + 16 * c]);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const f32x4*>(&ws[woff + 16 * b * LDW + 16 * c]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int a = 0; a < RPW; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][g], bf[b][g], acc[a][b], 0, 0, 0);
      }
    }
    // acc[a][b][r]: input pixel (y = RPW wave + a, x = 4 q + r), channel 16 b + i -> output pixel (2 y + py, 2 x + px)
#pragma unroll
    for (int a = 0; a < RPW; ++a) {
      const int oy = 2 * (ty * 16 + RPW * wave + a) + py;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ox = 2 * (tx * 16 + 4 * q + r) + px;
        float* orow = p.y + (((long)img * OH + oy) * OW + ox) * p.Co;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int co = 16 * b + i;
          if (co < p.Co) {
            const float val = acc[a][b][r] + (b ? bias1 : bias0) + p.out_add;
            orow[co] = p.accumulate ? (orow[co] + val) : val;
          }
        }
      }
    }
    if (cls < 3) {
      __syncthreads();  // every wave is done with this class's weights
      wstore();
      __syncthreads();
    }
  }
}


// Explicit im2col of Conv2d k4 s2 "same" (networks.py:771-798) for few-image batches -- the acting step (SURVEY
// 8(f) N1) runs the encoder on 1-16 images, where the tiled implicit-GEMM kernels have 1-16 workgroups walking the
// whole reduction (30-77 us per layer at one image); cols [N*OH*OW][(ci,ky,kx)] is in the reference weight's own
// order, so the product is a plain y = cols W.view(Co, 16 Ci)^T on the few-row / register-direct GEMM kernels.
__global__ void im2col_s2_kernel(const float* __restrict__ x, float* __restrict__ cols, int Nimg, int H, int W, int C) {
  const int OH = H >> 1, OW = W >> 1;
  const long K = 16L * C, total = (long)Nimg * OH * OW * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % K);
    long m = e / K;
    const int kx = k & 3, ky = (k >> 2) & 3, ci = k >> 4;
    const int ox = (int)(m % OW);
    m /= OW;
    const int oy = (int)(m % OH);
    const long n = m / OH;
    const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
    cols[e] = ok ? x[((n * H + iy) * W + ix) * C + ci] : 0.f;
  }
}

extern "C" int dv3_im2col_s2(const float* x, float* cols, int Nimg, int H, int W, int C, void* stream) {
  if (Nimg <= 0) return 0;
  if (!x || !cols || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return DV3_ERR_ARG;
  const long total = (long)Nimg * (H / 2) * (W / 2) * 16 * C;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(im2col_s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, cols, Nimg, H, W, C);
  return (int)hipGetLastError();
}

extern "C" int dv3_conv_s2_fwd(const float* x, const float* w_packed, float* y, int Nimg, int H, int W, int Ci, int Co,
                               int accumulate, void* stream) {
  if (Nimg <= 0) return 0;
  if (!x || !w_packed || !y || Ci <= 0 || Co <= 0 || !pow2_spatial(H, W)) return DV3_ERR_ARG;
  ConvParams p{x, w_packed, y, nullptr, 0.f, Nimg, H, W, Ci, Co, 0, 0, accumulate};
  const long M = (long)Nimg * (H / 2) * (W / 2);
  if (M > 0x7fffffffL - 256) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  auto go = [&](auto ts) {
    using TS = decltype(ts);
    p.tiles_m = (int)((M + TS::BM - 1) / TS::BM);
    p.tiles_n = (Co + TS::BN - 1) / TS::BN;
    hipLaunchKernelGGL((conv_s2_kernel<TS>), dim3(p.tiles_m * p.tiles_n), dim3(kThreads), 0, s, p);
  };
  // 128x128 tiles need more than one workgroup per CU to pay for themselves: the deepest encoder layer
  // (16k rows x 256 channels = 256 such tiles) runs 14 % faster on 64x64 tiles (1024 workgroups)
  const long tiles128 = ((M + 127) / 128) * ((Co + 127) / 128);
  // measured (tools/conv_bench.py, 1024 frames, us): 32->64 213 -> 193, 64->128 196 -> 163, 128->256 207 -> 159 against
  // the k-major 32x32x2 tiles / the register-direct kernel
  static const int env_l16 = DV3_ENV_INT("DV3_CONV_L16", 1);
  if (env_l16 && (Ci % 32) == 0 && Co >= 64 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0) {
    // column tile: 128 where Co is a multiple of it, 96 for the crafter widths (96 / 192: cnn_depth 96), else 64; row
    // tile 128 once that still gives every CU ~2 workgroups.  Measured (tools/conv_bench.py, us): 4096 frames depth 96,
    // 192->384: 64x128 5519, 128x128 5221; 1024 frames 64->128: 179 -> 169, 128->256 (256 tiles of 128x128): 159 -> 168.
    const int bn = (env_l16 != 3 && (Co % 128) == 0) ? 128 : (env_l16 != 3 && (Co % 96) == 0) ? 96 : 64;
    const bool big = bn != 64 && (env_l16 == 4 || (env_l16 == 1 && ((M + 127) / 128) * (Co / bn) >= 448));
    const int bm = big ? 128 : 64;
    p.tiles_m = (int)((M + bm - 1) / bm);
    p.tiles_n = (Co + bn - 1) / bn;
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
    if (bn == 128 && big) hipLaunchKernelGGL((conv_s2_l16_kernel<128, 128, false>), grid, dim3(256), 0, s, p);
    else if (bn == 128) hipLaunchKernelGGL((conv_s2_l16_kernel<64, 128, false>), grid, dim3(256), 0, s, p);
    else if (bn == 96 && big) hipLaunchKernelGGL((conv_s2_l16_kernel<128, 96, false>), grid, dim3(256), 0, s, p);
    else if (bn == 96) hipLaunchKernelGGL((conv_s2_l16_kernel<64, 96, false>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_s2_l16_kernel<64, 64, false>), grid, dim3(256), 0, s, p);
    return (int)hipGetLastError();
  }
  static const int env_direct = DV3_ENV_INT("DV3_CONV_DIRECT", 64);
  if ((Ci & 3) == 0 && Co <= env_direct && Co <= 128) {
    const dim3 grid((unsigned)((M + 127) / 128));
    if (Co <= 32) hipLaunchKernelGGL(conv_s2_direct_kernel<2>, grid, dim3(256), 0, s, p);
    else if (Co <= 64) hipLaunchKernelGGL(conv_s2_direct_kernel<4>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_s2_direct_kernel<8>, grid, dim3(256), 0, s, p);
    return (int)hipGetLastError();
  }
  if (Co <= 32) go(C128x32{});
  else if (Co <= 64 || tiles128 <= 256) go(C64{});
  else go(C128{});
  return (int)hipGetLastError();
}

extern "C" int dv3_convT_s2_fwd(const float* x, const float* w_packed, const float* bias, float out_add, float* y,
                                int Nimg, int IH, int IW, int Ci, int Co, int accumulate, void* stream) {
  if (Nimg <= 0) return 0;
  if (!x || !w_packed || !y || Ci <= 0 || Co <= 0 || IH <= 0 || IW <= 0) return DV3_ERR_ARG;
  ConvParams p{x, w_packed, y, bias, out_add, Nimg, IH, IW, Ci, Co, 0, 0, accumulate};
  const long M = (long)Nimg * IH * IW;
  if (M > 0x7fffffffL - 256) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  auto go = [&](auto ts) {
    using TS = decltype(ts);
    p.tiles_m = (int)((M + TS::BM - 1) / TS::BM);
    p.tiles_n = (Co + TS::BN - 1) / TS::BN;
    hipLaunchKernelGGL((convT_s2_kernel<TS>), dim3(p.tiles_m * p.tiles_n, 4), dim3(kThreads), 0, s, p);
  };
  static const int env_tile = DV3_ENV_INT("DV3_CONVT_TILE", 1);
  if (env_tile && Co <= 32 && (Ci == 16 || Ci == 32 || Ci == 48 || Ci == 64) && (IH % 16) == 0 && (IW % 16) == 0) {
    // one 16 x 16 input tile per workgroup, all four classes (measured, 1024 frames 64 -> 32: see DESIGN.md)
    const dim3 grid((unsigned)((long)Nimg * (IH / 16) * (IW / 16)));
    const size_t lds = (size_t)(18 * 18 * (Ci + 8) + 32 * (4 * Ci + 8)) * sizeof(float);
#define DV3_CT(CI_)                                                                                              \
  {                                                                                                              \
    static unsigned long long seen[4] = {0, 0, 0, 0};                                                            \
    if (dv3_first_on_device(seen))                                                                               \
      (void)hipFuncSetAttribute((const void*)convT_s2_tile_kernel<CI_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((convT_s2_tile_kernel<CI_>), grid, dim3(512), lds, s, p);                                  \
  }
    if (Ci == 16) DV3_CT(16) else if (Ci == 32) DV3_CT(32) else if (Ci == 48) DV3_CT(48) else DV3_CT(64)
#undef DV3_CT
    return (int)hipGetLastError();
  }
  static const int env_l16 = DV3_ENV_INT("DV3_CONVT_L16", 1);
  // measured (tools/conv_bench.py, 1024 frames, us): Co 32: direct 247 / l16 278; Co 64: 195 / 195; Co 128: 183 / 178 --
  // the register-direct kernels keep the narrow layers, the LDS tile takes Co >= 128 (and everything wider, which
  // used to run on the k-major 32x32x2 tiles)
  // (r02, crafter widths, 4096 frames: Co 96 on a 128x96 tile against the register-direct kernel)
  const bool co96 = (Co % 96) == 0 && (Co % 128) != 0 && env_l16 != 3;
  if (env_l16 && (Ci % 32) == 0 && (Co >= (env_l16 == 2 ? 32 : 128) || (co96 && env_l16 != 5)) &&
      ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0) {
    const int bn = (Co % 128) == 0 && env_l16 != 3 ? 128 : co96 ? 96 : (Co % 64) == 0 ? 64 : 32;
    // rows per class M; 128-row tiles once the four classes together still give every CU ~2 workgroups
    const bool big = (bn == 128 || bn == 96) && ((M + 127) / 128) * (Co / bn) * 4 >= 448;
    const int bm = (bn == 32 || big) ? 128 : 64;
    p.tiles_m = (int)((M + bm - 1) / bm);
    p.tiles_n = (Co + bn - 1) / bn;
    const dim3 grid(convT_grid((long)p.tiles_m * p.tiles_n));
    if (bn == 128 && big) hipLaunchKernelGGL((conv_s2_l16_kernel<128, 128, true>), grid, dim3(256), 0, s, p);
    else if (bn == 128) hipLaunchKernelGGL((conv_s2_l16_kernel<64, 128, true>), grid, dim3(256), 0, s, p);
    else if (bn == 96 && big) hipLaunchKernelGGL((conv_s2_l16_kernel<128, 96, true>), grid, dim3(256), 0, s, p);
    else if (bn == 96) hipLaunchKernelGGL((conv_s2_l16_kernel<64, 96, true>), grid, dim3(256), 0, s, p);
    else if (bn == 64) hipLaunchKernelGGL((conv_s2_l16_kernel<64, 64, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_s2_l16_kernel<128, 32, true>), grid, dim3(256), 0, s, p);
    return (int)hipGetLastError();
  }
  static const int env_direct = DV3_ENV_INT("DV3_CONVT_DIRECT", 128);
  if ((Ci & 3) == 0 && Co <= env_direct && Co <= 128) {
    const dim3 grid(convT_grid((M + 127) / 128));
    if (Co <= 32) hipLaunchKernelGGL(convT_s2_direct_kernel<2>, grid, dim3(256), 0, s, p);
    else if (Co <= 64) hipLaunchKernelGGL(convT_s2_direct_kernel<4>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(convT_s2_direct_kernel<8>, grid, dim3(256), 0, s, p);
    return (int)hipGetLastError();
  }
  if (Co <= 32) go(C128x32{});  // (a 256 x 32 tile measured 36 % slower)
  else if (Co <= 64) go(C64{});
  else go(C128{});
  return (int)hipGetLastError();
}

extern "C" int dv3_conv_s2_wgrad(const float* coarse, const float* fine, float* dw_packed, float* dw, int Nimg, int H,
                                 int W, int Cfine, int Ccoarse, void* stream) {
  if (Nimg <= 0) return 0;
  if (!coarse || !fine || !dw || !dw_packed || Cfine <= 0 || Ccoarse <= 0 || !pow2_spatial(H, W)) return DV3_ERR_ARG;
  const long rows = (long)Nimg * (H / 2) * (W / 2);
  if (rows > 0x7fffffffL - 4096) return DV3_ERR_ARG;
  static const int env_c3w = DV3_ENV_INT("DV3_C3W_TILE", 1);
  if (env_c3w && Cfine == 3 && (Ccoarse == 32 || Ccoarse == 96) && ((H / 2) % 16) == 0 && ((W / 2) % 16) == 0) {
    // image-side layers: tile walk with the accumulators in registers, straight into dw (reference layout)
    const long tiles = (long)Nimg * (H / 32) * (W / 32);
    const size_t lds = (size_t)(256 * (Ccoarse + 16) + 34 * 34 * 3) * sizeof(float);
    const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);
    if (Ccoarse == 32) {
      hipLaunchKernelGGL((conv_wgrad_c3_kernel<32>), dim3(grid), dim3(256), lds, (hipStream_t)stream, coarse, fine, dw, Nimg, H, W);
    } else {
      static unsigned long long seen[4] = {0, 0, 0, 0};
      if (dv3_first_on_device(seen))
        (void)hipFuncSetAttribute((const void*)conv_wgrad_c3_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((conv_wgrad_c3_kernel<96>), dim3(grid), dim3(256), lds, (hipStream_t)stream, coarse, fine, dw, Nimg, H, W);
    }
    return (int)hipGetLastError();
  }
  WgradParams p{coarse, fine, dw_packed, Nimg, H, W, Cfine, Ccoarse, 0, 0, 0, 0, 0};
  static const int env_group = DV3_ENV_INT("DV3_WGRAD_XCD", 1);
  auto go = [&](auto ts, long target_wgs) {
    using TS = decltype(ts);
    p.tiles_m = (Ccoarse + TS::BM - 1) / TS::BM;
    p.tiles_n = (16 * Cfine + TS::BN - 1) / TS::BN;
    const int tiles = p.tiles_m * p.tiles_n;
    // aim at target_wgs workgroups, at least 512 reduction rows each
    long splits = (target_wgs + tiles - 1) / tiles;
    const long max_splits = (rows + 511) / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long chunk = (rows + splits - 1) / splits;
    chunk = ((chunk + TS::BK - 1) / TS::BK) * TS::BK;
    splits = (rows + chunk - 1) / chunk;
    p.splits = (int)splits;
    p.chunk = (int)chunk;
    p.xcd_group = (env_group && splits >= 8) ? 1 : 0;
    const long gsplits = p.xcd_group ? (splits + 7) / 8 * 8 : splits;
    if (Cfine == 3) hipLaunchKernelGGL((conv_wgrad_kernel<TS, true>), dim3((unsigned)(tiles * gsplits)), dim3(kThreads), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<TS, false>), dim3((unsigned)(tiles * gsplits)), dim3(kThreads), 0, (hipStream_t)stream, p);
  };
  // 128x128 tiles halve the L2->LDS traffic per flop; they need >= 128 output channels to fill their rows
  // Measured (tools/conv_bench.py --only conv_wgrad; r02): BK 32 instead of 16 (twice the gathers in flight per
  // barrier) -4 % at cfg-2 sizes, -8 % at crafter widths; more K-splits at long reductions (a split of ~3k rows
  // instead of 512 workgroups in total): 4096 frames, depth 96: 15.0 / 10.4 / 10.3 ms -> 8.6 / 6.5 / 6.7 ms
  // (41-60 -> 72-95 TFLOP/s) -- with 1-2 workgroups per CU each marching through 12k rows of HBM-resident operands
  // the gathers' latency was exposed; cfg-2 sizes are unchanged by the rule (their splits are already ~1k rows).
  static const int env_k32 = DV3_ENV_INT("DV3_WGRAD_K32", 1);
  static const int env_wgs = DV3_ENV_INT("DV3_WGRAD_WGS", 0);
  auto target128 = [&](int bm, int bn) {
    if (env_wgs > 0) return (long)env_wgs;
    const long tiles = (long)((Ccoarse + bm - 1) / bm) * ((16 * Cfine + bn - 1) / bn);
    const long by_rows = tiles * ((rows + 3071) / 3072);
    return by_rows > 512 ? by_rows : 512L;
  };
  if (Ccoarse >= 128 && env_k32) go(C128K32{}, target128(128, 128));
  else if (Ccoarse >= 128) go(C128{}, target128(128, 128));
  else if (Ccoarse <= 32) go(C32x64S{}, 1024);  // image-side layers: 32 output channels x 48 (ky,kx,ci) columns
  else go(C64{}, 1024);
  const long total = 16L * Ccoarse * Cfine;
  unsigned ub = (unsigned)((total + 255) / 256);
  if (ub > 1024) ub = 1024;
  hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3(ub), dim3(256), 0, (hipStream_t)stream, dw_packed, dw, Ccoarse, Cfine);
  return (int)hipGetLastError();
}

// Narrow layers (Cfine 32, Ccoarse 64): tile walk with the whole gradient in one workgroup's accumulators, two-stage
// reduction through `partial` (dv3_conv_s2_wgrad_tile_scratch floats).  Returns DV3_ERR_ARG for other shapes.
extern "C" int dv3_conv_s2_wgrad_tile_scratch(int Nimg, int H, int W, int Cfine, int Ccoarse) {
  if (Cfine != 32 || Ccoarse != 64 || H <= 0 || W <= 0 || (H % 16) != 0 || (W % 16) != 0 || Nimg <= 0) return 0;
  const long tiles = (long)Nimg * (H / 16) * (W / 16);
  return (int)((tiles < 256 ? tiles : 256) * (long)Ccoarse * 16 * Cfine);
}
extern "C" int dv3_conv_s2_wgrad_tile(const float* coarse, const float* fine, float* partial, float* dw, int Nimg, int H,
                                      int W, int Cfine, int Ccoarse, void* stream) {
  if (Nimg <= 0) return 0;
  if (!coarse || !fine || !partial || !dw || dv3_conv_s2_wgrad_tile_scratch(Nimg, H, W, Cfine, Ccoarse) == 0)
    return DV3_ERR_ARG;
  const long tiles = (long)Nimg * (H / 16) * (W / 16);
  const int G = (int)(tiles < 256 ? tiles : 256);
  const size_t lds = (size_t)(64 * (64 + 16) + 18 * 18 * (32 + 8)) * sizeof(float);
  static unsigned long long seen[4] = {0, 0, 0, 0};
  if (dv3_first_on_device(seen))
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tile_kernel<32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((conv_wgrad_tile_kernel<32, 64>), dim3(G), dim3(512), lds, s, coarse, fine, partial, Nimg, H, W);
  hipLaunchKernelGGL(conv_wgrad_tile_reduce_kernel, dim3((64 * 16 * 32 + 255) / 256), dim3(256), 0, s, partial, dw, G, 64, 32);
  return (int)hipGetLastError();
}

// Specialised image-side layers (3 channels on one side).  Weights in the REFERENCE layout (no packing):
// conv: Conv2d weight [CW][3][4][4];  convT: ConvTranspose2d weight [CW][3][4][4] (in = CW, out = 3).
extern "C" int dv3_conv_s2_c3_fwd(const float* x, const float* w, float* y, int Nimg, int H, int W, int CW,
                                  int accumulate, void* stream) {
  if (Nimg <= 0) return 0;
  if (!x || !w || !y || !pow2_spatial(H, W)) return DV3_ERR_ARG;
  const long total = (long)Nimg * (H / 2) * (W / 2);
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
  static const int env_mfma = DV3_ENV_INT("DV3_C3_MFMA", 1);
  if (env_mfma && (CW == 32 || CW == 96) && total < 0x7fffffffL * 64) {
    const dim3 grid((unsigned)((total + 127) / 128));
    if (CW == 32) hipLaunchKernelGGL((conv_s2_c3_mfma_kernel<32>), grid, dim3(256), 0, s, x, w, y, Nimg, H, W, accumulate);
    else hipLaunchKernelGGL((conv_s2_c3_mfma_kernel<96>), grid, dim3(256), 0, s, x, w, y, Nimg, H, W, accumulate);
    return (int)hipGetLastError();
  }
  if (CW == 32) hipLaunchKernelGGL((conv_s2_c3_kernel<32>), dim3(blocks), dim3(256), 0, s, x, w, y, Nimg, H, W, accumulate);
  else if (CW == 96) hipLaunchKernelGGL((conv_s2_c3_kernel<96>), dim3(blocks), dim3(256), 0, s, x, w, y, Nimg, H, W, accumulate);
  else return DV3_ERR_ARG;
  return (int)hipGetLastError();
}
extern "C" int dv3_convT_s2_c3_fwd(const float* x, const float* w, const float* bias, float out_add, float* y,
                                   int Nimg, int IH, int IW, int CW, int accumulate, void* stream) {
  if (Nimg <= 0) return 0;
  if (!x || !w || !y || IH <= 0 || IW <= 0) return DV3_ERR_ARG;
  const long total = (long)Nimg * IH * IW;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
  static const int env_c3t = DV3_ENV_INT("DV3_C3T_MFMA", 1);
  if (env_c3t && (IH % 16) == 0 && (IW % 16) == 0 && (CW == 32 || CW == 96)) {
    const dim3 grid((unsigned)((long)Nimg * (IH / 16) * (IW / 16)));
    if (CW == 32) hipLaunchKernelGGL((convT_s2_c3_mfma_kernel<32>), grid, dim3(256), 0, s, x, w, bias, out_add, y, Nimg, IH, IW, accumulate);
    else hipLaunchKernelGGL((convT_s2_c3_mfma_kernel<96>), grid, dim3(256), 0, s, x, w, bias, out_add, y, Nimg, IH, IW, accumulate);
    return (int)hipGetLastError();
  }
  if (CW == 32) hipLaunchKernelGGL((convT_s2_c3_kernel<32>), dim3(blocks), dim3(256), 0, s, x, w, bias, out_add, y, Nimg, IH, IW, accumulate);
  else if (CW == 96) hipLaunchKernelGGL((convT_s2_c3_kernel<96>), dim3(blocks), dim3(256), 0, s, x, w, bias, out_add, y, Nimg, IH, IW, accumulate);
  else return DV3_ERR_ARG;
  return (int)hipGetLastError();
}
