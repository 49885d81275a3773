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
  if (CW == 32) hipLaunchKernelGGL((convT_s2_c3_kernel<32>), dim3(blocks), dim3(256), 0, s, x, w, bias, out_add, y, Nimg, IH, IW, accumulate);
  else if (CW == 96) hipLaunchKernelGGL((convT_s2_c3_kernel<96>), dim3(blocks), dim3(256), 0, s, x, w, bias, out_add, y, Nimg, IH, IW, accumulate);
  else return DV3_ERR_ARG;
  return (int)hipGetLastError();
}
