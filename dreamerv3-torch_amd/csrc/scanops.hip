// Few-row (M = replay batch) launches of the observe scan with the row operation of the PREVIOUS layer folded
// into the prologue of the few-row GEMM that consumes it (networks.RSSM.obs_step, networks.py:174-206).
//
// The scan is a chain of dependent launches on 16..64 rows; each costs a kernel boundary plus its own fill / drain
// (DESIGN.md section 4), whatever it computes.  A LayerNorm between two GEMMs is an all-to-all seam only on its INPUT
// side (the row statistics need the whole row); its output is elementwise.  So the consumer GEMM recomputes it: every
// workgroup reads the 16 complete rows (32..96 KB out of L2), its 8 waves compute the row sums of two rows each, and
// after one barrier every wave transforms exactly the A fragments it multiplies.  Workgroup column 0 also stores the
// transformed rows (saved activations / gradients the batched weight gradients read after the scan).
//
// What this costs is the redundant elementwise work: a workgroup does a whole 16 x K row block on ONE CU.  Measured
// (MI355X, cfg 2): with 3 transcendentals per element (the GRU gates: 16 x 512 x ~120 instructions = 6.4 us on four
// SIMDs) the fused launch is slower than the two it replaces -- the gates keep their own launch; with one
// (SiLU) it pays; in the reverse scan the transcendental factors depend on forward data only, so they are computed
// for all T*B rows at once in front of the scan (scan_ln_factors_kernel) and the prologue is multiply-add only.
//
// Forward arithmetic is that of the kernels replaced, element for element and in the same order (statistics: lane l
// owns float4 chunks l + 64 v as in ln_act_fwd_vec_kernel; products: the K split over 8 waves, chunk order and LDS
// reduction order of gemm_skinny_kernel): bit-identical to the two-launch form (tests/test_kernels_gpu.py).
#include "dv3_common.h"

namespace dv3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float hsum4(f32x4 a) { return (a.x + a.y) + (a.z + a.w); }

constexpr int kScanWaves = 8;

// shared epilogue: the 8 waves' 16 x 16 partial tiles -> C (same order as gemm_skinny_kernel)
__device__ __forceinline__ void scan_tile_out(const f32x4& acc, float (*red)[256], int rb, int n0, int M, int N, float* C,
                                              long ldc, const float* bias, bool accumulate) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r * 64 + lane] = acc[r];
  __syncthreads();
  if (tid < 256) {
    const int r = tid >> 6, l = tid & 63;
    const int row = rb + 4 * (l >> 4) + r, col = n0 + (l & 15);
    if (row < M && col < N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kScanWaves; ++w) v += red[w][tid];
      if (bias) v += bias[col];
      float* o = C + (long)row * ldc + col;
      if (accumulate) v += *o;
      *o = v;
    }
  }
}

struct ScanLnParams {
  const float* x;  // [M, K] pre-activations
  long ldx;
  const float* gamma;  // [K]
  const float* beta;
  float* y;  // [M, K] SiLU(LN(x)) (saved for the backward pass)
  long ldy;
  float* mean;  // [M]
  float* rstd;
  const float* W;  // [N, K] k-contiguous
  long ldw;
  const float* bias;  // [N] or null
  float* C;  // [M, N]
  long ldc;
  int M, K, N, accumulate;
};

// C[M,N] (+)= SiLU(LN(x)) W^T + bias   (ln_act_fwd_vec_kernel<64, NV> + gemm_skinny_kernel<true, 1, .>)
// grid (N / 16, 1, ceil(M / 16)), 512 threads.  K = 256 NV.
template <int NV>
__global__ __launch_bounds__(64 * kScanWaves) void scan_ln_gemm_kernel(ScanLnParams p) {
  constexpr int PER = 2 * NV;
  __shared__ float red[kScanWaves][256];
  __shared__ float sstat[2][16];
  __shared__ __attribute__((aligned(16))) float sgb[2][256 * NV];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16, rb = blockIdx.z * 16;
  const int row = rb + i;
  const bool rowok = row < p.M;
  const int ncol = n0 + i;
  const bool colok = ncol < p.N;
  const float amask = rowok ? 1.f : 0.f, bmask = colok ? 1.f : 0.f;
  const float* wrow = p.W + (long)(colok ? ncol : 0) * p.ldw;
  const float* xrow = p.x + (long)(rowok ? row : 0) * p.ldx;
  const int cb = wave * PER;
  f32x4 b[PER], xa[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int k = ((cb + u) << 4) + 4 * q;
    b[u] = *reinterpret_cast<const f32x4u*>(wrow + k);
    xa[u] = *reinterpret_cast<const f32x4u*>(xrow + k);
  }
  for (int c = tid; c < p.K / 4; c += 64 * kScanWaves) {
    *reinterpret_cast<f32x4*>(&sgb[0][4 * c]) = *reinterpret_cast<const f32x4u*>(p.gamma + 4 * c);
    *reinterpret_cast<f32x4*>(&sgb[1][4 * c]) = *reinterpret_cast<const f32x4u*>(p.beta + 4 * c);
  }
  const float inv_n = 1.f / (float)p.K;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int r = rb + 2 * wave + rr;
    if (r < p.M) {  // wave-uniform
      const float* pr = p.x + (long)r * p.ldx;
      f32x4 xv[NV];
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        xv[v] = *reinterpret_cast<const f32x4u*>(pr + 4 * (lane + 64 * v));
        s += hsum4(xv[v]);
      }
      const float mean = group_sum<64>(s) * inv_n;
      float qq = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x4 d = xv[v] - mean;
        qq += hsum4(d * d);
      }
      const float rstd = rsqrtf(group_sum<64>(qq) * inv_n + kLnEps);
      if (lane == 0) {
        sstat[0][2 * wave + rr] = mean;
        sstat[1][2 * wave + rr] = rstd;
        if (blockIdx.x == 0) {
          if (p.mean) p.mean[r] = mean;
          if (p.rstd) p.rstd[r] = rstd;
        }
      }
    }
  }
  __syncthreads();
  const float mean = sstat[0][rowok ? i : 0], rstd = sstat[1][rowok ? i : 0];
  const bool writer = blockIdx.x == 0 && rowok && p.y;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int k = ((cb + u) << 4) + 4 * q;
    f32x4 z = (xa[u] - mean) * rstd * *reinterpret_cast<const f32x4*>(&sgb[0][k]) + *reinterpret_cast<const f32x4*>(&sgb[1][k]);
    z.x = siluf_(z.x); z.y = siluf_(z.y); z.z = siluf_(z.z); z.w = siluf_(z.w);
    if (writer) *reinterpret_cast<f32x4u*>(p.y + (long)row * p.ldy + k) = z;
#pragma unroll
    for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(z[g] * amask, b[u][g] * bmask, acc, 0, 0, 0);
  }
  scan_tile_out(acc, red, rb, n0, p.M, p.N, p.C, p.ldc, p.bias, p.accumulate != 0);
}


// ------------------------------------------------------------------------------------------------------
// Reverse scan.  Data gradients are dX = dY W with W [K][N] row-major (n-contiguous), few rows, K split over
// workgroups and the partial tiles added with atomics (gemm_skinny_nn64_kernel's scheme: a wave owns 64 columns as
// four interleaved 16-column tiles, lane (i, q) reads W[k + g][n0 + 4 i .. + 3] as one float4).  The row operation
// that produces dY rides in the prologue.
// ------------------------------------------------------------------------------------------------------
struct ScanLnBwdParams {
  const float* dy;  // [M, K] gradient on SiLU(LN(x))
  long lddy;
  const float* xhat;  // [M, K] (x - mean) rstd                      } scan_ln_factors_kernel, once per update for
  const float* jac;   // [M, K] d SiLU(z) / dz at z = xhat gamma + beta } all T*B rows
  const float* gamma;
  const float* rstd;  // [M] saved by the forward pass
  float* dx;  // [M, K] gradient on x (saved: weight gradients after the scan)
  long lddx;
  float* dgamma;  // [K] accumulated (optional)
  float* dbeta;
  const float* W;  // [K, >= N]
  long ldb;
  float* C;  // [M, N]  C += dx W  (atomic)
  long ldc;
  int M, K, N;
};

// xhat = (x - mean) rstd and jac = SiLU'(xhat gamma + beta) for every row of a LayerNorm + SiLU layer: the factors of
// its backward pass that depend on forward data only (float4 per thread; K % 4 == 0).
__global__ __launch_bounds__(256) void scan_ln_factors_kernel(const float* __restrict__ x, long ldx,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ xhat,
                                                              float* __restrict__ jac, long R, int K) {
  const long total = R * (K / 4);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / (K / 4);
    const int c = 4 * (int)(e % (K / 4));
    const f32x4 xh = (*reinterpret_cast<const f32x4u*>(x + r * ldx + c) - mean[r]) * rstd[r];
    const f32x4 z = xh * *reinterpret_cast<const f32x4u*>(gamma + c) + *reinterpret_cast<const f32x4u*>(beta + c);
    f32x4 j;
    j.x = dsiluf_(z.x); j.y = dsiluf_(z.y); j.z = dsiluf_(z.z); j.w = dsiluf_(z.w);
    *reinterpret_cast<f32x4u*>(xhat + r * K + c) = xh;
    *reinterpret_cast<f32x4u*>(jac + r * K + c) = j;
  }
}

// epilogue of the n-contiguous few-row kernels: 8 waves x four interleaved 16 x 16 tiles -> atomic adds onto C
__device__ __forceinline__ void scan_tile64_out(const f32x4 (&acc)[4], float (*red)[4][257], int rb, int n0, int M, float* C,
                                                long ldc) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][j][r * 64 + lane] = acc[j][r];
  __syncthreads();
  for (int e = tid; e < 1024; e += 64 * kScanWaves) {
    const int row16 = e >> 6, cidx = e & 63;
    const int row = rb + row16;
    if (row < M) {
      const int x = (row16 & 3) * 64 + (row16 >> 2) * 16 + (cidx >> 2), j = cidx & 3;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kScanWaves; ++w) v += red[w][j][x];
      atomicAdd(C + (long)row * ldc + n0 + cidx, v);
    }
  }
}

// dx = LN/SiLU backward of dy (= ln_act_bwd_vec_kernel, act = 1) from the precomputed factors, then C += dx W.
// K = 256 NV, N % 64 == 0.  grid (N / 64, K / 128, ceil(M / 16)), 512 threads: a workgroup owns 8 16-k chunks, one
// per wave.
template <int NV>
__global__ __launch_bounds__(64 * kScanWaves) void scan_lnbwd_gemm_kernel(ScanLnBwdParams p) {
  __shared__ float red[kScanWaves][4][257];
  __shared__ float sst[2][16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 64, rb = blockIdx.z * 16;
  const int row = rb + i;
  const bool rowok = row < p.M;
  const int rr0 = rowok ? row : 0;
  const int k = ((blockIdx.y * kScanWaves + wave) << 4) + 4 * q;
  // this wave's operands, all in flight before the row-sum pass waits for anything
  f32x4 b[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const f32x4u*>(p.W + (long)(k + g) * p.ldb + n0 + 4 * i);
  const f32x4 dyv = *reinterpret_cast<const f32x4u*>(p.dy + (long)rr0 * p.lddy + k);
  const f32x4 xhat = *reinterpret_cast<const f32x4u*>(p.xhat + (long)rr0 * p.K + k);
  const f32x4 jv = *reinterpret_cast<const f32x4u*>(p.jac + (long)rr0 * p.K + k);
  const f32x4 gv = *reinterpret_cast<const f32x4u*>(p.gamma + k);
  const float rstd = p.rstd[rr0];
  // row sums of rows 2 wave, 2 wave + 1 (ln_act_bwd_vec_kernel<64, NV>'s order)
  const float inv_n = 1.f / (float)p.K;
  {
    f32x4 dy2[2][NV], xh2[2][NV], j2[2][NV], g4[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) g4[v] = *reinterpret_cast<const f32x4u*>(p.gamma + 4 * (lane + 64 * v));
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = min(rb + 2 * wave + rr, p.M - 1);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = 4 * (lane + 64 * v);
        dy2[rr][v] = *reinterpret_cast<const f32x4u*>(p.dy + (long)r * p.lddy + c);
        xh2[rr][v] = *reinterpret_cast<const f32x4u*>(p.xhat + (long)r * p.K + c);
        j2[rr][v] = *reinterpret_cast<const f32x4u*>(p.jac + (long)r * p.K + c);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x4 dxh = dy2[rr][v] * j2[rr][v] * g4[v];
        s1 += hsum4(dxh);
        s2 += hsum4(dxh * xh2[rr][v]);
      }
      s1 = group_sum<64>(s1) * inv_n;
      s2 = group_sum<64>(s2) * inv_n;
      if (lane == 0) {
        sst[0][2 * wave + rr] = s1;
        sst[1][2 * wave + rr] = s2;
      }
    }
  }
  __syncthreads();
  const float s1 = sst[0][rowok ? i : 0], s2 = sst[1][rowok ? i : 0];
  const f32x4 dz = rowok ? dyv * jv : (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4 dxh = dz * gv;
  f32x4 d = (dxh - s1 - xhat * s2) * rstd;
  if (!rowok) d = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0) {
    if (rowok) *reinterpret_cast<f32x4u*>(p.dx + (long)row * p.lddx + k) = d;
    if (p.dgamma) {  // column sums over the 16 rows of the block: lanes i = 0..15 of each q group
      f32x4 pg = dz * xhat, pb = dz;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pg[e] += __shfl_xor(pg[e], o, 64);
          pb[e] += __shfl_xor(pb[e], o, 64);
        }
      if (i == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          atomicAdd(p.dgamma + k + e, pg[e]);
          atomicAdd(p.dbeta + k + e, pb[e]);
        }
      }
    }
  }
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[g], b[g][j], acc[j], 0, 0, 0);
  scan_tile64_out(acc, red, rb, n0, p.M, p.C, p.ldc);
}

// GRU cell backward in the prologue of the data-gradient GEMM of its Linear (networks.py:760-768 reversed;
// = gru_bwd_vec_kernel + gemm_skinny_nn64_kernel).  Everything transcendental in the cell's backward depends on forward
// data only, so it is computed for all T*B rows at once in front of the reverse scan (scan_gru_factors_kernel):
//   dy_k = g_j A_k  for the three gate parts k = (r, c, u) x j,   dh_direct = g_j Ah_j,
//   the LayerNorm row sums collapse to two dot products over the hidden units: s1 = g . P1 / N, s2 = g . P2 / N
// (P1_j = sum_parts A gamma, P2_j = sum_parts A gamma xhat).  The per-step prologue reads g, P1, P2 of the 16 rows
// (3 De floats per row instead of the 3 De pre-activations + De state + their 3 transcendentals per unit) and the
// factors of its own K slice.
struct ScanGruBwdParams {
  const float* g;  // [M, De] gradient on the new deter
  long ldg;
  const float* xh;  // [M, 3 De] xhat of the LayerNorm           }
  const float* af;  // [M, 3 De] A_k                              } scan_gru_factors_kernel
  const float* p1;  // [M, De]                                    }
  const float* p2;
  const float* ah;     // [M, De] 1 - update gate
  const float* gamma;  // [3 De]
  const float* rstd;   // [M]
  float* dp;           // [M, 3 De] gradient on the GEMM output of the forward pass (saved: weight gradient after the scan)
  float* dh;           // [M, De] (ld) += g Ah (atomic: the GEMM below accumulates onto the same buffer)
  long lddh;
  float* dgamma;  // [3 De] accumulated (optional)
  float* dbeta;
  const float* W;  // [3 De, >= N]
  long ldb;
  float* C;  // [M, N]  C += dp W (atomic)
  long ldc;
  int M, De, N;
};

__global__ __launch_bounds__(256) void scan_gru_factors_kernel(const float* __restrict__ p, long ldp,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ h, long ldh,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ xh,
                                                               float* __restrict__ af, float* __restrict__ p1,
                                                               float* __restrict__ p2, float* __restrict__ ah, long R,
                                                               int De) {
  const long total = R * De;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / De;
    const int j = (int)(e % De);
    const float mu = mean[r], rs = rstd[r];
    const float* pr = p + r * ldp;
    const float xr = (pr[j] - mu) * rs, xc = (pr[De + j] - mu) * rs, xu = (pr[2 * De + j] - mu) * rs;
    const float gr = gamma[j], gc = gamma[De + j], gu = gamma[2 * De + j];
    const float yr = xr * gr + beta[j], yc = xc * gc + beta[De + j], yu = xu * gu + beta[2 * De + j];
    const float rg = sigmoidf_(yr);
    const float cg = tanhf(rg * yc);
    const float ug = sigmoidf_(yu - 1.f);
    const float drc = ug * (1.f - cg * cg);
    const float a_r = drc * yc * rg * (1.f - rg), a_c = drc * rg, a_u = (cg - h[r * ldh + j]) * ug * (1.f - ug);
    float* xo = xh + r * 3 * De;
    float* ao = af + r * 3 * De;
    xo[j] = xr; xo[De + j] = xc; xo[2 * De + j] = xu;
    ao[j] = a_r; ao[De + j] = a_c; ao[2 * De + j] = a_u;
    p1[e] = a_r * gr + a_c * gc + a_u * gu;
    p2[e] = a_r * gr * xr + a_c * gc * xc + a_u * gu * xu;
    ah[e] = 1.f - ug;
  }
}

// grid (N / 64, 3 De / (16 * 8 * CH), ceil(M / 16)), 512 threads: a workgroup owns 8 CH 16-k chunks, CH per wave.
template <int NVD, int CH>  // De = 256 NVD
__global__ __launch_bounds__(64 * kScanWaves) void scan_grubwd_gemm_kernel(ScanGruBwdParams p) {
  __shared__ float red[kScanWaves][4][257];
  __shared__ float sst[2][16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 64, rb = blockIdx.z * 16;
  const int De = p.De;
  const int row = rb + i;
  const bool rowok = row < p.M;
  const int rr0 = rowok ? row : 0;
  const float rstd = p.rstd[rr0];
  // this wave's chunks: everything in flight before the row-sum pass waits
  f32x4 b[CH][4], gq[CH], afq[CH], xhq[CH], gam[CH], ahq[CH];
  int kk[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int k = (((blockIdx.y * kScanWaves + wave) * CH + u) << 4) + 4 * q;
    kk[u] = k;
    const int part = k / De, j = k - part * De;
#pragma unroll
    for (int g = 0; g < 4; ++g) b[u][g] = *reinterpret_cast<const f32x4u*>(p.W + (long)(k + g) * p.ldb + n0 + 4 * i);
    gq[u] = *reinterpret_cast<const f32x4u*>(p.g + (long)rr0 * p.ldg + j);
    afq[u] = *reinterpret_cast<const f32x4u*>(p.af + (long)rr0 * 3 * De + k);
    xhq[u] = *reinterpret_cast<const f32x4u*>(p.xh + (long)rr0 * 3 * De + k);
    gam[u] = *reinterpret_cast<const f32x4u*>(p.gamma + k);
    ahq[u] = *reinterpret_cast<const f32x4u*>(p.ah + (long)rr0 * De + j);
  }
  // row sums of rows 2 wave, 2 wave + 1: two dot products over the hidden units
  const float inv_n = 1.f / (float)(3 * De);
  {
    f32x4 g2[2][NVD], a2[2][NVD], c2[2][NVD];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = min(rb + 2 * wave + rr, p.M - 1);
#pragma unroll
      for (int v = 0; v < NVD; ++v) {
        const int c = 4 * (lane + 64 * v);
        g2[rr][v] = *reinterpret_cast<const f32x4u*>(p.g + (long)r * p.ldg + c);
        a2[rr][v] = *reinterpret_cast<const f32x4u*>(p.p1 + (long)r * De + c);
        c2[rr][v] = *reinterpret_cast<const f32x4u*>(p.p2 + (long)r * De + c);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int v = 0; v < NVD; ++v) {
        s1 += hsum4(g2[rr][v] * a2[rr][v]);
        s2 += hsum4(g2[rr][v] * c2[rr][v]);
      }
      s1 = group_sum<64>(s1) * inv_n;
      s2 = group_sum<64>(s2) * inv_n;
      if (lane == 0) {
        sst[0][2 * wave + rr] = s1;
        sst[1][2 * wave + rr] = s2;
      }
    }
  }
  __syncthreads();
  const float s1 = sst[0][rowok ? i : 0], s2 = sst[1][rowok ? i : 0];
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int k = kk[u];
    const f32x4 dz = rowok ? gq[u] * afq[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 dxh = dz * gam[u];
    f32x4 d = (dxh - s1 - xhq[u] * s2) * rstd;
    if (!rowok) d = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (blockIdx.x == 0) {
      if (rowok) {
        *reinterpret_cast<f32x4u*>(p.dp + (long)row * 3 * De + k) = d;
        if (k < De) {  // the direct path of the state: dh += g (1 - u), once per hidden unit (the r-gate chunks)
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(p.dh + (long)row * p.lddh + k + e, gq[u][e] * ahq[u][e]);
        }
      }
      if (p.dgamma) {  // column sums over the 16 rows of the block
        f32x4 pg = dz * xhq[u], pb = dz;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pg[e] += __shfl_xor(pg[e], o, 64);
            pb[e] += __shfl_xor(pb[e], o, 64);
          }
        if (i == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            atomicAdd(p.dgamma + k + e, pg[e]);
            atomicAdd(p.dbeta + k + e, pb[e]);
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[g], b[u][g][j], acc[j], 0, 0, 0);
  }
  scan_tile64_out(acc, red, rb, n0, p.M, p.C, p.ldc);
}

struct ScanCarryParams {
  const float* dsin;  // [B, SD] gradient on the NEXT step's blended stoch input (null: no carry, last step)
  long ld_dsin;
  const float* ddin;  // [B, De] ... and on its blended deter input
  long ld_ddin;
  const float* first;  // [B] is_first of the next step
  const float* gs;  // [B, SD] gradient on this step's posterior sample from the heads (the carry is added on the fly)
  float* gd;        // [B, De] in: heads; out: + carry
  float* ds0;  // [SD], [De] gradients of the initial state (atomic)
  float* dd0;
  const float* logit;   // [B, S, 32] posterior logits of this step
  const float* dlogit;  // [B, S, 32] KL gradient on them
  float* dlogit_out;    // [B, S, 32] dlogit + straight-through term = the GEMM's A operand (must not alias dlogit: every
                        // column-tile workgroup re-reads the inputs)
  const float* W;      // [SD, >= N]
  long ldb;
  float* C;  // [B, N]  C += dlogit W (atomic)
  long ldc;
  int B, S, De, N;
  float unimix;
  int tiles;  // blockIdx.x >= tiles: deter-carry role
};

// The carry of the reverse scan into this step (obs_blend_bwd of the next step), the straight-through gradient of
// this step's posterior sample (OneHotDist, tools.py:452-460) and C += dlogit W in one launch
// (= obs_carry_st_bwd_kernel<32> + gemm_skinny_nn64_kernel).  D = 32 classes: a categorical group is two 16-k chunks;
// the lanes (i, q = 0..3) of row i hold its 32 classes, 8 each.  grid (N / 64 + deter blocks, S / 8, ceil(B / 16)).
__global__ __launch_bounds__(64 * kScanWaves) void scan_carry_st_gemm_kernel(ScanCarryParams p) {
  __shared__ float red[kScanWaves][4][257];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int SD = p.S * 32;
  if ((int)blockIdx.x >= p.tiles) {  // deter carry: gd += ddin (1 - m), dd0 += sum_b ddin m
    if (!p.dsin || blockIdx.y != 0 || blockIdx.z != 0) return;
    const long total = (long)p.B * p.De;
    const long nb = gridDim.x - p.tiles;
    for (long e = (long)(blockIdx.x - p.tiles) * (64 * kScanWaves) + tid; e < total; e += nb * 64 * kScanWaves) {
      const int bb = (int)(e / p.De), kk = (int)(e % p.De);
      const float m = p.first[bb];
      const float g = p.ddin[(long)bb * p.ld_ddin + kk];
      p.gd[e] += g * (1.f - m);
      if (m != 0.f) atomicAdd(p.dd0 + kk, g * m);
    }
    return;
  }
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 64, rb = blockIdx.z * 16;
  const int row = rb + i;
  const bool rowok = row < p.B;
  const int rr0 = rowok ? row : 0;
  const int grp = blockIdx.y * kScanWaves + wave;  // categorical group of this wave
  const int k0 = grp * 32 + 4 * q;                 // + 16 cc + e
  f32x4 b[2][4];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      b[cc][g] = *reinterpret_cast<const f32x4u*>(p.W + (long)(k0 + 16 * cc + g) * p.ldb + n0 + 4 * i);
  f32x4 lg[2], t[2], dl[2], ds[2];
  const float m = p.dsin ? p.first[rr0] : 0.f;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const long off = (long)rr0 * SD + k0 + 16 * cc;
    lg[cc] = *reinterpret_cast<const f32x4u*>(p.logit + off);
    t[cc] = *reinterpret_cast<const f32x4u*>(p.gs + off);
    dl[cc] = *reinterpret_cast<const f32x4u*>(p.dlogit + off);
    ds[cc] = p.dsin ? *reinterpret_cast<const f32x4u*>(p.dsin + (long)rr0 * p.ld_dsin + k0 + 16 * cc)
                    : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool writer = blockIdx.x == 0 && rowok;
  if (p.dsin) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      t[cc] += ds[cc] * (1.f - m);
      if (writer && m != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(p.ds0 + k0 + 16 * cc + e, ds[cc][e] * m);
      }
    }
  }
  // softmax over the group's 32 classes: 8 in this lane, the rest in the lanes q ^ 1, q ^ 2 of the same row
  float mx = fmaxf(fmaxf(fmaxf(lg[0].x, lg[0].y), fmaxf(lg[0].z, lg[0].w)), fmaxf(fmaxf(lg[1].x, lg[1].y), fmaxf(lg[1].z, lg[1].w)));
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  f32x4 ex[2];
  float se = 0.f;
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) ex[cc][e] = expf(lg[cc][e] - mx);
    se += hsum4(ex[cc]);
  }
  se += __shfl_xor(se, 16, 64);
  se += __shfl_xor(se, 32, 64);
  float dot = 0.f;
  f32x4 sm[2];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    sm[cc] = ex[cc] / se;
    dot += hsum4(sm[cc] * t[cc]);
  }
  dot += __shfl_xor(dot, 16, 64);
  dot += __shfl_xor(dot, 32, 64);
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    f32x4 a = dl[cc] + (1.f - p.unimix) * sm[cc] * (t[cc] - dot);
    if (writer) *reinterpret_cast<f32x4u*>(p.dlogit_out + (long)row * SD + k0 + 16 * cc) = a;
    if (!rowok) a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], b[cc][g][j], acc[j], 0, 0, 0);
  }
  scan_tile64_out(acc, red, rb, n0, p.B, p.C, p.ldc);
}

}  // namespace dv3

using namespace dv3;

extern "C" int dv3_scan_ln_gemm_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy,
                                    float* mean, float* rstd, const float* W, long ldw, const float* bias, float* C,
                                    long ldc, int M, int K, int N, int accumulate, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (!x || !gamma || !beta || !W || !C) return DV3_ERR_ARG;
  if (K <= 0 || K % 256 != 0 || K > 1024 || N % 16 != 0) return DV3_ERR_ARG;
  if (ldx < K || ldw < K || ldc < N || (y && ldy < K)) return DV3_ERR_ARG;
  ScanLnParams p{x, ldx, gamma, beta, y, ldy, mean, rstd, W, ldw, bias, C, ldc, M, K, N, accumulate};
  const dim3 grid(N / 16, 1, (M + 15) / 16), block(64 * kScanWaves);
  hipStream_t s = (hipStream_t)stream;
  switch (K / 256) {
    case 1: hipLaunchKernelGGL((scan_ln_gemm_kernel<1>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((scan_ln_gemm_kernel<2>), grid, block, 0, s, p); break;
    case 4: hipLaunchKernelGGL((scan_ln_gemm_kernel<4>), grid, block, 0, s, p); break;
    default: return DV3_ERR_ARG;
  }
  return (int)hipGetLastError();
}

extern "C" int dv3_scan_ln_factors(const float* x, long ldx, const float* gamma, const float* beta, const float* mean,
                                   const float* rstd, float* xhat, float* jac, long R, int K, void* stream) {
  if (R <= 0) return 0;
  if (!x || !gamma || !beta || !mean || !rstd || !xhat || !jac || K <= 0 || K % 4 != 0 || ldx < K) return DV3_ERR_ARG;
  long blocks = (R * (K / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scan_ln_factors_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                     mean, rstd, xhat, jac, R, K);
  return (int)hipGetLastError();
}

extern "C" int dv3_scan_lnbwd_gemm(const float* dy, long lddy, const float* xhat, const float* jac, const float* gamma,
                                   const float* rstd, float* dx, long lddx, float* dgamma, float* dbeta, const float* W,
                                   long ldb, float* C, long ldc, int M, int K, int N, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (!dy || !xhat || !jac || !gamma || !rstd || !dx || !W || !C || ((dgamma == nullptr) != (dbeta == nullptr)))
    return DV3_ERR_ARG;
  if (K <= 0 || K % 256 != 0 || K > 1024 || N % 64 != 0) return DV3_ERR_ARG;
  if (lddy < K || lddx < K || ldb < N || ldc < N) return DV3_ERR_ARG;
  ScanLnBwdParams p{dy, lddy, xhat, jac, gamma, rstd, dx, lddx, dgamma, dbeta, W, ldb, C, ldc, M, K, N};
  const dim3 grid(N / 64, K / (16 * kScanWaves), (M + 15) / 16), block(64 * kScanWaves);
  hipStream_t s = (hipStream_t)stream;
  switch (K / 256) {
    case 1: hipLaunchKernelGGL((scan_lnbwd_gemm_kernel<1>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((scan_lnbwd_gemm_kernel<2>), grid, block, 0, s, p); break;
    case 4: hipLaunchKernelGGL((scan_lnbwd_gemm_kernel<4>), grid, block, 0, s, p); break;
    default: return DV3_ERR_ARG;
  }
  return (int)hipGetLastError();
}

extern "C" int dv3_scan_carry_st_gemm(const float* dsin, long ld_dsin, const float* ddin, long ld_ddin,
                                      const float* is_first, const float* gs, float* gd, float* dstoch0,
                                      float* ddeter0, const float* logit, const float* dlogit, float* dlogit_out,
                                      const float* W, long ldb, float* C, long ldc,
                                      int B, int S, int D, int De, int N, float unimix, void* stream) {
  if (B <= 0 || N <= 0) return 0;
  if (!gs || !logit || !dlogit || !dlogit_out || dlogit_out == dlogit || !W || !C) return DV3_ERR_ARG;
  if (dsin && (!ddin || !is_first || !gd || !dstoch0 || !ddeter0 || De <= 0 || ld_dsin < (long)S * D || ld_ddin < De))
    return DV3_ERR_ARG;
  if (D != 32 || S <= 0 || S % kScanWaves != 0 || N % 64 != 0 || ldb < N || ldc < N) return DV3_ERR_ARG;
  const int tiles = N / 64;
  long nb_d = dsin ? ((long)B * De + 64 * kScanWaves - 1) / (64 * kScanWaves) : 0;
  if (nb_d > 64) nb_d = 64;
  ScanCarryParams p{dsin, ld_dsin, ddin, ld_ddin, is_first, gs, gd, dstoch0, ddeter0, logit, dlogit, dlogit_out, W, ldb,
                    C, ldc, B, S, De, N, unimix, tiles};
  const dim3 grid(tiles + (int)nb_d, S / kScanWaves, (B + 15) / 16), block(64 * kScanWaves);
  hipLaunchKernelGGL(scan_carry_st_gemm_kernel, grid, block, 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

extern "C" int dv3_scan_gru_factors(const float* p, long ldp, const float* gamma, const float* beta, const float* h,
                                    long ldh, const float* mean, const float* rstd, float* xhat, float* afac, float* p1,
                                    float* p2, float* ah, long R, int De, void* stream) {
  if (R <= 0) return 0;
  if (!p || !gamma || !beta || !h || !mean || !rstd || !xhat || !afac || !p1 || !p2 || !ah || De <= 0 || ldp < 3 * De ||
      ldh < De)
    return DV3_ERR_ARG;
  long blocks = (R * De + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scan_gru_factors_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, ldp, gamma, beta,
                     h, ldh, mean, rstd, xhat, afac, p1, p2, ah, R, De);
  return (int)hipGetLastError();
}

extern "C" int dv3_scan_grubwd_gemm(const float* g, long ldg, const float* xhat, const float* afac, const float* p1,
                                    const float* p2, const float* ah, const float* gamma, const float* rstd, float* dp,
                                    float* dh, long lddh, float* dgamma, float* dbeta, const float* W, long ldb, float* C,
                                    long ldc, int M, int De, int N, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (!g || !xhat || !afac || !p1 || !p2 || !ah || !gamma || !rstd || !dp || !dh || !W || !C ||
      ((dgamma == nullptr) != (dbeta == nullptr)))
    return DV3_ERR_ARG;
  if ((De != 256 && De != 512 && De != 1024) || N % 64 != 0 || ldg < De || lddh < De || ldb < N || ldc < N)
    return DV3_ERR_ARG;
  ScanGruBwdParams p{g, ldg, xhat, afac, p1, p2, ah, gamma, rstd, dp, dh, lddh, dgamma, dbeta, W, ldb, C, ldc, M, De, N};
  hipStream_t s = (hipStream_t)stream;
  const int chunks = 3 * De / 16;
  // one 16-k chunk per wave at De 256 / 512 (6 / 12 K-splits), two at De 1024 (12): 96 .. 288 workgroups
  if (De == 1024) {
    const dim3 grid(N / 64, chunks / (kScanWaves * 2), (M + 15) / 16);
    hipLaunchKernelGGL((scan_grubwd_gemm_kernel<4, 2>), grid, dim3(64 * kScanWaves), 0, s, p);
  } else {
    const dim3 grid(N / 64, chunks / kScanWaves, (M + 15) / 16);
    if (De == 512) hipLaunchKernelGGL((scan_grubwd_gemm_kernel<2, 1>), grid, dim3(64 * kScanWaves), 0, s, p);
    else hipLaunchKernelGGL((scan_grubwd_gemm_kernel<1, 1>), grid, dim3(64 * kScanWaves), 0, s, p);
  }
  return (int)hipGetLastError();
}
