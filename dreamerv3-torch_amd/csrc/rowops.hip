// Row-wise kernels of the hot path: LayerNorm(eps 1e-3)+SiLU and the LayerNorm-GRU gate math,
// forward and backward.  HBM-bound: one pass over the row, statistics in registers, wave (or
// sub-wave group) per row, parameter gradients accumulated in registers across a grid-stride
// loop and flushed with one atomicAdd per column per wave.
//
// Reference sites: nn.LayerNorm + SiLU after every bias-free Linear (networks.py:48-58, 62-78,
// 624-636), ImgChLayerNorm + SiLU after every conv (networks.py:475-477, 551-554, 801-810),
// GRUCell.forward (networks.py:760-768).
#include "dv3_common.h"

namespace dv3 {

constexpr int kMaxV = 32;  // per-lane cached elements (template NV <= kMaxV); rows up to 64*32 = 2048

// address of (row r, col c) for an activation tensor that is either plain [R][ld] or, when
// G > 0, the reference's (C,H,W) flatten of an NHWC tile: rows are (image, pixel) pairs with G
// pixels per image and the element lives at img*(N*G) + c*G + pix  (networks.py:494).
__device__ __forceinline__ long act_addr(long r, int c, long ld, int N, int G) {
  if (G <= 0) return r * ld + c;
  const long img = r / G;
  const int pix = (int)(r - img * G);
  return img * ((long)N * G) + (long)c * G + pix;
}

template <int LPR, int NV>
__global__ __launch_bounds__(256) void ln_act_fwd_kernel(const float* __restrict__ x, long ldx,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y,
                                                         long ldy, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, long R, int N, int act,
                                                         int G) {
  constexpr int RPB = 256 / LPR;
  const int sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  float g[NV], b[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = l + v * LPR;
    const bool ok = c < N;
    g[v] = ok ? gamma[c] : 0.f;
    b[v] = ok ? beta[c] : 0.f;
  }
  const float inv_n = 1.f / (float)N;
  for (long r = (long)blockIdx.x * RPB + sub; r < R; r += (long)gridDim.x * RPB) {
    float xv[NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = l + v * LPR;
      xv[v] = (c < N) ? x[r * ldx + c] : 0.f;
      s += xv[v];
    }
    const float mean = group_sum<LPR>(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = l + v * LPR;
      const float d = (c < N) ? xv[v] - mean : 0.f;
      q += d * d;
    }
    const float rstd = rsqrtf(group_sum<LPR>(q) * inv_n + kLnEps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = l + v * LPR;
      if (c < N) {
        float z = (xv[v] - mean) * rstd * g[v] + b[v];
        if (act) z = siluf_(z);
        y[act_addr(r, c, ldy, N, G)] = z;
      }
    }
    if (l == 0) {
      if (mean_out) mean_out[r] = mean;
      if (rstd_out) rstd_out[r] = rstd;
    }
  }
}

// dx = d(loss)/d(x) given dy = d(loss)/d(act(LN(x))).  dgamma/dbeta are ACCUMULATED (+=).
template <int LPR, int NV>
__global__ __launch_bounds__(256) void ln_act_bwd_kernel(const float* __restrict__ dy, long lddy,
                                                         const float* __restrict__ x, long ldx,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ mean_in,
                                                         const float* __restrict__ rstd_in, float* __restrict__ dx,
                                                         long lddx, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, long R, int N, int act, int G,
                                                         int accumulate_dx) {
  constexpr int RPB = 256 / LPR;
  const int sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  __shared__ float red[2][64 * kMaxV];
  for (int c = threadIdx.x; c < N; c += 256) {
    red[0][c] = 0.f;
    red[1][c] = 0.f;
  }
  __syncthreads();
  float g[NV], b[NV], pg[NV], pb[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = l + v * LPR;
    const bool ok = c < N;
    g[v] = ok ? gamma[c] : 0.f;
    b[v] = ok ? beta[c] : 0.f;
    pg[v] = 0.f;
    pb[v] = 0.f;
  }
  const float inv_n = 1.f / (float)N;
  for (long r = (long)blockIdx.x * RPB + sub; r < R; r += (long)gridDim.x * RPB) {
    const float mean = mean_in[r], rstd = rstd_in[r];
    float xh[NV], dxh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = l + v * LPR;
      xh[v] = 0.f;
      dxh[v] = 0.f;
      if (c < N) {
        const float xhat = (x[r * ldx + c] - mean) * rstd;
        float dz = dy[act_addr(r, c, lddy, N, G)];
        if (act) dz *= dsiluf_(xhat * g[v] + b[v]);
        pg[v] += dz * xhat;
        pb[v] += dz;
        xh[v] = xhat;
        dxh[v] = dz * g[v];
        s1 += dxh[v];
        s2 += dxh[v] * xhat;
      }
    }
    s1 = group_sum<LPR>(s1) * inv_n;
    s2 = group_sum<LPR>(s2) * inv_n;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = l + v * LPR;
      if (c < N) {
        const float d = rstd * (dxh[v] - s1 - xh[v] * s2);
        float* o = dx + r * lddx + c;
        *o = accumulate_dx ? (*o + d) : d;
      }
    }
  }
  if (dgamma) {
    // block-level reduce in LDS, then one global atomic per column per block
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = l + v * LPR;
      if (c < N) {
        atomicAdd(&red[0][c], pg[v]);
        atomicAdd(&red[1][c], pb[v]);
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 256) {
      atomicAdd(dgamma + c, red[0][c]);
      atomicAdd(dbeta + c, red[1][c]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm-GRU gates (networks.py:760-768).  One wave per row.
//   y = LN(p) over all 3*De;  r = sigmoid(y[0:De]); c = tanh(r * y[De:2De]); u = sigmoid(y[2De:] - 1)
//   h' = u*c + (1-u)*h
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_fwd_kernel(const float* __restrict__ p, long ldp,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ h,
                                                      long ldh, float* __restrict__ hn, long ldhn,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      int M, int De) {
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int N = 3 * De;
  const float inv_n = 1.f / (float)N;
  for (int r = blockIdx.x * 4 + wave; r < M; r += gridDim.x * 4) {
    const float* pr = p + (long)r * ldp;
    float s = 0.f;
    for (int c = l; c < N; c += 64) s += pr[c];
    const float mean = group_sum<64>(s) * inv_n;
    float q = 0.f;
    for (int c = l; c < N; c += 64) {
      const float d = pr[c] - mean;
      q += d * d;
    }
    const float rstd = rsqrtf(group_sum<64>(q) * inv_n + kLnEps);
    for (int j = l; j < De; j += 64) {
      const float yr = (pr[j] - mean) * rstd * gamma[j] + beta[j];
      const float yc = (pr[De + j] - mean) * rstd * gamma[De + j] + beta[De + j];
      const float yu = (pr[2 * De + j] - mean) * rstd * gamma[2 * De + j] + beta[2 * De + j];
      const float rg = sigmoidf_(yr);
      const float cg = tanhf(rg * yc);
      const float ug = sigmoidf_(yu - 1.f);
      const float hp = h[(long)r * ldh + j];
      hn[(long)r * ldhn + j] = ug * cg + (1.f - ug) * hp;
    }
    if (l == 0) {
      mean_out[r] = mean;
      rstd_out[r] = rstd;
    }
  }
}

// Backward: dp (w.r.t. the GEMM output p), dh (direct (1-u) path; OVERWRITTEN or accumulated), and
// accumulated dgamma/dbeta.  dy of the LN output is staged in LDS (3*De floats per wave).
__global__ __launch_bounds__(256) void gru_bwd_kernel(const float* __restrict__ dhn, long lddhn,
                                                      const float* __restrict__ p, long ldp,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ h,
                                                      long ldh, const float* __restrict__ mean_in,
                                                      const float* __restrict__ rstd_in, float* __restrict__ dp,
                                                      long lddp, float* __restrict__ dh, long lddh,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                      int De, int accumulate_dh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 4 waves x 3*De (dy) + 2 x 3*De (dgamma,dbeta)
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int N = 3 * De;
  float* dyw = smem + (long)wave * N;
  float* accg = smem + 4L * N;
  float* accb = accg + N;
  for (int c = threadIdx.x; c < 2 * N; c += 256) accg[c] = 0.f;
  __syncthreads();
  const float inv_n = 1.f / (float)N;
  for (int r = blockIdx.x * 4 + wave; r < M; r += gridDim.x * 4) {
    const float* pr = p + (long)r * ldp;
    const float mean = mean_in[r], rstd = rstd_in[r];
    float s1 = 0.f, s2 = 0.f;
    for (int j = l; j < De; j += 64) {
      const float xr = (pr[j] - mean) * rstd, xc = (pr[De + j] - mean) * rstd, xu = (pr[2 * De + j] - mean) * rstd;
      const float gr = gamma[j], gc = gamma[De + j], gu = gamma[2 * De + j];
      const float yr = xr * gr + beta[j], yc = xc * gc + beta[De + j], yu = xu * gu + beta[2 * De + j];
      const float rg = sigmoidf_(yr);
      const float cg = tanhf(rg * yc);
      const float ug = sigmoidf_(yu - 1.f);
      const float hp = h[(long)r * ldh + j];
      const float g = dhn[(long)r * lddhn + j];
      const float du = g * (cg - hp) * ug * (1.f - ug);
      const float drc = g * ug * (1.f - cg * cg);
      const float dr = drc * yc * rg * (1.f - rg);
      const float dc = drc * rg;
      float* o = dh + (long)r * lddh + j;
      const float dhd = g * (1.f - ug);
      *o = accumulate_dh ? (*o + dhd) : dhd;
      dyw[j] = dr;
      dyw[De + j] = dc;
      dyw[2 * De + j] = du;
      s1 += dr * gr + dc * gc + du * gu;
      s2 += dr * gr * xr + dc * gc * xc + du * gu * xu;
    }
    s1 = group_sum<64>(s1) * inv_n;
    s2 = group_sum<64>(s2) * inv_n;
    // same lane wrote and reads dyw[c] for c == l (mod 64) only when De % 64 == 0; otherwise sync the wave
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (int c = l; c < N; c += 64) {
      const float xh = (pr[c] - mean) * rstd;
      const float dy = dyw[c];
      dp[(long)r * lddp + c] = rstd * (dy * gamma[c] - s1 - xh * s2);
      if (dgamma) {
        atomicAdd(accg + c, dy * xh);
        atomicAdd(accb + c, dy);
      }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
  }
  if (dgamma) {
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 256) {
      atomicAdd(dgamma + c, accg[c]);
      atomicAdd(dbeta + c, accb[c]);
    }
  }
}


// ================================================================================================
// Vectorised forms (16 bytes per lane per access).  Used whenever N % 4 == 0 and the tensor is plain
// row-major; the scalar kernels above remain for ragged widths and the (C,H,W)-flatten addressing.
// A row is owned by LPR lanes; lane l holds float4 chunks l + LPR*v, v < NV (columns 4*(l+LPR*v)..+3).
// ================================================================================================
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float hsum(f4 a) { return (a.x + a.y) + (a.z + a.w); }

template <int LPR, int NV>
__global__ __launch_bounds__(256) void ln_act_fwd_vec_kernel(const float* __restrict__ x, long ldx,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             long ldy, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out, long R, int N, int act) {
  constexpr int RPB = 256 / LPR;
  const int sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  f4 g[NV], b[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = 4 * (l + LPR * v);
    const bool ok = c < N;
    g[v] = ok ? *reinterpret_cast<const f4u*>(gamma + c) : (f4){0.f, 0.f, 0.f, 0.f};
    b[v] = ok ? *reinterpret_cast<const f4u*>(beta + c) : (f4){0.f, 0.f, 0.f, 0.f};
  }
  const float inv_n = 1.f / (float)N;
  for (long r = (long)blockIdx.x * RPB + sub; r < R; r += (long)gridDim.x * RPB) {
    f4 xv[NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = 4 * (l + LPR * v);
      xv[v] = (c < N) ? *reinterpret_cast<const f4u*>(x + r * ldx + c) : (f4){0.f, 0.f, 0.f, 0.f};
      s += hsum(xv[v]);
    }
    const float mean = group_sum<LPR>(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (4 * (l + LPR * v) < N) {
        const f4 d = xv[v] - mean;
        q += hsum(d * d);
      }
    }
    const float rstd = rsqrtf(group_sum<LPR>(q) * inv_n + kLnEps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = 4 * (l + LPR * v);
      if (c < N) {
        f4 z = (xv[v] - mean) * rstd * g[v] + b[v];
        if (act) {
          z.x = siluf_(z.x); z.y = siluf_(z.y); z.z = siluf_(z.z); z.w = siluf_(z.w);
        }
        *reinterpret_cast<f4u*>(y + r * ldy + c) = z;
      }
    }
    if (l == 0) {
      if (mean_out) mean_out[r] = mean;
      if (rstd_out) rstd_out[r] = rstd;
    }
  }
}

constexpr int kColRoleRows = 128;  // rows per column-role workgroup (32 per wave)
constexpr int kColBatch = 8;       // rows a wave loads together (branch-free, clamped) before using them

// ------------------------------------------------------------------------------------------------
// Column role for d-gamma / d-beta (the observe scan issues these kernels with 16 rows once per step, on the
// critical path; the heads with 1k-15k rows).
// A CU can retire roughly one 256-byte atomic instruction per 50 ns, so flushing d-gamma / d-beta from the
// 1..4 row-role workgroups that few rows give costs 10-17 us -- several times the kernel itself.  Instead the
// same launch carries extra workgroups that OWN 64 columns each: lane = column, the 4 waves split the rows,
// the partial column sums meet in LDS and one plain read-modify-write per column finishes the job.  No
// atomics, no second launch, and the two roles run concurrently on different CUs (both only read inputs).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ln_bwd_cols(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                            float* __restrict__ dgamma, float* __restrict__ dbeta, long R, int N, int act,
                                            int cblk, float* lds) {
  // cblk = slice * ncb + cb: with more than kColRoleRows rows the rows are cut into slices of that many, one
  // set of column workgroups per slice, and the slices meet through one atomic per column
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ncb = (N + 63) / 64, cb = cblk % ncb, slice = cblk / ncb;
  const bool single = R <= kColRoleRows;
  x += (long)slice * kColRoleRows * ldx;
  dy += (long)slice * kColRoleRows * lddy;
  mean_in += (long)slice * kColRoleRows;
  rstd_in += (long)slice * kColRoleRows;
  R -= (long)slice * kColRoleRows;
  if (R > kColRoleRows) R = kColRoleRows;
  const int c = cb * 64 + lane;
  const bool ok = c < N;
  const int cc = ok ? c : 0;
  const float g = gamma[cc], b = beta[cc];
  float pg = 0.f, pb = 0.f;
  for (int r0 = wave; r0 < (int)R; r0 += 4 * kColBatch) {
    float xv[kColBatch], dv[kColBatch], mu[kColBatch], rs[kColBatch];
#pragma unroll
    for (int i = 0; i < kColBatch; ++i) {  // clamped rows: branch-free loads, all in flight together
      const int r = r0 + 4 * i, rr = r < (int)R ? r : (int)R - 1;
      xv[i] = x[(long)rr * ldx + cc];
      dv[i] = dy[(long)rr * lddy + cc];
      mu[i] = mean_in[rr];
      rs[i] = rstd_in[rr];
    }
#pragma unroll
    for (int i = 0; i < kColBatch; ++i) {
      const float xh = (xv[i] - mu[i]) * rs[i];
      float dz = (r0 + 4 * i < (int)R) ? dv[i] : 0.f;
      if (act) dz *= dsiluf_(xh * g + b);
      pg += dz * xh;
      pb += dz;
    }
  }
  lds[wave * 128 + lane] = pg;
  lds[wave * 128 + 64 + lane] = pb;
  __syncthreads();
  if (wave == 0 && ok) {
    const float sg = lds[lane] + lds[128 + lane] + lds[256 + lane] + lds[384 + lane];
    const float sb = lds[64 + lane] + lds[192 + lane] + lds[320 + lane] + lds[448 + lane];
    if (single) {
      dgamma[c] += sg;
      dbeta[c] += sb;
    } else {
      atomicAdd(dgamma + c, sg);
      atomicAdd(dbeta + c, sb);
    }
  }
}

__device__ __forceinline__ void gru_bwd_cols(const float* __restrict__ dhn, long lddhn, const float* __restrict__ p,
                                             long ldp, const float* __restrict__ gamma, const float* __restrict__ beta,
                                             const float* __restrict__ h, long ldh, const float* __restrict__ mean_in,
                                             const float* __restrict__ rstd_in, float* __restrict__ dgamma,
                                             float* __restrict__ dbeta, int M, int De, int cblk, float* lds) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ncb = De / 64, cb = cblk % ncb, slice = cblk / ncb;  // row slices as in ln_bwd_cols
  const bool single = M <= kColRoleRows;
  p += (long)slice * kColRoleRows * ldp;
  h += (long)slice * kColRoleRows * ldh;
  dhn += (long)slice * kColRoleRows * lddhn;
  mean_in += slice * kColRoleRows;
  rstd_in += slice * kColRoleRows;
  M -= slice * kColRoleRows;
  if (M > kColRoleRows) M = kColRoleRows;
  const int j = cb * 64 + lane;  // hidden unit; De % 64 == 0
  const float gr = gamma[j], gc = gamma[De + j], gu = gamma[2 * De + j];
  const float br = beta[j], bc = beta[De + j], bu = beta[2 * De + j];
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r0 = wave; r0 < M; r0 += 4 * kColBatch) {
    float pr[kColBatch], pc[kColBatch], pu[kColBatch], hp[kColBatch], go[kColBatch], mu[kColBatch], rs[kColBatch];
#pragma unroll
    for (int i = 0; i < kColBatch; ++i) {
      const int r = r0 + 4 * i, rr = r < M ? r : M - 1;
      const float* row = p + (long)rr * ldp;
      pr[i] = row[j];
      pc[i] = row[De + j];
      pu[i] = row[2 * De + j];
      hp[i] = h[(long)rr * ldh + j];
      go[i] = dhn[(long)rr * lddhn + j];
      mu[i] = mean_in[rr];
      rs[i] = rstd_in[rr];
    }
#pragma unroll
    for (int i = 0; i < kColBatch; ++i) {
      const float xr = (pr[i] - mu[i]) * rs[i], xc = (pc[i] - mu[i]) * rs[i], xu = (pu[i] - mu[i]) * rs[i];
      const float yr = xr * gr + br, yc = xc * gc + bc, yu = xu * gu + bu;
      const float rg = sigmoidf_(yr), cg = tanhf(rg * yc), ug = sigmoidf_(yu - 1.f);
      const float g = (r0 + 4 * i < M) ? go[i] : 0.f;
      const float drc = g * ug * (1.f - cg * cg);
      const float dyr = drc * yc * rg * (1.f - rg), dyc = drc * rg, dyu = g * (cg - hp[i]) * ug * (1.f - ug);
      acc[0] += dyr * xr; acc[1] += dyc * xc; acc[2] += dyu * xu;
      acc[3] += dyr; acc[4] += dyc; acc[5] += dyu;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) lds[(wave * 6 + k) * 64 + lane] = acc[k];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float v = lds[k * 64 + lane] + lds[(6 + k) * 64 + lane] + lds[(12 + k) * 64 + lane] + lds[(18 + k) * 64 + lane];
      float* dst = (k < 3 ? dgamma : dbeta) + (k % 3) * De + j;
      if (single) *dst += v;
      else atomicAdd(dst, v);
    }
  }
}

template <int LPR, int NV>
__global__ __launch_bounds__(256) void ln_act_bwd_vec_kernel(const float* __restrict__ dy, long lddy,
                                                             const float* __restrict__ x, long ldx,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in,
                                                             float* __restrict__ dx, long lddx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             long R, int N, int act, int accumulate_dx, int row_blocks,
                                                             float* __restrict__ part) {
  constexpr int RPB = 256 / LPR;
  const int sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  __shared__ __attribute__((aligned(16))) float red[2][64 * kMaxV];
  if ((int)blockIdx.x >= row_blocks) {  // column role: owns 64 columns of d-gamma / d-beta
    ln_bwd_cols(dy, lddy, x, ldx, gamma, beta, mean_in, rstd_in, dgamma, dbeta, R, N, act, (int)blockIdx.x - row_blocks,
                &red[0][0]);
    return;
  }
  const bool flush = dgamma && (int)gridDim.x == row_blocks;
  if (flush) for (int c = threadIdx.x; c < N; c += 256) {
    red[0][c] = 0.f;
    red[1][c] = 0.f;
  }
  __syncthreads();
  f4 g[NV], b[NV], pg[NV], pb[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = 4 * (l + LPR * v);
    const bool ok = c < N;
    g[v] = ok ? *reinterpret_cast<const f4u*>(gamma + c) : (f4){0.f, 0.f, 0.f, 0.f};
    b[v] = ok ? *reinterpret_cast<const f4u*>(beta + c) : (f4){0.f, 0.f, 0.f, 0.f};
    pg[v] = (f4){0.f, 0.f, 0.f, 0.f};
    pb[v] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  const float inv_n = 1.f / (float)N;
  for (long r = (long)blockIdx.x * RPB + sub; r < R; r += (long)row_blocks * RPB) {
    const float mean = mean_in[r], rstd = rstd_in[r];
    f4 xh[NV], dxh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = 4 * (l + LPR * v);
      xh[v] = (f4){0.f, 0.f, 0.f, 0.f};
      dxh[v] = (f4){0.f, 0.f, 0.f, 0.f};
      if (c < N) {
        const f4 xhat = (*reinterpret_cast<const f4u*>(x + r * ldx + c) - mean) * rstd;
        f4 dz = *reinterpret_cast<const f4u*>(dy + r * lddy + c);
        if (act) {
          const f4 z = xhat * g[v] + b[v];
          dz.x *= dsiluf_(z.x); dz.y *= dsiluf_(z.y); dz.z *= dsiluf_(z.z); dz.w *= dsiluf_(z.w);
        }
        pg[v] += dz * xhat;
        pb[v] += dz;
        xh[v] = xhat;
        dxh[v] = dz * g[v];
        s1 += hsum(dxh[v]);
        s2 += hsum(dxh[v] * xhat);
      }
    }
    s1 = group_sum<LPR>(s1) * inv_n;
    s2 = group_sum<LPR>(s2) * inv_n;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = 4 * (l + LPR * v);
      if (c < N) {
        f4 d = (dxh[v] - s1 - xh[v] * s2) * rstd;
        f4u* o = reinterpret_cast<f4u*>(dx + r * lddx + c);
        if (accumulate_dx) d += *o;
        *o = d;
      }
    }
  }
  if (flush && part != nullptr) {
    // two-stage form (dv3_ln_act_bwd_ws, N <= 512): this workgroup's column sums go to ITS row of the partial buffer
    // with plain stores; ln_fold_partials_kernel adds the rows up.  (All row blocks adding onto the same N addresses
    // instead serialise in the memory-side atomic unit -- 2048 x 2 N atomics on four cache lines at N = 64 -- and the
    // ds_add_f32 reduction in front of them costs ~4 us per workgroup at N = 512 (4096 lane-atomics): together +40 us on
    // a 37 us kernel.)  The block's own
    // reduction without atomics: the lane groups of a wave hold the same columns for different rows (xor shuffles over
    // the group stride), then the four waves meet in LDS ([wave][2][N] floats = the 16 KB of `red`).
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) {
        pg[v].x += __shfl_xor(pg[v].x, off); pg[v].y += __shfl_xor(pg[v].y, off);
        pg[v].z += __shfl_xor(pg[v].z, off); pg[v].w += __shfl_xor(pg[v].w, off);
        pb[v].x += __shfl_xor(pb[v].x, off); pb[v].y += __shfl_xor(pb[v].y, off);
        pb[v].z += __shfl_xor(pb[v].z, off); pb[v].w += __shfl_xor(pb[v].w, off);
      }
    }
    float* w4 = &red[0][0];
    const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
    __syncthreads();  // (the zeroing pass above used red)
    if (wl < LPR) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = 4 * (wl + LPR * v);
        if (c < N) {
          *reinterpret_cast<f4*>(w4 + (long)(wave * 2 + 0) * N + c) = pg[v];
          *reinterpret_cast<f4*>(w4 + (long)(wave * 2 + 1) * N + c) = pb[v];
        }
      }
    }
    __syncthreads();
    float* row = part + (long)blockIdx.x * 2 * N;
    for (int c = threadIdx.x; c < 2 * N; c += 256)
      row[c] = (w4[c] + w4[2 * N + c]) + (w4[4 * N + c] + w4[6 * N + c]);
    return;
  }
  if (flush) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = 4 * (l + LPR * v);
      if (c < N) {
        atomicAdd(&red[0][c], pg[v].x); atomicAdd(&red[0][c + 1], pg[v].y);
        atomicAdd(&red[0][c + 2], pg[v].z); atomicAdd(&red[0][c + 3], pg[v].w);
        atomicAdd(&red[1][c], pb[v].x); atomicAdd(&red[1][c + 1], pb[v].y);
        atomicAdd(&red[1][c + 2], pb[v].z); atomicAdd(&red[1][c + 3], pb[v].w);
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 256) {
      atomicAdd(dgamma + c, red[0][c]);
      atomicAdd(dbeta + c, red[1][c]);
    }
  }
}

// Second stage of dv3_ln_act_bwd_ws: part [nb][2 N] -> dgamma / dbeta (+=).  grid (column blocks of 64, row slices);
// lane = column, the 4 waves split the slice's rows, 8 rows in flight per lane; one atomic per column and slice.
constexpr int kFoldSlices = 16;
__global__ __launch_bounds__(256) void ln_fold_partials_kernel(const float* __restrict__ part, int nb, int N,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + lane, W = 2 * N;
  const bool ok = c < W;
  const int cc = ok ? c : 0;
  const int per = (nb + (int)gridDim.y - 1) / (int)gridDim.y;
  const int r_beg = blockIdx.y * per, r_end = min(nb, r_beg + per);
  float sum = 0.f;
  for (int r0 = r_beg + wave; r0 < r_end; r0 += 32) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = r0 + 4 * k;
      v[k] = (r < r_end) ? part[(long)r * W + cc] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += v[k];
  }
  red[wave][lane] = sum;
  __syncthreads();
  if (wave == 0 && ok) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    atomicAdd((c < N) ? dgamma + c : dbeta + (c - N), t);
  }
}

// Optional second output of the kernels that finish a state of the observe scan: the NEXT step's input after its
// reset blend, out[r] = v*(1 - first[r]) + init*first[r] (init = the learned initial state, one row).
struct NextBlend {
  const float* first;  // [rows] is_first of the next step
  const float* init;   // [cols]
  float* out;          // nullptr = no second output
  long ld;
};

// GRU gates, vectorised: De % 256 == 0, NVG = De/256 float4 chunks per gate per lane.  Lane l owns chunks
// l + 64*v of each gate, so r, c, u of one hidden unit sit in the same lane.  One wave per row, whole row
// of p (3*De floats) in registers, single pass.
template <int NVG>
__global__ __launch_bounds__(256) void gru_fwd_vec_kernel(const float* __restrict__ p, long ldp,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ h,
                                                          long ldh, float* __restrict__ hn, long ldhn,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          int M, int De, NextBlend nb) {
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const float inv_n = 1.f / (float)(3 * De);
  for (int r = blockIdx.x * 4 + wave; r < M; r += gridDim.x * 4) {
    const float* pr = p + (long)r * ldp;
    f4 pv[3][NVG];
    float s = 0.f;
#pragma unroll
    for (int gte = 0; gte < 3; ++gte)
#pragma unroll
      for (int v = 0; v < NVG; ++v) {
        pv[gte][v] = *reinterpret_cast<const f4u*>(pr + gte * De + 4 * (l + 64 * v));
        s += hsum(pv[gte][v]);
      }
    const float mean = group_sum<64>(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int gte = 0; gte < 3; ++gte)
#pragma unroll
      for (int v = 0; v < NVG; ++v) {
        const f4 d = pv[gte][v] - mean;
        q += hsum(d * d);
      }
    const float rstd = rsqrtf(group_sum<64>(q) * inv_n + kLnEps);
#pragma unroll
    for (int v = 0; v < NVG; ++v) {
      const int j = 4 * (l + 64 * v);
      const f4 yr = (pv[0][v] - mean) * rstd * *reinterpret_cast<const f4u*>(gamma + j) + *reinterpret_cast<const f4u*>(beta + j);
      const f4 yc = (pv[1][v] - mean) * rstd * *reinterpret_cast<const f4u*>(gamma + De + j) + *reinterpret_cast<const f4u*>(beta + De + j);
      const f4 yu = (pv[2][v] - mean) * rstd * *reinterpret_cast<const f4u*>(gamma + 2 * De + j) + *reinterpret_cast<const f4u*>(beta + 2 * De + j);
      const f4 hp = *reinterpret_cast<const f4u*>(h + (long)r * ldh + j);
      f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float rg = sigmoidf_(yr[e]);
        const float cg = tanhf(rg * yc[e]);
        const float ug = sigmoidf_(yu[e] - 1.f);
        o[e] = ug * cg + (1.f - ug) * hp[e];
      }
      *reinterpret_cast<f4u*>(hn + (long)r * ldhn + j) = o;
      if (nb.out) {  // the next observe step's reset blend of this state (networks.py:183-191), fused here
        const float m = nb.first[r];
        *reinterpret_cast<f4u*>(nb.out + (long)r * nb.ld + j) = o * (1.f - m) + *reinterpret_cast<const f4u*>(nb.init + j) * m;
      }
    }
    if (l == 0) {
      mean_out[r] = mean;
      rstd_out[r] = rstd;
    }
  }
}

template <int NVG>
__global__ __launch_bounds__(256) void gru_bwd_vec_kernel(const float* __restrict__ dhn, long lddhn,
                                                          const float* __restrict__ p, long ldp,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ h,
                                                          long ldh, const float* __restrict__ mean_in,
                                                          const float* __restrict__ rstd_in, float* __restrict__ dp,
                                                          long lddp, float* __restrict__ dh, long lddh,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                          int De, int accumulate_dh, int row_blocks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 x 3*De block accumulators (dgamma, dbeta)
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int N = 3 * De;
  float* accg = smem;
  float* accb = smem + N;
  if ((int)blockIdx.x >= row_blocks) {  // column role: owns 64 hidden units of d-gamma / d-beta
    gru_bwd_cols(dhn, lddhn, p, ldp, gamma, beta, h, ldh, mean_in, rstd_in, dgamma, dbeta, M, De,
                 (int)blockIdx.x - row_blocks, smem);
    return;
  }
  const bool flush = dgamma && (int)gridDim.x == row_blocks;
  if (flush) {
    for (int c = threadIdx.x; c < 2 * N; c += 256) smem[c] = 0.f;
    __syncthreads();
  }
  const float inv_n = 1.f / (float)N;
  f4 pg[3][NVG], pb[3][NVG];
#pragma unroll
  for (int gte = 0; gte < 3; ++gte)
#pragma unroll
    for (int v = 0; v < NVG; ++v) {
      pg[gte][v] = (f4){0.f, 0.f, 0.f, 0.f};
      pb[gte][v] = (f4){0.f, 0.f, 0.f, 0.f};
    }
  for (int r = blockIdx.x * 4 + wave; r < M; r += row_blocks * 4) {
    const float* pr = p + (long)r * ldp;
    const float mean = mean_in[r], rstd = rstd_in[r];
    f4 xh[3][NVG], dy[3][NVG];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NVG; ++v) {
      const int j = 4 * (l + 64 * v);
      f4 gm[3], bt[3];
#pragma unroll
      for (int gte = 0; gte < 3; ++gte) {
        xh[gte][v] = (*reinterpret_cast<const f4u*>(pr + gte * De + j) - mean) * rstd;
        gm[gte] = *reinterpret_cast<const f4u*>(gamma + gte * De + j);
        bt[gte] = *reinterpret_cast<const f4u*>(beta + gte * De + j);
      }
      const f4 hp = *reinterpret_cast<const f4u*>(h + (long)r * ldh + j);
      const f4 go = *reinterpret_cast<const f4u*>(dhn + (long)r * lddhn + j);
      f4 dhd;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yr = xh[0][v][e] * gm[0][e] + bt[0][e];
        const float yc = xh[1][v][e] * gm[1][e] + bt[1][e];
        const float yu = xh[2][v][e] * gm[2][e] + bt[2][e];
        const float rg = sigmoidf_(yr);
        const float cg = tanhf(rg * yc);
        const float ug = sigmoidf_(yu - 1.f);
        const float g = go[e];
        const float du = g * (cg - hp[e]) * ug * (1.f - ug);
        const float drc = g * ug * (1.f - cg * cg);
        dy[0][v][e] = drc * yc * rg * (1.f - rg);
        dy[1][v][e] = drc * rg;
        dy[2][v][e] = du;
        dhd[e] = g * (1.f - ug);
      }
      f4u* o = reinterpret_cast<f4u*>(dh + (long)r * lddh + j);
      if (accumulate_dh) dhd += *o;
      *o = dhd;
#pragma unroll
      for (int gte = 0; gte < 3; ++gte) {
        const f4 dxh = dy[gte][v] * gm[gte];
        s1 += hsum(dxh);
        s2 += hsum(dxh * xh[gte][v]);
        pg[gte][v] += dy[gte][v] * xh[gte][v];
        pb[gte][v] += dy[gte][v];
        dy[gte][v] = dxh;  // keep dy*gamma for the second pass
      }
    }
    s1 = group_sum<64>(s1) * inv_n;
    s2 = group_sum<64>(s2) * inv_n;
#pragma unroll
    for (int gte = 0; gte < 3; ++gte)
#pragma unroll
      for (int v = 0; v < NVG; ++v) {
        const int j = 4 * (l + 64 * v);
        *reinterpret_cast<f4u*>(dp + (long)r * lddp + gte * De + j) = (dy[gte][v] - s1 - xh[gte][v] * s2) * rstd;
      }
  }
  if (flush) {
#pragma unroll
    for (int gte = 0; gte < 3; ++gte)
#pragma unroll
      for (int v = 0; v < NVG; ++v) {
        const int c = gte * De + 4 * (l + 64 * v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          atomicAdd(accg + c + e, pg[gte][v][e]);
          atomicAdd(accb + c + e, pb[gte][v][e]);
        }
      }
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 256) {
      atomicAdd(dgamma + c, accg[c]);
      atomicAdd(dbeta + c, accb[c]);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Wide GRU cells (De = 1024 * NK > 1024: the crafter-size models, dyn_deter 2048..4096, i.e. LayerNorm rows of
// 6144..12288): a workgroup per row, thread t owns columns 4*(t + 256 k) .. +3 of each gate (k < NK), the row's
// 3*De pre-activations stay in registers, statistics by two block reductions.  Same math as gru_fwd_kernel.
// ------------------------------------------------------------------------------------------------
template <int NK>
__global__ __launch_bounds__(256) void gru_fwd_wide_kernel(const float* __restrict__ p, long ldp,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ h,
                                                           long ldh, float* __restrict__ hn, long ldhn,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int M, int De, NextBlend nb) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int N = 3 * De;
  const float inv_n = 1.f / (float)N;
  for (int r = blockIdx.x; r < M; r += gridDim.x) {
    const float* pr = p + (long)r * ldp;
    f4 x[3][NK];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        x[g][k] = *reinterpret_cast<const f4u*>(pr + (long)g * De + 4 * (tid + 256 * k));
        s += hsum(x[g][k]);
      }
    const float mean = block_sum_256(s, red) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const f4 d = x[g][k] - mean;
        q += hsum(d * d);
      }
    const float rstd = rsqrtf(block_sum_256(q, red) * inv_n + kLnEps);
    const float nf = nb.out ? nb.first[r] : 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int j = 4 * (tid + 256 * k);
      const f4 gr = *reinterpret_cast<const f4u*>(gamma + j), br = *reinterpret_cast<const f4u*>(beta + j);
      const f4 gc = *reinterpret_cast<const f4u*>(gamma + De + j), bc = *reinterpret_cast<const f4u*>(beta + De + j);
      const f4 gu = *reinterpret_cast<const f4u*>(gamma + 2 * De + j), bu = *reinterpret_cast<const f4u*>(beta + 2 * De + j);
      const f4 hp = *reinterpret_cast<const f4u*>(h + (long)r * ldh + j);
      f4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yr = (x[0][k][e] - mean) * rstd * gr[e] + br[e];
        const float yc = (x[1][k][e] - mean) * rstd * gc[e] + bc[e];
        const float yu = (x[2][k][e] - mean) * rstd * gu[e] + bu[e];
        const float rg = sigmoidf_(yr);
        const float cg = tanhf(rg * yc);
        const float ug = sigmoidf_(yu - 1.f);
        o[e] = ug * cg + (1.f - ug) * hp[e];
      }
      *reinterpret_cast<f4u*>(hn + (long)r * ldhn + j) = o;
      if (nb.out) {
        const f4 in = *reinterpret_cast<const f4u*>(nb.init + j);
        *reinterpret_cast<f4u*>(nb.out + (long)r * nb.ld + j) = o * (1.f - nf) + in * nf;
      }
    }
    if (tid == 0) {
      mean_out[r] = mean;
      rstd_out[r] = rstd;
    }
  }
}

// Backward of the wide cell.  dgamma/dbeta partial sums stay in registers over the workgroup's rows (a thread owns
// the same columns in every row) and leave with one atomic per column per workgroup.
template <int NK>
__global__ __launch_bounds__(256) void gru_bwd_wide_kernel(const float* __restrict__ dhn, long lddhn,
                                                           const float* __restrict__ p, long ldp,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ h,
                                                           long ldh, const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, float* __restrict__ dp,
                                                           long lddp, float* __restrict__ dh, long lddh,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                           int De, int accumulate_dh) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int N = 3 * De;
  const float inv_n = 1.f / (float)N;
  f4 ag[3][NK], ab[3][NK];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      ag[g][k] = (f4){0.f, 0.f, 0.f, 0.f};
      ab[g][k] = (f4){0.f, 0.f, 0.f, 0.f};
    }
  for (int r = blockIdx.x; r < M; r += gridDim.x) {
    const float* pr = p + (long)r * ldp;
    const float mean = mean_in[r], rstd = rstd_in[r];
    f4 xh[3][NK], dy[3][NK];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int j = 4 * (tid + 256 * k);
      const f4 gr = *reinterpret_cast<const f4u*>(gamma + j), br = *reinterpret_cast<const f4u*>(beta + j);
      const f4 gc = *reinterpret_cast<const f4u*>(gamma + De + j), bc = *reinterpret_cast<const f4u*>(beta + De + j);
      const f4 gu = *reinterpret_cast<const f4u*>(gamma + 2 * De + j), bu = *reinterpret_cast<const f4u*>(beta + 2 * De + j);
      xh[0][k] = (*reinterpret_cast<const f4u*>(pr + j) - mean) * rstd;
      xh[1][k] = (*reinterpret_cast<const f4u*>(pr + De + j) - mean) * rstd;
      xh[2][k] = (*reinterpret_cast<const f4u*>(pr + 2 * De + j) - mean) * rstd;
      const f4 hp = *reinterpret_cast<const f4u*>(h + (long)r * ldh + j);
      const f4 gg = *reinterpret_cast<const f4u*>(dhn + (long)r * lddhn + j);
      f4 dhd;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yr = xh[0][k][e] * gr[e] + br[e], yc = xh[1][k][e] * gc[e] + bc[e], yu = xh[2][k][e] * gu[e] + bu[e];
        const float rg = sigmoidf_(yr);
        const float cg = tanhf(rg * yc);
        const float ug = sigmoidf_(yu - 1.f);
        const float g = gg[e];
        const float du = g * (cg - hp[e]) * ug * (1.f - ug);
        const float drc = g * ug * (1.f - cg * cg);
        const float dr = drc * yc * rg * (1.f - rg);
        const float dc = drc * rg;
        dhd[e] = g * (1.f - ug);
        dy[0][k][e] = dr;
        dy[1][k][e] = dc;
        dy[2][k][e] = du;
        s1 += dr * gr[e] + dc * gc[e] + du * gu[e];
        s2 += dr * gr[e] * xh[0][k][e] + dc * gc[e] * xh[1][k][e] + du * gu[e] * xh[2][k][e];
      }
      f4u* o = reinterpret_cast<f4u*>(dh + (long)r * lddh + j);
      if (accumulate_dh) dhd += *o;
      *o = dhd;
    }
    s1 = block_sum_256(s1, red) * inv_n;
    s2 = block_sum_256(s2, red) * inv_n;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int c = g * De + 4 * (tid + 256 * k);
        const f4 gm = *reinterpret_cast<const f4u*>(gamma + c);
        *reinterpret_cast<f4u*>(dp + (long)r * lddp + c) = (dy[g][k] * gm - s1 - xh[g][k] * s2) * rstd;
        ag[g][k] += dy[g][k] * xh[g][k];
        ab[g][k] += dy[g][k];
      }
  }
  if (dgamma) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int c = g * De + 4 * (tid + 256 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          atomicAdd(dgamma + c + e, ag[g][k][e]);
          atomicAdd(dbeta + c + e, ab[g][k][e]);
        }
      }
  }
}

template <int LPR, int NV>
static void launch_ln_fwd(const float* x, long ldx, const float* g, const float* b, float* y, long ldy, float* mean,
                          float* rstd, long R, int N, int act, int G, hipStream_t s) {
  const long rpb = 256 / LPR;
  long blocks = (R + rpb - 1) / rpb;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((ln_act_fwd_kernel<LPR, NV>), dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, g, b, y, ldy, mean,
                     rstd, R, N, act, G);
}
template <int LPR, int NV>
static void launch_ln_bwd(const float* dy, long lddy, const float* x, long ldx, const float* g, const float* b,
                          const float* mean, const float* rstd, float* dx, long lddx, float* dg, float* db, long R,
                          int N, int act, int G, int acc, hipStream_t s) {
  const long rpb = 256 / LPR;
  long blocks = (R + rpb - 1) / rpb;
  if (blocks > 512) blocks = 512;  // fewer blocks -> fewer dgamma/dbeta atomics
  hipLaunchKernelGGL((ln_act_bwd_kernel<LPR, NV>), dim3((unsigned)blocks), dim3(256), 0, s, dy, lddy, x, ldx, g, b, mean,
                     rstd, dx, lddx, dg, db, R, N, act, G, acc);
}
static int pick_lpr(int N) {
  int lpr = 4;
  while (lpr < 64 && lpr < N) lpr <<= 1;
  return lpr;
}
// dispatch on (lanes per row, elements per lane): sub-wave groups for short rows (conv channels),
// a full wave with 1..32 cached elements per lane for long ones
#define DV3_LN_DISPATCH(FN, ...)                                   \
  do {                                                             \
    const int lpr_ = pick_lpr(N);                                  \
    if (lpr_ == 4) FN<4, 1>(__VA_ARGS__);                          \
    else if (lpr_ == 8) FN<8, 1>(__VA_ARGS__);                     \
    else if (lpr_ == 16) FN<16, 1>(__VA_ARGS__);                   \
    else if (lpr_ == 32) FN<32, 1>(__VA_ARGS__);                   \
    else {                                                         \
      const int nv_ = (N + 63) / 64;                               \
      if (nv_ <= 1) FN<64, 1>(__VA_ARGS__);                        \
      else if (nv_ <= 2) FN<64, 2>(__VA_ARGS__);                   \
      else if (nv_ <= 4) FN<64, 4>(__VA_ARGS__);                   \
      else if (nv_ <= 8) FN<64, 8>(__VA_ARGS__);                   \
      else if (nv_ <= 16) FN<64, 16>(__VA_ARGS__);                 \
      else FN<64, 32>(__VA_ARGS__);                                \
    }                                                              \
  } while (0)

}  // namespace dv3

using namespace dv3;

// (lanes per row, float4 chunks per lane) for the vectorised kernels
static bool vec_shape(int N, int& lpr, int& nv) {
  if (N % 4 != 0 || N < 16) return false;
  const int chunks = N / 4;
  lpr = 4;
  while (lpr < 64 && lpr < chunks) lpr <<= 1;
  nv = (chunks + lpr - 1) / lpr;
  return nv <= 8;
}
#define DV3_LNV_DISPATCH(KERNEL, GRIDCAP, EXTRA, ...)                                                     \
  do {                                                                                               \
    int lpr_, nv_;                                                                                   \
    vec_shape(N, lpr_, nv_);                                                                         \
    const long rpb_ = 256 / lpr_;                                                                    \
    long blocks_ = (R + rpb_ - 1) / rpb_;                                                            \
    if (blocks_ > (GRIDCAP)) blocks_ = (GRIDCAP);                                                    \
    const dim3 g_((unsigned)(blocks_ + (EXTRA))), b_(256);                                           \
    if (lpr_ == 4) hipLaunchKernelGGL((KERNEL<4, 1>), g_, b_, 0, s, __VA_ARGS__);                    \
    else if (lpr_ == 8) hipLaunchKernelGGL((KERNEL<8, 1>), g_, b_, 0, s, __VA_ARGS__);               \
    else if (lpr_ == 16) hipLaunchKernelGGL((KERNEL<16, 1>), g_, b_, 0, s, __VA_ARGS__);             \
    else if (lpr_ == 32) hipLaunchKernelGGL((KERNEL<32, 1>), g_, b_, 0, s, __VA_ARGS__);             \
    else if (nv_ <= 1) hipLaunchKernelGGL((KERNEL<64, 1>), g_, b_, 0, s, __VA_ARGS__);               \
    else if (nv_ <= 2) hipLaunchKernelGGL((KERNEL<64, 2>), g_, b_, 0, s, __VA_ARGS__);               \
    else if (nv_ <= 4) hipLaunchKernelGGL((KERNEL<64, 4>), g_, b_, 0, s, __VA_ARGS__);               \
    else hipLaunchKernelGGL((KERNEL<64, 8>), g_, b_, 0, s, __VA_ARGS__);                             \
  } while (0)

extern "C" int dv3_ln_act_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy,
                              float* mean, float* rstd, long R, int N, int act, int chw_group, void* stream) {
  if (R <= 0) return 0;
  if (N <= 0 || N > 64 * kMaxV || !x || !y || !gamma || !beta) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  {
    int lpr0, nv0;
    if (chw_group <= 0 && vec_shape(N, lpr0, nv0)) {
      DV3_LNV_DISPATCH(ln_act_fwd_vec_kernel, 8192, 0, x, ldx, gamma, beta, y, ldy, mean, rstd, R, N, act);
      return (int)hipGetLastError();
    }
  }
  DV3_LN_DISPATCH(launch_ln_fwd, x, ldx, gamma, beta, y, ldy, mean, rstd, R, N, act, chw_group, s);
  return (int)hipGetLastError();
}

constexpr long kLnPartRows = 2048;  // row blocks of the vectorised backward (its grid cap) = rows of the partial buffer

static int ln_act_bwd_impl(const float* dy, long lddy, const float* x, long ldx, const float* gamma, const float* beta,
                           const float* mean, const float* rstd, float* dx, long lddx, float* dgamma, float* dbeta,
                           long R, int N, int act, int chw_group, int accumulate_dx, float* ws, long ws_floats,
                           void* stream) {
  if (R <= 0) return 0;
  if (N <= 0 || N > 64 * kMaxV || !x || !dy || !dx || !gamma || !beta || !mean || !rstd) return DV3_ERR_ARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return DV3_ERR_ARG;
  if (ws && ws_floats < kLnPartRows * 2 * N) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  {
    int lpr0, nv0;
    if (chw_group <= 0 && vec_shape(N, lpr0, nv0)) {
      // d-gamma / d-beta WITHOUT a workspace: every block ends with N atomics per array onto the same N addresses:
      // few, fat blocks (256) keep that contention off the critical path for N >= 128; wide rows (N >= 256) get
      // column-role workgroups instead (they re-read the inputs); narrow rows keep the wide grid and pay the
      // serialised atomics.  (Never the column role when dx overwrites dy in place: it re-reads dy while the row
      // role writes dx.)  WITH a workspace (dv3_ln_act_bwd_ws): partial rows + a second, column-parallel launch.
      const bool two_stage = ws != nullptr && dgamma != nullptr;
      const bool col_role = !two_stage && dgamma && dx != dy && N >= 256;
      const long cap = (!two_stage && dgamma && !col_role && N >= 128) ? 256 : kLnPartRows;
      const int col_blocks = col_role ? ((N + 63) / 64) * (int)((R + kColRoleRows - 1) / kColRoleRows) : 0;
      float* part = two_stage ? ws : nullptr;
      int nb = 0;
      DV3_LNV_DISPATCH(ln_act_bwd_vec_kernel, cap, col_blocks, dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dgamma,
                       dbeta, R, N, act, accumulate_dx, (nb = (int)blocks_), part);
      if (two_stage) {
        const int slices = nb >= 256 ? kFoldSlices : 1;
        hipLaunchKernelGGL(ln_fold_partials_kernel, dim3((2 * N + 63) / 64, slices), dim3(256), 0, s, part, nb, N, dgamma,
                           dbeta);
      }
      return (int)hipGetLastError();
    }
  }
  DV3_LN_DISPATCH(launch_ln_bwd, dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dgamma, dbeta, R, N, act, chw_group,
                  accumulate_dx, s);
  return (int)hipGetLastError();
}

extern "C" int dv3_ln_act_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                              const float* beta, const float* mean, const float* rstd, float* dx, long lddx,
                              float* dgamma, float* dbeta, long R, int N, int act, int chw_group, int accumulate_dx,
                              void* stream) {
  return ln_act_bwd_impl(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dgamma, dbeta, R, N, act, chw_group,
                         accumulate_dx, nullptr, 0, stream);
}

extern "C" int dv3_ln_act_bwd_ws(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                                 const float* beta, const float* mean, const float* rstd, float* dx, long lddx,
                                 float* dgamma, float* dbeta, long R, int N, int act, int chw_group, int accumulate_dx,
                                 float* ws, long ws_floats, void* stream) {
  return ln_act_bwd_impl(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dgamma, dbeta, R, N, act, chw_group,
                         accumulate_dx, ws, ws_floats, stream);
}

static int gru_fwd_impl(const float* p, long ldp, const float* gamma, const float* beta, const float* h, long ldh,
                        float* h_new, long ldhn, float* mean, float* rstd, int M, int De, NextBlend nb, void* stream);

extern "C" int dv3_gru_fwd(const float* p, long ldp, const float* gamma, const float* beta, const float* h, long ldh,
                           float* h_new, long ldhn, float* mean, float* rstd, int M, int De, void* stream) {
  return gru_fwd_impl(p, ldp, gamma, beta, h, ldh, h_new, ldhn, mean, rstd, M, De, NextBlend{nullptr, nullptr, nullptr, 0},
                      stream);
}

extern "C" int dv3_gru_fwd_blend(const float* p, long ldp, const float* gamma, const float* beta, const float* h,
                                 long ldh, float* h_new, long ldhn, float* mean, float* rstd, int M, int De,
                                 const float* next_first, const float* init, float* next_out, long ld_next,
                                 void* stream) {
  if (!next_first || !init || !next_out || ld_next < De || ld_next % 4 != 0) return DV3_ERR_ARG;
  if (!((De % 256 == 0 && De <= 1024) || (De % 1024 == 0 && De <= 4096))) return DV3_ERR_ARG;
  return gru_fwd_impl(p, ldp, gamma, beta, h, ldh, h_new, ldhn, mean, rstd, M, De,
                      NextBlend{next_first, init, next_out, ld_next}, stream);
}

static int gru_fwd_impl(const float* p, long ldp, const float* gamma, const float* beta, const float* h, long ldh,
                        float* h_new, long ldhn, float* mean, float* rstd, int M, int De, NextBlend nb, void* stream) {
  if (M <= 0) return 0;
  if (De <= 0 || !p || !gamma || !beta || !h || !h_new || !mean || !rstd) return DV3_ERR_ARG;
  int blocks = (M + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (De % 256 == 0 && De <= 1024) {
    hipStream_t s = (hipStream_t)stream;
    if (De == 256) hipLaunchKernelGGL((gru_fwd_vec_kernel<1>), dim3(blocks), dim3(256), 0, s, p, ldp, gamma, beta, h, ldh, h_new, ldhn, mean, rstd, M, De, nb);
    else if (De == 512) hipLaunchKernelGGL((gru_fwd_vec_kernel<2>), dim3(blocks), dim3(256), 0, s, p, ldp, gamma, beta, h, ldh, h_new, ldhn, mean, rstd, M, De, nb);
    else if (De == 768) hipLaunchKernelGGL((gru_fwd_vec_kernel<3>), dim3(blocks), dim3(256), 0, s, p, ldp, gamma, beta, h, ldh, h_new, ldhn, mean, rstd, M, De, nb);
    else hipLaunchKernelGGL((gru_fwd_vec_kernel<4>), dim3(blocks), dim3(256), 0, s, p, ldp, gamma, beta, h, ldh, h_new, ldhn, mean, rstd, M, De, nb);
    return (int)hipGetLastError();
  }
  if (De % 1024 == 0 && De <= 4096 && ldp % 4 == 0 && ldh % 4 == 0 && ldhn % 4 == 0) {
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(M < 8192 ? M : 8192), block(256);
    if (De == 2048) hipLaunchKernelGGL((gru_fwd_wide_kernel<2>), grid, block, 0, s, p, ldp, gamma, beta, h, ldh, h_new, ldhn, mean, rstd, M, De, nb);
    else if (De == 3072) hipLaunchKernelGGL((gru_fwd_wide_kernel<3>), grid, block, 0, s, p, ldp, gamma, beta, h, ldh, h_new, ldhn, mean, rstd, M, De, nb);
    else hipLaunchKernelGGL((gru_fwd_wide_kernel<4>), grid, block, 0, s, p, ldp, gamma, beta, h, ldh, h_new, ldhn, mean, rstd, M, De, nb);
    return (int)hipGetLastError();
  }
  if (nb.out) return DV3_ERR_ARG;
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, ldp, gamma, beta, h, ldh,
                     h_new, ldhn, mean, rstd, M, De);
  return (int)hipGetLastError();
}

extern "C" int dv3_gru_bwd(const float* dh_new, long lddhn, const float* p, long ldp, const float* gamma,
                           const float* beta, const float* h, long ldh, const float* mean, const float* rstd,
                           float* dp, long lddp, float* dh, long lddh, float* dgamma, float* dbeta, int M, int De,
                           int accumulate_dh, void* stream) {
  if (M <= 0) return 0;
  if (De <= 0 || !dh_new || !p || !gamma || !beta || !h || !mean || !rstd || !dp || !dh) return DV3_ERR_ARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return DV3_ERR_ARG;
  if (De > 1024 && De % 1024 == 0 && De <= 4096 && ldp % 4 == 0 && lddp % 4 == 0 && ldh % 4 == 0 && lddh % 4 == 0 &&
      lddhn % 4 == 0) {
    hipStream_t s = (hipStream_t)stream;
    // with parameter gradients few fat workgroups (their 2 * 3 De atomics each are the tail); without, the whole chip
    const int cap = dgamma ? 256 : 4096;
    const dim3 grid(M < cap ? M : cap), block(256);
#define DV3_GRUBW(NK_) hipLaunchKernelGGL((gru_bwd_wide_kernel<NK_>), grid, block, 0, s, dh_new, lddhn, p, ldp, gamma, beta, h, \
                                          ldh, mean, rstd, dp, lddp, dh, lddh, dgamma, dbeta, M, De, accumulate_dh)
    if (De == 2048) DV3_GRUBW(2);
    else if (De == 3072) DV3_GRUBW(3);
    else DV3_GRUBW(4);
#undef DV3_GRUBW
    return (int)hipGetLastError();
  }
  const size_t shmem = (size_t)6 * 3 * De * sizeof(float);
  if (shmem > 150 * 1024) return DV3_ERR_ARG;  // generic kernel: De <= 2133 (wider cells: De % 1024 == 0 above)
  int blocks = (M + 3) / 4;
  if (blocks > 256) blocks = 256;
  if (De % 256 == 0 && De <= 1024) {
    hipStream_t s = (hipStream_t)stream;
    const size_t sh = (size_t)2 * 3 * De * sizeof(float);
    const int col_blocks =
        (dgamma && dh != dh_new && dh != h && dp != p) ? (De / 64) * ((M + kColRoleRows - 1) / kColRoleRows) : 0;
#define DV3_GRUB(NVG_) hipLaunchKernelGGL((gru_bwd_vec_kernel<NVG_>), dim3(blocks + col_blocks), dim3(256), sh, s, dh_new, lddhn, p, ldp, \
                                          gamma, beta, h, ldh, mean, rstd, dp, lddp, dh, lddh, dgamma, dbeta, M, De, accumulate_dh, blocks)
    if (De == 256) DV3_GRUB(1);
    else if (De == 512) DV3_GRUB(2);
    else if (De == 768) DV3_GRUB(3);
    else DV3_GRUB(4);
#undef DV3_GRUB
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(blocks), dim3(256), shmem, (hipStream_t)stream, dh_new, lddhn, p, ldp, gamma,
                     beta, h, ldh, mean, rstd, dp, lddp, dh, lddh, dgamma, dbeta, M, De, accumulate_dh);
  return (int)hipGetLastError();
}

// dst = s0 | s1 | ... | s5 (flat, float): the acting step's outputs packed in one launch for a single D2H hop
struct Concat6 {
  const float* src[6];
  long end[6];  // exclusive prefix ends
};
__global__ void concat6_kernel(Concat6 c, float* __restrict__ dst) {
  const long total = c.end[5];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int k = 0;
    long base = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (e >= c.end[j]) {
        k = j + 1;
        base = c.end[j];
      }
    dst[e] = c.src[k][e - base];
  }
}

extern "C" int dv3_concat6(const float* s0, long n0, const float* s1, long n1, const float* s2, long n2, const float* s3,
                           long n3, const float* s4, long n4, const float* s5, long n5, float* dst, void* stream) {
  const float* src[6] = {s0, s1, s2, s3, s4, s5};
  const long n[6] = {n0, n1, n2, n3, n4, n5};
  Concat6 c;
  long acc = 0;
  for (int j = 0; j < 6; ++j) {
    if (n[j] < 0 || (n[j] > 0 && !src[j])) return DV3_ERR_ARG;
    acc += n[j];
    c.src[j] = src[j];
    c.end[j] = acc;
  }
  if (acc == 0) return 0;
  if (!dst) return DV3_ERR_ARG;
  unsigned blocks = (unsigned)((acc + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(concat6_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c, dst);
  return (int)hipGetLastError();
}
