// Dense fp32 GEMM on v_mfma_f32_32x32x2_f32 -- every Linear layer of the hot path.
//
//   C[M,N] (+)= [A | A2][M,K] * op(B)[K,N] + bias[N]
//
// Replaces, on the reference side, the ATen mm/addmm launches issued by nn.Linear inside
// networks.RSSM.img_step / obs_step (networks.py:195-233), networks.GRUCell.forward
// (networks.py:760-768), networks.MLP.forward (networks.py:657-681) and
// networks.ConvDecoder._linear_layer (networks.py:569), plus their autograd transposes:
//   forward  y = x W^T   : transA=0 transB=1   (W is [N][K], k-contiguous)
//   dgrad    dx = dy W   : transA=0 transB=0   (W is [K=N_out][N=K_in])
//   wgrad    dW = dy^T x : transA=1 transB=0   (A = dy [rows][N_out], B = x [rows][K_in])
// The two-segment A ([A | A2] along K) is the reference's torch.cat([...], -1) in front of the
// Linear (networks.py:196, 216, 762; get_feat networks.py:154-159), done without materialising it.
#include <stdlib.h>

#include "mfma_gemm.h"
#include "dv3_common.h"

namespace dv3 {

struct GemmParams {
  const float* A;
  const float* A2;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K, K1;
  long lda, lda2, ldb, ldc;
  int accumulate;
  int vecA, vecB;
  int tiles_m, tiles_n;
  int splits, kchunk;  // split-K over workgroups (accumulating GEMMs only: partial sums land with atomics)
  // l16 kernel: output columns >= nsplit go to C2[row * ldc2 + (col - nsplit)] with their own accumulate flag
  // (one product feeding two destinations: the GRU matmul's data gradient = [dx | dh])
  float* C2;
  long ldc2;
  int nsplit, accumulate2;
  int xcd_m, xcd_n;    // direct / l16 kernels: the 8 XCDs as an xcd_m x xcd_n grid of tile blocks (0: linear order)
  int transC;          // skinny kernel only: store C[col*ldc + row] and index bias by row
  // direct kernel, EPI = 1: C holds the logits of N/32 categorical groups of 32 classes per row; each group is
  // sampled in the epilogue (tools.OneHotDist.sample, tools.py:452-460; same draws as dv3_onehot_sample_fwd)
  const float* smp_noise;
  const unsigned long long* smp_rng;
  unsigned long long smp_off;
  float* smp_onehot;
  int* smp_idx;
  const int* smp_forced;
  unsigned int* smp_flips;
  float smp_unimix;
  int smp_mode;
  // direct kernel, ALN = 1: A holds PRE-activations; SiLU(LayerNorm(A) * gamma + beta) is applied as the operand is
  // loaded (row statistics computed by the workgroup itself over its 32 rows), so the LayerNorm of the producing
  // layer needs no launch of its own.  ln_mean / ln_rstd [M] receive the statistics (saved for the backward).
  const float* ln_gamma;
  const float* ln_beta;
  float* ln_mean;
  float* ln_rstd;
};

template <class TS, bool TA, bool TB>
__global__ __launch_bounds__(kThreads) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float lds[TS::lds_floats];
  // A operand: rows = M.  TA=false -> A[m][k] (k-contiguous).  TA=true -> A[k][m].
  using ATile = DenseTile<TS::BM, TS::BK, !TA>;
  // B operand: rows = N.  TB=true -> B[n][k] (k-contiguous).  TB=false -> B[k][n].
  using BTile = DenseTile<TS::BN, TS::BK, TB>;
  DenseOperand<!TA> aop{p.A, p.A2, p.lda, p.lda2, p.M, p.K, p.K1, p.vecA != 0};
  DenseOperand<TB> bop{p.B, nullptr, p.ldb, 0, p.N, p.K, p.K, p.vecB != 0};

  const int tiles = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, tiles * p.splits);
  const int wg = id % tiles, split = id / tiles;
  // consecutive workgroups walk N fastest: neighbours share the A row panel in L2
  const int m0 = (wg / p.tiles_n) * TS::BM;
  const int n0 = (wg % p.tiles_n) * TS::BN;
  const int kb = split * p.kchunk;
  const int ke = (p.splits > 1) ? min(p.K, kb + p.kchunk) : p.K;

  f32x16 acc[TS::TM][TS::TN];
  bool owner;
  mfma_mainloop<TS, ATile, BTile>(aop, bop, m0, n0, kb, ke, lds, acc, owner);
  if (kb >= ke || !owner) return;

  const int tid = threadIdx.x;
  const int wave = (tid >> 6) % (TS::WM * TS::WN), lane = tid & 63;
  const int wm = wave / TS::WN, wn = wave % TS::WN;
  const int col_l = lane & 31, h = lane >> 5;
#pragma unroll
  for (int a = 0; a < TS::TM; ++a) {
#pragma unroll
    for (int b = 0; b < TS::TN; ++b) {
      const int n = n0 + (wn * TS::TN + b) * 32 + col_l;
      if (n >= p.N) continue;
      const float bv = (p.bias && split == 0) ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TS::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M) {
          float* c = p.C + (long)m * p.ldc + n;
          const float v = acc[a][b][r] + bv;
          if (p.splits > 1) atomicAdd(c, v);
          else *c = p.accumulate ? (*c + v) : v;
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Grouped weight gradients: up to kMaxGroup independent products C_g (+)= A_g^T B_g (A_g [K_g][M_g], B_g [K_g][N_g], both
// pixel / row major as the backward pass holds them) in ONE grid.  The world model's seventeen K = B*T = 1024 weight
// gradients have 64 ... 1024 output tiles of 64 x 64 each: launched one by one they need split-K with atomics to cover
// the chip and still pay a launch boundary per 20-60 us of work (0.56 ms per update on a 128-CU lane at 32 % MFMA-busy);
// as one grid every tile walks its full K range (no atomics: the sums are reproducible) and the lane stays full.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxGroup = 48;

struct GroupItem {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  int tiles_n, tile0, accumulate;
};

struct GroupParams {
  int n, total;
  GroupItem it[kMaxGroup];
};

template <class TS>
__global__ __launch_bounds__(kThreads) void gemm_tn_grouped_kernel(GroupParams p) {
  __shared__ __attribute__((aligned(16))) float lds[TS::lds_floats];
  using ATile = DenseTile<TS::BM, TS::BK, false>;
  using BTile = DenseTile<TS::BN, TS::BK, false>;
  const int id = xcd_remap(blockIdx.x, p.total);
  int g = 0;
  while (g + 1 < p.n && id >= p.it[g + 1].tile0) ++g;  // (uniform over the workgroup: scalar loads of the kernel arguments)
  const GroupItem it = p.it[g];
  const int wg = id - it.tile0;
  const int m0 = (wg / it.tiles_n) * TS::BM;
  const int n0 = (wg % it.tiles_n) * TS::BN;
  DenseOperand<false> aop{it.A, nullptr, it.lda, 0, it.M, it.K, it.K, true};
  DenseOperand<false> bop{it.B, nullptr, it.ldb, 0, it.N, it.K, it.K, true};
  f32x16 acc[TS::TM][TS::TN];
  bool owner;
  mfma_mainloop<TS, ATile, BTile>(aop, bop, m0, n0, 0, it.K, lds, acc, owner);
  if (!owner) return;
  const int tid = threadIdx.x;
  const int wave = (tid >> 6) % (TS::WM * TS::WN), lane = tid & 63;
  const int wm = wave / TS::WN, wn = wave % TS::WN;
  const int col_l = lane & 31, h = lane >> 5;
#pragma unroll
  for (int a = 0; a < TS::TM; ++a) {
#pragma unroll
    for (int b = 0; b < TS::TN; ++b) {
      const int n = n0 + (wn * TS::TN + b) * 32 + col_l;
      if (n >= it.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TS::TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < it.M) {
          float* c = it.C + (long)m * it.ldc + n;
          *c = it.accumulate ? (*c + acc[a][b][r]) : acc[a][b][r];
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Skinny GEMM for the observe scan: M <= 32 rows (the replay batch) against a full weight matrix.
// Weight-streaming bound (every step re-reads ~11 MB of weights from L2 / Infinity Cache), so the
// shape is one 16-column output tile per workgroup with K split over its 4 waves (LDS reduce):
// N/16 workgroups stream disjoint weight rows at full width, operands go straight from global memory
// to the v_mfma_f32_16x16x4_f32 operand registers (no LDS staging: nothing is reused across waves).
// Per 16-k chunk a lane (i = l&15, q = l>>4) loads A[i][k+4q..+3] and B[n0+i][k+4q..+3] (16 B each)
// and issues 4 MFMAs; element g of both fragments is k = k+4q+g, so the k-order is permuted
// identically for A and B (still an exact fp32 fma chain, in a different order).
// ------------------------------------------------------------------------------------------------
constexpr int kSkinnyWaves = 8;   // K is split over the waves of a workgroup (4 when K is also split over workgroups)
// kSkinnyBatch (template): 16-k chunks whose loads a wave issues together before their MFMAs -- sized by the host
// so that a wave's whole K share is ONE batch where registers allow (one memory round trip per launch)

template <bool TB, int MT, int kSkinnyBatch>
__global__ __launch_bounds__(64 * kSkinnyWaves) void gemm_skinny_kernel(GemmParams p) {
  __shared__ float red[kSkinnyWaves][MT][256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nwaves = blockDim.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int rb = blockIdx.z * (16 * MT);  // row block (gridDim.z > 1: up to 128 rows, e.g. the acting step's encoder)
  // K range of this workgroup (gridDim.y splits, accumulating calls only), then of this wave, in 16-k chunks
  const int chunks = (p.K + 15) >> 4;
  const int per_wg = (chunks + gridDim.y - 1) / gridDim.y;
  const int wcb = blockIdx.y * per_wg, wce = min(chunks, wcb + per_wg);
  const int per = (max(wce - wcb, 0) + nwaves - 1) / nwaves;
  const int cb = wcb + wave * per, ce = min(wce, cb + per);
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ncol = n0 + i;
  const bool colok = ncol < p.N;
  const float* bcol = p.B + (TB ? (long)(colok ? ncol : 0) * p.ldb : (long)(colok ? ncol : 0));
  const float bmask = colok ? 1.f : 0.f;
  float amask[MT];
  const float* arow[MT];
  const float* arow2[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int row = rb + i + 16 * t;
    const bool ok = row < p.M;
    amask[t] = ok ? 1.f : 0.f;
    arow[t] = p.A + (long)(ok ? row : 0) * p.lda;
    arow2[t] = p.A2 ? p.A2 + (long)(ok ? row : 0) * p.lda2 : nullptr;
  }
  for (int c0 = cb; c0 < ce; c0 += kSkinnyBatch) {
    f32x4 a[kSkinnyBatch][MT], b[kSkinnyBatch];
    // chunks of the batch beyond this wave's range are clamped onto its last chunk (loaded, never multiplied);
    // the fast path needs every loaded 4-element group inside K and inside one A segment
    const int clast = ce - 1;
    const int cend = min(c0 + kSkinnyBatch, ce);
    const bool fast = (cend * 16 <= p.K) && ((c0 * 16 >= p.K1) || (cend * 16 <= p.K1));
    if (fast) {
      const bool seg2 = c0 * 16 >= p.K1;
#pragma unroll
      for (int u = 0; u < kSkinnyBatch; ++u) {
        const int k = (min(c0 + u, clast) << 4) + 4 * q;
#pragma unroll
        for (int t = 0; t < MT; ++t)
          a[u][t] = *reinterpret_cast<const f32x4u*>(seg2 ? arow2[t] + (k - p.K1) : arow[t] + k);
        if (TB) {
          b[u] = *reinterpret_cast<const f32x4u*>(bcol + k);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) b[u][e] = bcol[(long)(k + e) * p.ldb];
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < kSkinnyBatch; ++u) {
        const int c = c0 + u;
        const int k = (c << 4) + 4 * q;
        const bool seg2 = k >= p.K1;
        const int ka = seg2 ? k - p.K1 : k;
        const int kend = seg2 ? p.K - p.K1 : p.K1;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          const float* src = (seg2 ? arow2[t] : arow[t]) + ka;
          if (c < ce) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (ka + e < kend) v[e] = src[e];
          }
          a[u][t] = v;
        }
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (c < ce) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < p.K) w[e] = TB ? bcol[k + e] : bcol[(long)(k + e) * p.ldb];
        }
        b[u] = w;
      }
    }
#pragma unroll
    for (int u = 0; u < kSkinnyBatch; ++u)
      if (c0 + u < ce) {  // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int t = 0; t < MT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t][g] * amask[t], b[u][g] * bmask, acc[t], 0, 0, 0);
      }
  }
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][t][r * 64 + lane] = acc[t][r];
  __syncthreads();
  // finish the MT*256 outputs: element x = r*64 + lane -> row 4*(lane>>4)+r (+16 t), col lane&15
  const bool split = gridDim.y > 1;
  for (int e = tid; e < MT * 256; e += blockDim.x) {
    const int t = e >> 8, x = e & 255;
    const int r = x >> 6, l = x & 63;
    const int row = rb + 16 * t + 4 * (l >> 4) + r, col = n0 + (l & 15);
    if (row < p.M && col < p.N) {
      float v = 0.f;
      for (int w = 0; w < nwaves; ++w) v += red[w][t][x];
      if (p.bias && blockIdx.y == 0) v += p.bias[p.transC ? row : col];
      float* o = p.transC ? p.C + (long)col * p.ldc + row : p.C + (long)row * p.ldc + col;
      if (split) {
        atomicAdd(o, v);  // accumulate semantics: C already holds the value being added to
      } else {
        if (p.accumulate) v += *o;
        *o = v;
      }
    }
  }
}

// XCD-aware tile order for the 1-D grids of the direct / l16 kernels.  Workgroup ids go round-robin to the 8 XCDs,
// each with a private L2 that pulls its own copy of every operand panel its workgroups touch: with the XCDs laid
// out as an xcd_m x xcd_n grid of tile blocks, the fabric sees the A panel xcd_n times and the B panel xcd_m times.
// The host picks the factorisation of 8 that minimises bytes(A) * xcd_n + bytes(B) * xcd_m among those that divide
// the tile counts (pick_xcd_grid); e.g. 1024 x 1536 x 1024: 2 x 4 -> 28 MB, against 52 MB for row strips (8 x 1).
__device__ __forceinline__ void xcd_tile(const GemmParams& p, int id, int& tm, int& tn) {
  if (p.xcd_m > 0) {
    const int xcd = id & 7, loc = id >> 3;
    const int bm = p.tiles_m / p.xcd_m, bn = p.tiles_n / p.xcd_n;
    tm = (xcd / p.xcd_n) * bm + loc / bn;
    tn = (xcd % p.xcd_n) * bn + loc % bn;
  } else {
    tm = id / p.tiles_n;
    tn = id % p.tiles_n;
  }
}

static void pick_xcd_grid(GemmParams& p) {
  p.xcd_m = p.xcd_n = 0;
  static const int env_xm = DV3_ENV_INT("DV3_XCD_M", 0);  // development: force the grid
  if (env_xm == -1) return;  // linear tile order
  if (env_xm > 0 && 8 % env_xm == 0 && p.tiles_m % env_xm == 0 && p.tiles_n % (8 / env_xm) == 0) {
    p.xcd_m = env_xm;
    p.xcd_n = 8 / env_xm;
    return;
  }
  const double a = (double)p.M * p.K, b = (double)p.N * p.K;
  double best = 0.0;
  for (int xm = 8; xm >= 1; xm >>= 1) {
    const int xn = 8 / xm;
    if (p.tiles_m % xm || p.tiles_n % xn) continue;
    const double cost = a * xn + b * xm;
    if (p.xcd_m == 0 || cost < best) {
      best = cost;
      p.xcd_m = xm;
      p.xcd_n = xn;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Register-direct GEMM for mid-size outputs (the 1024 imagination rows x 512..1536 columns): the same
// no-LDS scheme as the few-row kernel, tiled over M.  A workgroup owns 32 rows x 16*RN columns; its waves split
// K (LDS reduce at the end); per 16-k chunk a wave loads 2 A fragments and RN B fragments (16 bytes per lane,
// 64 contiguous bytes per matrix row) and issues 8*RN v_mfma_f32_16x16x4_f32 -- each loaded fragment is used
// RN (A) or 2 (B) times from registers, which is what the LDS staging of the tile engine buys at this size,
// without its LDS write+read of every element and its barriers.  TB: B is [N][K] (else [K][N], scalar loads).
// grid = ceil(N / (16*RN)) * ceil(M / 32) workgroups (XCD-aware order, see below); blockDim = 64 * waves (4 or 8).
// ------------------------------------------------------------------------------------------------
// PIPE = 1: the loads of batch t+1 are issued before the MFMAs of batch t (two register sets).  PIPE = 0: one register
// set, a batch is loaded then consumed; latency is hidden by the other resident waves only -- with BATCH = 2 a wave
// then consumes whole 128-byte lines of its 16 operand rows per batch (two adjacent 16-k chunks) at ~110 VGPRs.
template <bool TB, int RN, int BATCH, int EPI = 0, int PIPE = 1, int ALN = 0>
__global__ __launch_bounds__(512) void gemm_direct_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float dred[];  // [waves][2][RN][256]
  __shared__ float ctile[EPI == 1 ? 32 * (16 * RN + 1) : 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nwaves = blockDim.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  int tm, tn;
  xcd_tile(p, blockIdx.x, tm, tn);  // XCD-aware order of the tiles_m x tiles_n tiles (see xcd_tile)
  const int n0 = tn * (16 * RN), m0 = tm * 32;
  const int chunks = (p.K + 15) >> 4;
  const int per = (chunks + nwaves - 1) / nwaves;
  const int cb = wave * per, ce = min(chunks, cb + per);
  f32x4 acc[2][RN];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < RN; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* bcol[RN];
  float bmask[RN];
#pragma unroll
  for (int c = 0; c < RN; ++c) {
    const int ncol = n0 + 16 * c + i;
    const bool ok = ncol < p.N;
    bcol[c] = p.B + (TB ? (long)(ok ? ncol : 0) * p.ldb : (long)(ok ? ncol : 0));
    bmask[c] = ok ? 1.f : 0.f;
  }
  float amask[2];
  const float* arow[2];
  const float* arow2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = m0 + i + 16 * t;
    const bool ok = row < p.M;
    amask[t] = ok ? 1.f : 0.f;
    arow[t] = p.A + (long)(ok ? row : 0) * p.lda;
    arow2[t] = p.A2 ? p.A2 + (long)(ok ? row : 0) * p.lda2 : nullptr;
  }
  float lmean[2] = {0.f, 0.f}, lrstd[2] = {1.f, 1.f};
  f32x4 gk[BATCH], bk[BATCH];
  if constexpr (ALN) {
    __shared__ float lnst[32][2];
    const float inv_k = 1.f / (float)p.K;
    for (int rr = wave; rr < 32; rr += nwaves) {  // wave-uniform
      const int row = m0 + rr;
      if (row < p.M) {
        const float* ar = p.A + (long)row * p.lda;
        float sm = 0.f;
        for (int k = 4 * lane; k < p.K; k += 256) {
          const f32x4 v = *reinterpret_cast<const f32x4u*>(ar + k);
          sm += (v[0] + v[1]) + (v[2] + v[3]);
        }
        const float mean = group_sum<64>(sm) * inv_k;
        float qv = 0.f;
        for (int k = 4 * lane; k < p.K; k += 256) {
          const f32x4 v = *reinterpret_cast<const f32x4u*>(ar + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) qv += (v[e] - mean) * (v[e] - mean);
        }
        const float rstd = rsqrtf(group_sum<64>(qv) * inv_k + kLnEps);
        if (lane == 0) {
          lnst[rr][0] = mean;
          lnst[rr][1] = rstd;
          if (tn == 0) {
            if (p.ln_mean) p.ln_mean[row] = mean;
            if (p.ln_rstd) p.ln_rstd[row] = rstd;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (m0 + i + 16 * t < p.M) {
        lmean[t] = lnst[i + 16 * t][0];
        lrstd[t] = lnst[i + 16 * t][1];
      }
  }
  // software pipeline: the loads of batch t+1 are issued before the MFMAs of batch t (two register sets)
  f32x4 a0[BATCH][2], b0[BATCH][RN], a1[BATCH][2], b1[BATCH][RN];
  const int clast = ce - 1;
  auto load = [&](f32x4 (&a)[BATCH][2], f32x4 (&b)[BATCH][RN], int c0) {
    if (c0 >= ce) return;
    const int cend = min(c0 + BATCH, ce);
    const bool fast = (cend * 16 <= p.K) && ((c0 * 16 >= p.K1) || (cend * 16 <= p.K1));
    if (fast) {
      const bool seg2 = c0 * 16 >= p.K1;
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int k = (min(c0 + u, clast) << 4) + 4 * q;
#pragma unroll
        for (int t = 0; t < 2; ++t)
          a[u][t] = *reinterpret_cast<const f32x4u*>(seg2 ? arow2[t] + (k - p.K1) : arow[t] + k);
        if constexpr (ALN) {
          gk[u] = *reinterpret_cast<const f32x4u*>(p.ln_gamma + k);
          bk[u] = *reinterpret_cast<const f32x4u*>(p.ln_beta + k);
        }
#pragma unroll
        for (int c = 0; c < RN; ++c) {
          if (TB) {
            b[u][c] = *reinterpret_cast<const f32x4u*>(bcol[c] + k);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) b[u][c][e] = bcol[c][(long)(k + e) * p.ldb];
          }
        }
      }
    } else {  // ragged end of K or a segment edge inside the batch
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int cc = c0 + u;
        const int k = (cc << 4) + 4 * q;
        const bool seg2 = k >= p.K1;
        const int ka = seg2 ? k - p.K1 : k;
        const int kend = seg2 ? p.K - p.K1 : p.K1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          const float* src = (seg2 ? arow2[t] : arow[t]) + ka;
          if (cc < ce) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (ka + e < kend) v[e] = src[e];
          }
          a[u][t] = v;
        }
        if constexpr (ALN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            gk[u][e] = (cc < ce && k + e < p.K) ? p.ln_gamma[k + e] : 0.f;
            bk[u][e] = (cc < ce && k + e < p.K) ? p.ln_beta[k + e] : 0.f;
          }
        }
#pragma unroll
        for (int c = 0; c < RN; ++c) {
          f32x4 w = {0.f, 0.f, 0.f, 0.f};
          if (cc < ce) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (k + e < p.K) w[e] = TB ? bcol[c][k + e] : bcol[c][(long)(k + e) * p.ldb];
          }
          b[u][c] = w;
        }
      }
    }
  };
  auto compute = [&](const f32x4 (&a)[BATCH][2], const f32x4 (&b)[BATCH][RN], int c0) {
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (c0 + u < ce) {  // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            float av = a[u][t][g];
            if constexpr (ALN) av = siluf_((av - lmean[t]) * lrstd[t] * gk[u][g] + bk[u][g]);
            av *= amask[t];
#pragma unroll
            for (int c = 0; c < RN; ++c)
              acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u][c][g] * bmask[c], acc[t][c], 0, 0, 0);
          }
      }
  };
  if constexpr (PIPE) {
    load(a0, b0, cb);
    for (int c0 = cb; c0 < ce; c0 += 2 * BATCH) {
      load(a1, b1, c0 + BATCH);
      compute(a0, b0, c0);
      load(a0, b0, c0 + 2 * BATCH);
      compute(a1, b1, c0 + BATCH);
    }
  } else {
    for (int c0 = cb; c0 < ce; c0 += BATCH) {
      load(a0, b0, c0);
      compute(a0, b0, c0);
    }
  }
  float* red = dred + (long)wave * (2 * RN * 256);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < RN; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(t * RN + c) * 256 + r * 64 + lane] = acc[t][c][r];
  __syncthreads();
  // element x = r*64 + l of tile (t, c): row 16 t + 4 (l >> 4) + r, column 16 c + (l & 15)
  for (int e = tid; e < 2 * RN * 256; e += blockDim.x) {
    const int tc = e >> 8, x = e & 255;
    const int t = tc / RN, c = tc % RN;
    const int r = x >> 6, l = x & 63;
    const int row = m0 + 16 * t + 4 * (l >> 4) + r, col = n0 + 16 * c + (l & 15);
    if (row < p.M && col < p.N) {
      float v = 0.f;
      for (int w = 0; w < nwaves; ++w) v += dred[(long)w * (2 * RN * 256) + e];
      if (p.bias) v += p.bias[col];
      float* o = p.C + (long)row * p.ldc + col;
      if (p.accumulate) v += *o;
      *o = v;
      if constexpr (EPI == 1) ctile[(16 * t + 4 * (l >> 4) + r) * (16 * RN + 1) + 16 * c + (l & 15)] = v;
    }
  }
  if constexpr (EPI == 1) {
    // the finished 32 x (16 RN) logit tile holds RN/2 whole categorical groups per row: 32 lanes per group
    constexpr int GT = 16 * RN / 32;
    __syncthreads();
    const int S = p.N >> 5;
    const int d = tid & 31;
    unsigned long long seed = 0, offset = 0;
    if (!p.smp_mode && !p.smp_noise) {
      seed = p.smp_rng[0];
      offset = p.smp_rng[1] + p.smp_off;
    }
    // 32 * GT (row, group) pairs over blockDim / 32 lane groups: UN pairs per lane group are worked on together so
    // that their dependent shuffle chains (softmax max / sum, argmax) overlap instead of running back to back
    constexpr int UN = 4;
    const int ngrp = blockDim.x >> 5;
    for (int pr0 = tid >> 5; pr0 < 32 * GT; pr0 += UN * ngrp) {
      float lg[UN], sc[UN];
      long gi[UN];
      bool rv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int pr = min(pr0 + u * ngrp, 32 * GT - 1);
        const int row_l = pr / GT, gl = pr % GT;
        const int row = m0 + row_l, col0 = n0 + 32 * gl;
        rv[u] = (pr0 + u * ngrp < 32 * GT) && row < p.M && col0 < p.N;
        lg[u] = ctile[row_l * (16 * RN + 1) + 32 * gl + d];
        gi[u] = (long)row * S + (col0 >> 5);
      }
      float mx[UN], ex[UN], sm_[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) mx[u] = lg[u];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < UN; ++u) mx[u] = fmaxf(mx[u], __shfl_xor(mx[u], o, 64));
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        ex[u] = expf(lg[u] - mx[u]);
        sm_[u] = ex[u];
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < UN; ++u) sm_[u] += __shfl_xor(sm_[u], o, 64);
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const float ph = (ex[u] / sm_[u]) * (1.f - p.smp_unimix) + p.smp_unimix / 32.f;
        float score = ph;
        if (!p.smp_mode) {
          float q;
          if (p.smp_noise) {
            q = rv[u] ? p.smp_noise[gi[u] * 32 + d] : 1.f;
          } else {
            uint32_t o4[4];
            const unsigned long long e = (unsigned long long)gi[u] * 32 + d;
            Philox ph4(seed);
            ph4(offset + (e >> 2), 0x5eedULL, o4);
            q = fmaxf(-logf(u01(o4[e & 3])), 1e-30f);
          }
          score = ph / q;
        }
        sc[u] = score;
      }
      int bi[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) bi[u] = d;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const float ob = __shfl_xor(sc[u], o, 64);
          const int oi = __shfl_xor(bi[u], o, 64);
          if (ob > sc[u] || (ob == sc[u] && oi < bi[u])) {
            sc[u] = ob;
            bi[u] = oi;
          }
        }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (p.smp_forced) {
          const int f = rv[u] ? p.smp_forced[gi[u]] : 0;
          if (rv[u] && d == 0 && p.smp_flips && f != bi[u]) atomicAdd(p.smp_flips, 1u);
          bi[u] = f;
        }
        if (rv[u]) {
          p.smp_onehot[gi[u] * 32 + d] = (d == bi[u]) ? 1.f : 0.f;
          if (d == 0 && p.smp_idx) p.smp_idx[gi[u]] = bi[u];
        }
      }
    }
  }
}

template <bool TB, int RN>
static void launch_direct_rn(const GemmParams& p0, int waves, hipStream_t s) {
  GemmParams p = p0;
  p.tiles_n = (p.N + 16 * RN - 1) / (16 * RN);
  p.tiles_m = (p.M + 31) / 32;
  pick_xcd_grid(p);
  const dim3 grid(p.tiles_m * p.tiles_n), block(64 * waves);
  const size_t sh = (size_t)waves * 2 * RN * 256 * sizeof(float);
  // one chunk of lookahead (BATCH 1) measured best: ~100 VGPRs keep 4 waves per SIMD resident, which hides more
  // latency than deeper register prefetch at 170-230 VGPRs (DV3_DIRECT_BATCH: development switch)
  static const int env_batch = DV3_ENV_INT("DV3_DIRECT_BATCH", 0);
  if constexpr (TB && RN == 4) {
    if (p.smp_onehot) {
      if (p.ln_gamma) hipLaunchKernelGGL((gemm_direct_kernel<TB, RN, 2, 1, 0, 1>), grid, block, sh, s, p);
      else hipLaunchKernelGGL((gemm_direct_kernel<TB, RN, 2, 1, 0>), grid, block, sh, s, p);
      return;
    }
  }
  // measured (r02e, hipGraph back-to-back, us): y = x W^T shapes 1024x1536x1024 46.1 -> 43.8, 1024x512x512 10.9 -> 10.1,
  // 1024x1024x512 18.1 -> 17.0 with PIPE 0 / BATCH 2; the [K][N] (data-gradient) form is better pipelined (40.8 vs 43.9)
  static const int env_pipe = DV3_ENV_INT("DV3_DIRECT_PIPE", (TB ? 0 : 1));
  if (env_pipe == 0) {
    if (env_batch == 4) hipLaunchKernelGGL((gemm_direct_kernel<TB, RN, 4, 0, 0>), grid, block, sh, s, p);
    else if (env_batch == 1) hipLaunchKernelGGL((gemm_direct_kernel<TB, RN, 1, 0, 0>), grid, block, sh, s, p);
    else hipLaunchKernelGGL((gemm_direct_kernel<TB, RN, 2, 0, 0>), grid, block, sh, s, p);
    return;
  }
  if (env_batch == 2) hipLaunchKernelGGL((gemm_direct_kernel<TB, RN, 2>), grid, block, sh, s, p);
  else if (env_batch == 3) hipLaunchKernelGGL((gemm_direct_kernel<TB, RN, 3>), grid, block, sh, s, p);
  else hipLaunchKernelGGL((gemm_direct_kernel<TB, RN, 1>), grid, block, sh, s, p);
}
template <bool TB>
static void launch_direct(const GemmParams& p, hipStream_t s) {
  static const int env_waves = DV3_ENV_INT("DV3_DIRECT_WAVES", 0);
  static const int env_rn = DV3_ENV_INT("DV3_DIRECT_RN", 0);
  const int chunks = (p.K + 15) / 16;
  // K over 8 waves once every wave still gets four 16-k chunks (measured: never worse than 4, better at N = 512..1024)
  int waves = chunks >= 32 ? 8 : 4;
  if (env_waves == 4 || env_waves == 8) waves = env_waves;
  int rn = 4;
  if (env_rn == 2 || env_rn == 4 || env_rn == 8) rn = env_rn;
  if (rn == 2) launch_direct_rn<TB, 2>(p, waves, s);
  else if (rn == 8) launch_direct_rn<TB, 8>(p, waves, s);
  else launch_direct_rn<TB, 4>(p, waves, s);
}

// ------------------------------------------------------------------------------------------------
// LDS-tiled y = [A|A2] W^T (+ bias) -- tiles 11-15: the mid-size and large products of the path (the 1024-row
// imagination GEMMs up to 4096 x 12288 x 5120) -- v_mfma_f32_16x16x4_f32 with BOTH operands kept k-contiguous in LDS:
// a lane's MFMA fragment for four consecutive k steps is one ds_read_b128 (row i = l & 15, k = 16 kk + 4 (l >> 4) .. +3)
// and a staged float4 is one ds_write_b128 -- a quarter of the LDS instructions of the k-major 32x32x2 tile engine (no
// transposing ds_write_b32), and every loaded element is reused BN/32 (A) or BM/32 (B) times from registers.
// Row stride 40 floats: the 16 lanes a ds_read_b128 serves together (0-3, 12-15, 20-27 | ...) then cover all 64
// banks exactly once (i*40 + 4q mod 64 is a permutation of the 16 four-bank windows), and the 8 lanes of a
// ds_write_b128 group write 128 contiguous bytes.  One workgroup = 4 waves (2 x 2) = one BM x BN tile, BK = 32,
// double-buffered: global loads of K-tile t+1 are issued before the MFMAs of tile t, one barrier per K-tile.
// Tile shapes 32x64 / 64x64 / 64x96 / 128x128, picked by output size (launch_l16 / pick_tile; measurements in
// dv3hip/ops.py pick_gemm_tile).  Tile order: xcd_tile (each XCD owns a block of tiles chosen by operand bytes).
// Requires K % 32 == 0, K1 % 32 == 0, lda/lda2/ldb % 4 == 0, 16-byte aligned operands (l16_ok).
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int PF, int EPI = 0>
__global__ __launch_bounds__(256) void gemm_l16_kernel(GemmParams p) {
  constexpr int BK = 32, LD = 40;
  constexpr int TMW = BM / 32, TNW = BN / 32;  // 16 x 16 blocks per wave (wave tile = BM/2 x BN/2)
  constexpr int NA = BM * (BK / 4) / 256, NB = BN * (BK / 4) / 256;
  static_assert(BM % 32 == 0 && BN % 32 == 0 && NA >= 1 && NB >= 1, "tile");
  __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  xcd_tile(p, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  // staging: float4 f = tid + 256 j of a [rows][8] grid -> row f >> 3, k offset 4 (f & 7).  Rows past the edge of
  // the matrix read row 0 instead: they only feed outputs past the edge, which the epilogue does not store (no
  // masking of the loaded values, which would make the wave wait for its prefetch as soon as it is issued).
  const float* asrc[NA];
  const float* asrc2[NA];
  const float* bsrc[NB];
  const int c4 = (tid & 7) * 4;
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int row = m0 + ((tid + 256 * j) >> 3);
    asrc[j] = p.A + (long)(row < p.M ? row : 0) * p.lda + c4;
    asrc2[j] = p.A2 ? p.A2 + (long)(row < p.M ? row : 0) * p.lda2 + c4 : nullptr;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int col = n0 + ((tid + 256 * j) >> 3);
    bsrc[j] = p.B + (long)(col < p.N ? col : 0) * p.ldb + c4;
  }
  // PF register sets: the loads of K-tile t + PF are in flight while tile t is multiplied.  Measured (1024-row
  // shapes): PF 2 / 4 equal PF 1 within noise -- the default; the template parameter stays for longer-latency operands.
  f32x4 ra[PF][NA], rb[PF][NB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](f32x4 (&xa)[NA], f32x4 (&xb)[NB], int k0) {
    const bool seg2 = k0 >= p.K1;
#pragma unroll
    for (int j = 0; j < NA; ++j)
      xa[j] = *reinterpret_cast<const f32x4*>(seg2 ? asrc2[j] + (k0 - p.K1) : asrc[j] + k0);
#pragma unroll
    for (int j = 0; j < NB; ++j) xb[j] = *reinterpret_cast<const f32x4*>(bsrc[j] + k0);
  };
  auto lstore = [&](const f32x4 (&xa)[NA], const f32x4 (&xb)[NB], int buf) {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      *reinterpret_cast<f32x4*>(&As[buf][((tid + 256 * j) >> 3) * LD + c4]) = xa[j];
#pragma unroll
    for (int j = 0; j < NB; ++j)
      *reinterpret_cast<f32x4*>(&Bs[buf][((tid + 256 * j) >> 3) * LD + c4]) = xb[j];
  };
  f32x4 acc[TMW][TNW];
#pragma unroll
  for (int a = 0; a < TMW; ++a)
#pragma unroll
    for (int b = 0; b < TNW; ++b) acc[a][b] = zero4;
  const int nk = p.K / BK;
  gload(ra[0], rb[0], 0);
  lstore(ra[0], rb[0], 0);
#pragma unroll
  for (int u = 1; u < PF; ++u)
    if (u < nk) gload(ra[u], rb[u], u * BK);
  __syncthreads();
  const int aoff = (wm * (BM / 2) + i) * LD + 4 * q;
  const int boff = (wn * (BN / 2) + i) * LD + 4 * q;
  for (int t0 = 0; t0 < nk; t0 += PF)
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int t = t0 + u;
    if (t >= nk) break;
    const int cur = t & 1;
    if (t + PF < nk) gload(ra[u], rb[u], (t + PF) * BK);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f32x4 af[TMW], bf[TNW];
#pragma unroll
      for (int a = 0; a < TMW; ++a) af[a] = *reinterpret_cast<const f32x4*>(&As[cur][aoff + 16 * a * LD + 16 * kk]);
#pragma unroll
      for (int b = 0; b < TNW; ++b) bf[b] = *reinterpret_cast<const f32x4*>(&Bs[cur][boff + 16 * b * LD + 16 * kk]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int a = 0; a < TMW; ++a)
#pragma unroll
          for (int b = 0; b < TNW; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][g], bf[b][g], acc[a][b], 0, 0, 0);
    }
    if (t + 1 < nk) lstore(ra[(u + 1) % PF], rb[(u + 1) % PF], cur ^ 1);
    __syncthreads();
  }
  if constexpr (EPI == 1) {
    // Sampling epilogue (as gemm_direct_kernel EPI 1; 32 x 64 tile only): a wave's 16 x 32 tile is one whole
    // categorical group of 16 rows, lane (i, q) holding classes i and 16 + i of rows 4q .. 4q+3, so the softmax,
    // the p_hat / q argmax and the one-hot stay in registers (four xor-shuffles per reduction, the four rows of a
    // lane interleaved).  Same reduction order as dv3_onehot_sample_fwd: class d first meets d ^ 16 (the lane's
    // other register), then d ^ 8 .. d ^ 1 -- the sample is bit-equal to the two-launch form.
    static_assert(BM == 32 && BN == 64, "sampling epilogue: 32 x 64 tile");
    const int S = p.N >> 5;
    const int colb = n0 + wn * 32;
    if (colb >= p.N) return;  // wave-uniform; no barrier follows
    unsigned long long seed = 0, offset = 0;
    if (!p.smp_mode && !p.smp_noise) {
      seed = p.smp_rng[0];
      offset = p.smp_rng[1] + p.smp_off;
    }
    const float bias0 = p.bias ? p.bias[colb + i] : 0.f, bias1 = p.bias ? p.bias[colb + 16 + i] : 0.f;
    float l0[4], l1[4], mx[4], e0[4], e1[4], sm[4], sc[4];
    int bi[4];
    long gi[4];
    bool rv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * 16 + 4 * q + r;
      rv[r] = row < p.M;
      gi[r] = (long)row * S + (colb >> 5);
      l0[r] = acc[0][0][r] + bias0;
      l1[r] = acc[0][1][r] + bias1;
      if (rv[r]) {
        float* o = p.C + (long)row * p.ldc + colb + i;
        o[0] = l0[r];
        o[16] = l1[r];
      }
      mx[r] = fmaxf(l0[r], l1[r]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e0[r] = expf(l0[r] - mx[r]);
      e1[r] = expf(l1[r] - mx[r]);
      sm[r] = e0[r] + e1[r];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) sm[r] += __shfl_xor(sm[r], o, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ph0 = (e0[r] / sm[r]) * (1.f - p.smp_unimix) + p.smp_unimix / 32.f;
      const float ph1 = (e1[r] / sm[r]) * (1.f - p.smp_unimix) + p.smp_unimix / 32.f;
      float s0 = ph0, s1 = ph1;
      if (!p.smp_mode) {
        float q0, q1;
        if (p.smp_noise) {
          q0 = rv[r] ? p.smp_noise[gi[r] * 32 + i] : 1.f;
          q1 = rv[r] ? p.smp_noise[gi[r] * 32 + 16 + i] : 1.f;
        } else {
          uint32_t o4[4];
          const unsigned long long ea = (unsigned long long)gi[r] * 32 + i, eb = ea + 16;
          Philox ph4(seed);
          ph4(offset + (ea >> 2), 0x5eedULL, o4);
          q0 = fmaxf(-logf(u01(o4[ea & 3])), 1e-30f);
          ph4(offset + (eb >> 2), 0x5eedULL, o4);
          q1 = fmaxf(-logf(u01(o4[eb & 3])), 1e-30f);
        }
        s0 = ph0 / q0;
        s1 = ph1 / q1;
      }
      const bool hi = s1 > s0;  // tie: the lower class index
      sc[r] = hi ? s1 : s0;
      bi[r] = hi ? 16 + i : i;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ob = __shfl_xor(sc[r], o, 64);
        const int oi = __shfl_xor(bi[r], o, 64);
        if (ob > sc[r] || (ob == sc[r] && oi < bi[r])) {
          sc[r] = ob;
          bi[r] = oi;
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (p.smp_forced) {
        const int f = rv[r] ? p.smp_forced[gi[r]] : 0;
        if (rv[r] && i == 0 && p.smp_flips && f != bi[r]) atomicAdd(p.smp_flips, 1u);
        bi[r] = f;
      }
      if (rv[r]) {
        float* o = p.smp_onehot + gi[r] * 32 + i;
        o[0] = (i == bi[r]) ? 1.f : 0.f;
        o[16] = (16 + i == bi[r]) ? 1.f : 0.f;
        if (i == 0 && p.smp_idx) p.smp_idx[gi[r]] = bi[r];
      }
    }
    return;
  }
  // accumulator register r of block (a, b): row 16 a + 4 q + r, column 16 b + i
#pragma unroll
  for (int a = 0; a < TMW; ++a)
#pragma unroll
    for (int b = 0; b < TNW; ++b) {
      const int col = n0 + wn * (BN / 2) + 16 * b + i;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
      const bool second = p.C2 && col >= p.nsplit;
      float* const cbase = second ? p.C2 + (col - p.nsplit) : p.C + col;
      const long ldo = second ? p.ldc2 : p.ldc;
      const int accf = second ? p.accumulate2 : p.accumulate;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * (BM / 2) + 16 * a + 4 * q + r;
        if (row < p.M) {
          float* o = cbase + (long)row * ldo;
          float v = acc[a][b][r] + bv;
          if (accf) v += *o;
          *o = v;
        }
      }
    }
}

static bool l16_ok(const GemmParams& p, int transA, int transB);
static void launch_l16_sample(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  p.tiles_m = (p.M + 31) / 32;
  p.tiles_n = (p.N + 63) / 64;
  pick_xcd_grid(p);
  hipLaunchKernelGGL((gemm_l16_kernel<32, 64, 1, 1>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
}

static bool l16_ok(const GemmParams& p, int transA, int transB) {
  return !transA && transB && p.K >= 32 && (p.K % 32) == 0 && (p.K1 % 32) == 0 && (p.lda % 4) == 0 &&
         (!p.A2 || (p.lda2 % 4) == 0) && (p.ldb % 4) == 0 && ((uintptr_t)p.A % 16) == 0 &&
         (!p.A2 || ((uintptr_t)p.A2 % 16) == 0) && ((uintptr_t)p.B % 16) == 0;
}

// tile shape: measured (tools/gemm_bench.py; 128 x 128: 4096^3 1059 us = 130 TFLOP/s against 1228 on the k-major
// 128x128x32 tile), 1024 x 1536 x 1024: 32x64 35.9 us (768 workgroups, three per CU: one
// workgroup's barrier / LDS-fill bubbles are covered by the MFMAs of the others), 64x96 39.8 (one per CU), 64x64 44.3,
// register-direct 44.0; 4096^3: 64x64 and 64x96 115-119 TFLOP/s.  Prefetch distance 1 / 2 / 4 measured equal.
static void launch_l16(const GemmParams& p0, int force, hipStream_t s) {
  GemmParams p = p0;
  auto wgs = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  // tile 11 / the split-output entry: by size, as pick_tile does (128 x 128 from 448 such tiles, 64 x 64 from 512, else 32 x 64)
  int sel = wgs(128, 128) >= 448 ? 4 : wgs(64, 64) >= 512 ? 2 : 3;
  if (force >= 1 && force <= 3) sel = force;
  if (force == 4) sel = 4;  // 128 x 128 (tile 15)
  if (force == 5 || force == 6) sel = force;  // 128 x 64 (tile 16), 64 x 128 (tile 17)
  const int bm = sel == 3 ? 32 : (sel == 4 || sel == 5) ? 128 : 64;
  const int bn = sel == 1 ? 96 : (sel == 4 || sel == 6) ? 128 : 64;
  p.tiles_m = (p.M + bm - 1) / bm;
  p.tiles_n = (p.N + bn - 1) / bn;
  pick_xcd_grid(p);
  const dim3 grid(p.tiles_m * p.tiles_n), block(256);
  static const int pf = DV3_ENV_INT("DV3_L16_PF", 1);
#define DV3_L16_LAUNCH(PFV)                                                                     \
  do {                                                                                          \
    if (sel == 1) hipLaunchKernelGGL((gemm_l16_kernel<64, 96, PFV>), grid, block, 0, s, p);      \
    else if (sel == 2) hipLaunchKernelGGL((gemm_l16_kernel<64, 64, PFV>), grid, block, 0, s, p); \
    else if (sel == 4) hipLaunchKernelGGL((gemm_l16_kernel<128, 128, PFV>), grid, block, 0, s, p); \
    else if (sel == 5) hipLaunchKernelGGL((gemm_l16_kernel<128, 64, PFV>), grid, block, 0, s, p); \
    else if (sel == 6) hipLaunchKernelGGL((gemm_l16_kernel<64, 128, PFV>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((gemm_l16_kernel<32, 64, PFV>), grid, block, 0, s, p);               \
  } while (0)
  if (pf == 2) DV3_L16_LAUNCH(2);
  else DV3_L16_LAUNCH(1);
#undef DV3_L16_LAUNCH
}

// Register-direct weight gradient: C[M,N] += A^T B with A [K][M] and B [K][N] (both row-major over the batch
// rows K): every fragment element is one 4-byte load (a lane's four k values sit in four different rows), 16
// lanes per 64 contiguous bytes.  K is split over gridDim.y workgroups (atomic adds) and then over the waves.
template <int RN, int BATCH>
__global__ __launch_bounds__(256) void gemm_direct_tn_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float dred[];  // [waves][2][RN][256]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nwaves = blockDim.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x % p.tiles_n;
  const int n0 = tn * (16 * RN), m0 = tm * 32;
  const int chunks = (p.K + 15) >> 4;
  const int per_wg = (chunks + gridDim.y - 1) / gridDim.y;
  const int wcb = blockIdx.y * per_wg, wce = min(chunks, wcb + per_wg);
  const int per = (max(wce - wcb, 0) + nwaves - 1) / nwaves;
  const int cb = wcb + wave * per, ce = min(wce, cb + per);
  f32x4 acc[2][RN];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < RN; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* acol[2];
  const float* bcol[RN];
  float amask[2], bmask[RN];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = m0 + 16 * t + i;
    amask[t] = r < p.M ? 1.f : 0.f;
    acol[t] = p.A + (r < p.M ? r : 0);
  }
#pragma unroll
  for (int c = 0; c < RN; ++c) {
    const int n = n0 + 16 * c + i;
    bmask[c] = n < p.N ? 1.f : 0.f;
    bcol[c] = p.B + (n < p.N ? n : 0);
  }
  f32x4 a0[BATCH][2], b0[BATCH][RN], a1[BATCH][2], b1[BATCH][RN];
  auto load = [&](f32x4 (&a)[BATCH][2], f32x4 (&b)[BATCH][RN], int c0) {
    if (c0 >= ce) return;
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int k = (min(c0 + u, ce - 1) << 4) + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = min(k + e, p.K - 1);  // rows past K are clamped here and zeroed through kmask below
        const float kmask = (k + e < p.K) ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) a[u][t][e] = acol[t][(long)kk * p.lda] * kmask;
#pragma unroll
        for (int c = 0; c < RN; ++c) b[u][c][e] = bcol[c][(long)kk * p.ldb];
      }
    }
  };
  auto compute = [&](const f32x4 (&a)[BATCH][2], const f32x4 (&b)[BATCH][RN], int c0) {
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (c0 + u < ce) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float av = a[u][t][g] * amask[t];
#pragma unroll
            for (int c = 0; c < RN; ++c)
              acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u][c][g] * bmask[c], acc[t][c], 0, 0, 0);
          }
      }
  };
  load(a0, b0, cb);
  for (int c0 = cb; c0 < ce; c0 += 2 * BATCH) {
    load(a1, b1, c0 + BATCH);
    compute(a0, b0, c0);
    load(a0, b0, c0 + 2 * BATCH);
    compute(a1, b1, c0 + BATCH);
  }
  float* red = dred + (long)wave * (2 * RN * 256);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < RN; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(t * RN + c) * 256 + r * 64 + lane] = acc[t][c][r];
  __syncthreads();
  const bool split = gridDim.y > 1;
  for (int e = tid; e < 2 * RN * 256; e += blockDim.x) {
    const int tc = e >> 8, x = e & 255;
    const int t = tc / RN, c = tc % RN;
    const int r = x >> 6, l = x & 63;
    const int row = m0 + 16 * t + 4 * (l >> 4) + r, col = n0 + 16 * c + (l & 15);
    if (row < p.M && col < p.N) {
      float v = 0.f;
      for (int w = 0; w < nwaves; ++w) v += dred[(long)w * (2 * RN * 256) + e];
      float* o = p.C + (long)row * p.ldc + col;
      if (split) atomicAdd(o, v);
      else *o = p.accumulate ? (*o + v) : v;
    }
  }
}

static void launch_direct_tn(const GemmParams& p0, hipStream_t s) {
  constexpr int RN = 4, WAVES = 4;
  GemmParams p = p0;
  p.tiles_n = (p.N + 16 * RN - 1) / (16 * RN);
  p.tiles_m = (p.M + 31) / 32;
  const int tiles = p.tiles_m * p.tiles_n, chunks = (p.K + 15) / 16;
  int splits = 1;
  if (p.accumulate) {  // ~1k workgroups, every wave at least four 16-k chunks
    splits = (1024 + tiles - 1) / tiles;
    const int maxs = chunks / (4 * WAVES);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
  }
  const size_t sh = (size_t)WAVES * 2 * RN * 256 * sizeof(float);
  static const int env_batch = DV3_ENV_INT("DV3_DIRECT_BATCH", 0);
  if (env_batch == 2) hipLaunchKernelGGL((gemm_direct_tn_kernel<RN, 2>), dim3(tiles, splits), dim3(64 * WAVES), sh, s, p);
  else hipLaunchKernelGGL((gemm_direct_tn_kernel<RN, 1>), dim3(tiles, splits), dim3(64 * WAVES), sh, s, p);
}

// Few-row product against an n-contiguous B (data gradients: dX = dY * W, W [K,N] row-major), K split over
// workgroups, partial tiles added atomically (accumulate == 2 only).  The 16-column kernel above needs four
// 4-byte loads per lane per chunk here (a lane's B operand is one column); this one gives a wave 64 columns
// as four INTERLEAVED 16-column tiles -- tile j holds columns n0 + 4*i + j -- so that lane (i, q) reads
// B[k][n0+4i .. 4i+3] as one float4 and feeds its j-th element to tile j: per 16-k chunk 1 + 4 vector loads
// and 16 MFMAs, and the finished 16 x 64 block leaves as 256-byte coalesced atomic rows.
// Requires N % 64 == 0, no A2.  4 waves; K range = gridDim.y splits, then the waves.
template <int MT, int BATCH>
__global__ __launch_bounds__(256) void gemm_skinny_nn64_kernel(GemmParams p) {
  __shared__ float red[4][MT][4][257];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 64;
  const int chunks = (p.K + 15) >> 4;
  const int per_wg = (chunks + gridDim.y - 1) / gridDim.y;
  const int wcb = blockIdx.y * per_wg, wce = min(chunks, wcb + per_wg);
  const int per = (max(wce - wcb, 0) + 3) / 4;
  const int cb = wcb + wave * per, ce = min(wce, cb + per);
  f32x4 acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float amask[MT];
  const float* arow[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int row = i + 16 * t;
    const bool ok = row < p.M;
    amask[t] = ok ? 1.f : 0.f;
    arow[t] = p.A + (long)(ok ? row : 0) * p.lda;
  }
  const float* bptr = p.B + n0 + 4 * i;
  for (int c0 = cb; c0 < ce; c0 += BATCH) {
    f32x4 a[BATCH][MT], b[BATCH][4];
    const int clast = ce - 1;
    const bool fast = min(c0 + BATCH, ce) * 16 <= p.K;
    if (fast) {
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int k = (min(c0 + u, clast) << 4) + 4 * q;
#pragma unroll
        for (int t = 0; t < MT; ++t) a[u][t] = *reinterpret_cast<const f32x4u*>(arow[t] + k);
#pragma unroll
        for (int g = 0; g < 4; ++g) b[u][g] = *reinterpret_cast<const f32x4u*>(bptr + (long)(k + g) * p.ldb);
      }
    } else {  // ragged end of K
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int k = (min(c0 + u, clast) << 4) + 4 * q;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < p.K) v[e] = arow[t][k + e];
          a[u][t] = v;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 w = {0.f, 0.f, 0.f, 0.f};
          if (k + g < p.K) w = *reinterpret_cast<const f32x4u*>(bptr + (long)(k + g) * p.ldb);
          b[u][g] = w;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (c0 + u < ce) {  // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const float av = a[u][t][g] * amask[t];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u][g][j], acc[t][j], 0, 0, 0);
          }
      }
  }
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][t][j][r * 64 + lane] = acc[t][j][r];
  __syncthreads();
  // 16*MT rows x 64 columns; consecutive threads take consecutive columns of one row
  for (int e = tid; e < MT * 1024; e += 256) {
    const int t = e >> 10, row16 = (e >> 6) & 15, cidx = e & 63;
    const int row = 16 * t + row16;
    if (row < p.M) {
      const int x = (row16 & 3) * 64 + (row16 >> 2) * 16 + (cidx >> 2), j = cidx & 3;
      float v = red[0][t][j][x] + red[1][t][j][x] + red[2][t][j][x] + red[3][t][j][x];
      if (p.bias && blockIdx.y == 0) v += p.bias[n0 + cidx];
      atomicAdd(p.C + (long)row * p.ldc + n0 + cidx, v);
    }
  }
}

template <int MT>
static void launch_skinny_nn64(const GemmParams& p, hipStream_t s) {
  const int tiles = p.N / 64, chunks = (p.K + 15) / 16;
  int splits = (256 + tiles - 1) / tiles;
  const int maxs = chunks / 4 > 0 ? chunks / 4 : 1;
  if (splits > maxs) splits = maxs;
  const int per = ((chunks + splits - 1) / splits + 3) / 4;
  const dim3 grid(tiles, splits), block(256);
  if (per <= 2) hipLaunchKernelGGL((gemm_skinny_nn64_kernel<MT, 2>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_skinny_nn64_kernel<MT, 4>), grid, block, 0, s, p);
}

// Launch of the few-row kernel.  accumulate == 2 ("C += ..., summation order free") lets K be split over
// workgroups: enough workgroups of 4 waves to put ~1k waves on the chip while leaving every wave at least two
// 16-k chunks; the partial tiles are added with atomics.  accumulate == 1 keeps one workgroup per column tile
// (deterministic order).
template <bool TB, int MT>
static void launch_skinny(const GemmParams& p, int accumulate, hipStream_t s) {
  const int tiles = (p.N + 15) / 16, chunks = (p.K + 15) / 16;
  int splits = 1, waves = kSkinnyWaves;
  if (accumulate == 2) {
    waves = 4;
    splits = (1024 + tiles * waves - 1) / (tiles * waves);
    const int maxs = chunks / (2 * waves);
    if (splits > maxs) splits = maxs;
    if (splits <= 1) { splits = 1; waves = kSkinnyWaves; }
  }
  const int per_wg = (chunks + splits - 1) / splits;
  const int per = (per_wg + waves - 1) / waves;
  const dim3 grid(tiles, splits, (p.M + 16 * MT - 1) / (16 * MT)), block(64 * waves);
  if (per <= 3) hipLaunchKernelGGL((gemm_skinny_kernel<TB, MT, 3>), grid, block, 0, s, p);
  else if (per <= 6 || MT > 1) hipLaunchKernelGGL((gemm_skinny_kernel<TB, MT, 6>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_skinny_kernel<TB, MT, (MT > 1 ? 6 : 12)>), grid, block, 0, s, p);
}

template <class TS>
static hipError_t launch_ts(const GemmParams& p0, int transA, int transB, hipStream_t s) {
  GemmParams p = p0;
  p.tiles_m = (p.M + TS::BM - 1) / TS::BM;
  p.tiles_n = (p.N + TS::BN - 1) / TS::BN;
  // accumulating GEMMs with a long reduction and few output tiles (weight gradients: K = rows of the
  // batch) are split over K so that the grid covers the chip; partial sums are added atomically
  const int tiles = p.tiles_m * p.tiles_n;
  p.splits = 1;
  p.kchunk = p.K;
  if (p.accumulate && !p.A2 && tiles < 192 && p.K >= 8 * TS::BK) {
    int want = (384 + tiles - 1) / tiles;
    const int maxs = p.K / (4 * TS::BK);
    if (want > maxs) want = maxs;
    if (want > 1) {
      int chunk = (p.K + want - 1) / want;
      chunk = ((chunk + TS::BK - 1) / TS::BK) * TS::BK;
      p.kchunk = chunk;
      p.splits = (p.K + chunk - 1) / chunk;
    }
  }
  dim3 grid(tiles * p.splits), block(kThreads);
  if (!transA && transB) hipLaunchKernelGGL((gemm_kernel<TS, false, true>), grid, block, 0, s, p);
  else if (!transA && !transB) hipLaunchKernelGGL((gemm_kernel<TS, false, false>), grid, block, 0, s, p);
  else if (transA && !transB) hipLaunchKernelGGL((gemm_kernel<TS, true, false>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<TS, true, true>), grid, block, 0, s, p);
  return hipGetLastError();
}

using T128 = TileShape<2, 2, 2, 2, 16>;    // 128 x 128, BK 16
using T64 = TileShape<2, 2, 1, 1, 32>;     // 64 x 64,  BK 32
using T128K32 = TileShape<2, 2, 2, 2, 32>;  // 128 x 128, BK 32 (64 MFMAs per wave per barrier)
using T64K64 = TileShape<2, 2, 1, 1, 64>;   // 64 x 64,  BK 64 (32 MFMAs per wave per barrier)
using T32x64S = TileShape<1, 2, 1, 1, 64, 2>;  // 32 x 64, BK 64, K split over 2 wave-groups (small M*N)
using T32x32S4 = TileShape<1, 1, 1, 1, 64, 4>;  // 32 x 32, BK 64, K split over all 4 waves: 2 workgroups per CU at M*N = 512k
using T32x128 = TileShape<1, 4, 1, 1, 32>;  // 32 x 128, BK 32 (few rows: the observe scan, M = batch)

// Tile choice when the caller passes tile = -1 (the Python wrapper normally decides, same rule).
static int legacy_tile(int M, int N) {
  const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
  if (t64 <= 512) return 9;
  const long c128 = (((long)((M + 127) / 128) * ((N + 127) / 128)) + 255) / 256 * 4;
  const long c64 = (t64 + 255) / 256;
  return (c64 < c128) ? 1 : 4;
}
static int pick_tile(int M, int N, int K, int accumulate) {
  if (M <= 32) return 2;
  if (accumulate && K >= 4096 && (long)M * N >= 512L * 1024) return 4;
  const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
  if (t64 <= 128) return 9;
  // y = x W^T by output size: see dv3hip/ops.py pick_gemm_tile for the measurements behind the thresholds
  if ((long)((M + 127) / 128) * ((N + 127) / 128) >= 448) return 15;
  if (t64 < 512) return 14;
  if (t64 <= 1024) return 13;
  if (t64 <= 2048 && (N >= 3072 || N <= 512)) return 14;
  return legacy_tile(M, N);
}

}  // namespace dv3

using namespace dv3;

extern "C" int dv3_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda,
                            const float* A2, long lda2, int K1, const float* B, long ldb, float* C, long ldc,
                            const float* bias, int accumulate, int tile, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (K < 0 || !A || !B || !C) return DV3_ERR_ARG;
  if (tile == 7) {
    // narrow output (N <= 32, e.g. the actor's mean/std heads): C^T[N,M] = W[N,K] * A[M,K]^T on the skinny
    // kernel, weights as the few-row operand, transposed store.  Needs the plain y = x W^T + b form.
    if (transA || !transB || A2 || N > 32) return DV3_ERR_ARG;
    GemmParams q{};
    q.A = B; q.A2 = nullptr; q.B = A; q.C = C; q.bias = bias;
    q.M = N; q.N = M; q.K = K; q.K1 = K;
    q.lda = ldb; q.lda2 = 0; q.ldb = lda; q.ldc = ldc;
    q.accumulate = accumulate; q.transC = 1;
    hipStream_t s7 = (hipStream_t)stream;
    if (N <= 16) launch_skinny<true, 1>(q, accumulate, s7);
    else launch_skinny<true, 2>(q, accumulate, s7);
    return (int)hipGetLastError();
  }
  GemmParams p{};
  p.A = A; p.A2 = A2; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.K1 = (A2 ? K1 : K);
  p.lda = lda; p.lda2 = lda2; p.ldb = ldb; p.ldc = ldc;
  p.accumulate = accumulate;
  if (A2) {
    if (transA) return DV3_ERR_ARG;             // K-concat only for the k-contiguous orientation
    if (K1 <= 0 || K1 >= K || (K1 % 32) != 0) return DV3_ERR_ARG;  // segment edge on a K-tile boundary
  }
  // float4 path needs dword alignment only (gfx950 global loads); pointers from torch are >= 4B aligned.
  p.vecA = 1; p.vecB = 1;
  int t = ((tile >= 0 && tile <= 6) || (tile >= 8 && tile <= 17)) ? tile : pick_tile(M, N, K, accumulate);
  if (t >= 11 && tile < 0 && !l16_ok(p, transA, transB)) t = legacy_tile(M, N);
  if (t == 9 && tile < 0 && (transA || (A2 && (K1 % 16) != 0))) t = ((long)M * N <= 512L * 1024) ? 8 : 6;
  if (A2 && (K1 % 64) != 0 && (t == 5 || t == 6 || t == 8)) t = 1;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  if (t == 3) {
    // skinny path: M <= 128 (row blocks of 16 / 32 over grid.z), A k-contiguous; segment edge on a 16-k chunk boundary
    if (M > 128 || transA || (A2 && (K1 % 16) != 0)) return DV3_ERR_ARG;
    if (M <= 32 && accumulate == 2 && !transB && !A2 && (N % 64) == 0) {
      if (M <= 16) launch_skinny_nn64<1>(p, s);
      else launch_skinny_nn64<2>(p, s);
    } else if (M <= 16) {
      if (transB) launch_skinny<true, 1>(p, accumulate, s);
      else launch_skinny<false, 1>(p, accumulate, s);
    } else {
      if (transB) launch_skinny<true, 2>(p, accumulate, s);
      else launch_skinny<false, 2>(p, accumulate, s);
    }
    return (int)hipGetLastError();
  }
  if (t == 10) {
    // register-direct weight gradient: A [K][M], B [K][N]
    if (!transA || transB || A2 || bias) return DV3_ERR_ARG;
    launch_direct_tn(p, s);
    return (int)hipGetLastError();
  }
  if (t >= 11) {
    // k-contiguous LDS tiles, 16x16x4 MFMA: y = x W^T only, K and the segment edge on 32-k tile boundaries
    // (11: tile shape by size; 12 / 13 / 14 / 15: 64x96 / 64x64 / 32x64 / 128x128)
    if (!l16_ok(p, transA, transB)) return DV3_ERR_ARG;
    launch_l16(p, t - 11, s);
    return (int)hipGetLastError();
  }
  if (t == 9) {
    // register-direct kernel: A k-contiguous; segment edge on a 16-k chunk boundary
    if (transA || (A2 && (K1 % 16) != 0)) return DV3_ERR_ARG;
    if (transB) launch_direct<true>(p, s);
    else launch_direct<false>(p, s);
    return (int)hipGetLastError();
  }
  if (t == 0) e = launch_ts<T128>(p, transA, transB, s);
  else if (t == 4) e = launch_ts<T128K32>(p, transA, transB, s);
  else if (t == 5) e = launch_ts<T64K64>(p, transA, transB, s);
  else if (t == 6) e = launch_ts<T32x64S>(p, transA, transB, s);
  else if (t == 8) e = launch_ts<T32x32S4>(p, transA, transB, s);
  else if (t == 1) e = launch_ts<T64>(p, transA, transB, s);
  else e = launch_ts<T32x128>(p, transA, transB, s);
  return (int)e;
}

// y = [A|A2] W^T + bias as dv3_gemm_f32 (transA = 0, transB = 1, register-direct kernel) with the categorical
// sampling of the result fused into the epilogue: C [M, N] holds the logits of N/32 groups of 32 classes per row
// (RSSM._suff_stats_layer + get_dist + OneHotDist.sample, networks.py:241-250, 161-166, tools.py:452-460).
// Same draws, outputs and options as dv3_onehot_sample_fwd_ex on C.  N % 64 == 0.
extern "C" int dv3_gemm_sample_f32(int M, int N, int K, const float* A, long lda, const float* A2, long lda2, int K1,
                                   const float* B, long ldb, float* C, long ldc, const float* bias,
                                   const float* noise, const unsigned long long* rng_state,
                                   unsigned long long rng_offset, float* onehot, int* idx, const int* forced,
                                   unsigned int* flips, float unimix, int mode, const float* ln_gamma,
                                   const float* ln_beta, float* ln_mean, float* ln_rstd, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0 || !A || !B || !C || !onehot || (N % 64) != 0) return DV3_ERR_ARG;
  if (ln_gamma && (!ln_beta || A2 || (K % 4) != 0 || (lda % 4) != 0)) return DV3_ERR_ARG;
  if (!mode && !noise && !rng_state) return DV3_ERR_ARG;
  if (A2 && (K1 <= 0 || K1 >= K || (K1 % 16) != 0)) return DV3_ERR_ARG;
  GemmParams p{};
  p.A = A; p.A2 = A2; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.K1 = (A2 ? K1 : K);
  p.lda = lda; p.lda2 = lda2; p.ldb = ldb; p.ldc = ldc;
  p.accumulate = 0;
  p.vecA = 1; p.vecB = 1;
  p.smp_noise = noise; p.smp_rng = rng_state; p.smp_off = rng_offset; p.smp_onehot = onehot; p.smp_idx = idx;
  p.smp_forced = forced; p.smp_flips = flips; p.smp_unimix = unimix; p.smp_mode = mode;
  p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_mean = ln_mean; p.ln_rstd = ln_rstd;
  // same kernel family as dv3_gemm_f32 picks for the plain product of this size (bit-equal logits)
  if (!ln_gamma && pick_tile(M, N, K, 0) >= 11 && l16_ok(p, 0, 1)) {
    launch_l16_sample(p, (hipStream_t)stream);
    return (int)hipGetLastError();
  }
  const int chunks = (K + 15) / 16;
  launch_direct_rn<true, 4>(p, chunks >= 32 ? 8 : 4, (hipStream_t)stream);
  return (int)hipGetLastError();
}

// y = A W^T (dv3_gemm_f32 transA = 0, transB = 1, k-contiguous LDS tile kernel) with the output columns split over two
// destinations: columns [0, nsplit) -> C (ldc, accumulate), columns [nsplit, N) -> C2 (ldc2, accumulate2).
extern "C" int dv3_gemm_split_f32(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C,
                                  long ldc, int accumulate, float* C2, long ldc2, int nsplit, int accumulate2,
                                  void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0 || !A || !B || !C || !C2 || nsplit <= 0 || nsplit >= N || (nsplit % 16) != 0) return DV3_ERR_ARG;
  GemmParams p{};
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K; p.K1 = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.accumulate = accumulate ? 1 : 0;
  p.C2 = C2; p.ldc2 = ldc2; p.nsplit = nsplit; p.accumulate2 = accumulate2 ? 1 : 0;
  if (!l16_ok(p, 0, 1)) return DV3_ERR_ARG;
  launch_l16(p, 0, (hipStream_t)stream);
  return (int)hipGetLastError();
}

template <class TS>
static int launch_grouped(GroupParams& p, const int* M, const int* N, hipStream_t s) {
  int total = 0;
  for (int g = 0; g < p.n; ++g) {
    GroupItem& it = p.it[g];
    it.tiles_n = (N[g] + TS::BN - 1) / TS::BN;
    it.tile0 = total;
    total += ((M[g] + TS::BM - 1) / TS::BM) * it.tiles_n;
  }
  p.total = total;
  hipLaunchKernelGGL((gemm_tn_grouped_kernel<TS>), dim3(total), dim3(kThreads), 0, s, p);
  return (int)hipGetLastError();
}

extern "C" int dv3_gemm_tn_grouped_f32(int n, const float* const* A, const long* lda, const float* const* B,
                                       const long* ldb, float* const* C, const long* ldc, const int* M, const int* N,
                                       const int* K, const int* accumulate, void* stream) {
  if (n <= 0) return 0;
  if (n > kMaxGroup || !A || !lda || !B || !ldb || !C || !ldc || !M || !N || !K || !accumulate) return DV3_ERR_ARG;
  GroupParams p{};
  for (int g = 0; g < n; ++g) {
    if (M[g] <= 0 || N[g] <= 0 || K[g] <= 0 || !A[g] || !B[g] || !C[g]) return DV3_ERR_ARG;
    if (lda[g] < M[g] || ldb[g] < N[g] || ldc[g] < N[g] || lda[g] > 0x7fffffffL || ldb[g] > 0x7fffffffL ||
        ldc[g] > 0x7fffffffL)
      return DV3_ERR_ARG;
    GroupItem& it = p.it[p.n++];
    it.A = A[g]; it.B = B[g]; it.C = C[g];
    it.M = M[g]; it.N = N[g]; it.K = K[g];
    it.lda = (int)lda[g]; it.ldb = (int)ldb[g]; it.ldc = (int)ldc[g];
    it.accumulate = accumulate[g] ? 1 : 0;
  }
  // 64 x 64 x 32 tiles.  (128 x 128 x 16, the tile of the long single reductions, measured on the world model's two
  // clusters, DV3_GROUP_TILE=128 in the development library: 420 against 379 us and 321 against 313 on a 128-CU lane.)
  if (DV3_ENV_INT("DV3_GROUP_TILE", 64) == 128) return launch_grouped<TileShape<2, 2, 2, 2, 16>>(p, M, N, (hipStream_t)stream);
  return launch_grouped<TileShape<2, 2, 1, 1, 32>>(p, M, N, (hipStream_t)stream);
}
