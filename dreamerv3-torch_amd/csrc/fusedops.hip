// Row-fused kernels of the imagination step (models.ImagBehavior._imagine, models.py:448-548, and the layers
// it drives: networks.RSSM.img_step networks.py:208-233, networks.MLP.forward networks.py:657-681).
//
// 1. onehot_linear_ln: a Linear whose input is [stoch | dense tail] (the reference's
//    torch.cat([stoch.flat, action]) at networks.py:216, get_feat at networks.py:154-159) with the stoch part an
//    EXACT one-hot per categorical group (tools.py:452-460: the forward value of OneHotDist.sample is
//    one_hot(argmax) + (p - p.detach()) = a one-hot).  x @ W^T over the stoch columns is therefore the sum of S
//    columns of W -- S rows of the transposed weight -- picked by the class indices: S*N adds per row instead of
//    S*D*N multiply-adds (1/32 of the GEMM at D = 32), an L2-resident gather instead of an MFMA launch.  A wave owns
//    a whole output row, so the LayerNorm + SiLU that follows the Linear (networks.py:55-56, 631-633) is fused:
//    one launch replaces GEMM + LN.  The dense part of a wider input (deter for the MLP heads, whose first layer
//    reads feat = [stoch | deter]) comes in as `base` from a K = deter GEMM.
// 2. actor_head: LayerNorm + SiLU of the actor trunk's last layer, its mean/std (or logit) heads, the action
//    sample and the entropy in one row-wise launch (networks.py:672-681, 693-700, 713-714, tools.py:594-598)
//    instead of LN, two narrow GEMMs, the noise fill and the sampling kernel.
// 3. helpers: 2-D transpose (weights -> [K][N] for the gather), one-hot -> class index.
//
// HBM/L2-bound row kernels: wave per row, 16 B per lane, statistics two-pass in registers (as rowops.hip).
#include "dv3_common.h"

namespace dv3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct OLParams {
  const int* idx;  // [M][S] class index of every categorical group
  int S, D;
  const float* x2;  // dense tail inputs [M][A2] (action) or null
  long ldx2;
  int A2;
  const float* WT;  // transposed weight [S*D + A2][N]
  long ldw;
  const float* base;  // optional [M][N] added first (product of the remaining input columns)
  long ldbase;
  float* pre;  // pre-activation out [M][N] (may alias base)
  long ldpre;
  const float* gamma;
  const float* beta;
  float* y;  // SiLU(LN(pre)) out, or null: only pre is produced
  long ldy;
  float* mean;
  float* rstd;
  long M;
  int N;
  int act;
  // SMP variant (observe scan, D == 32): the row's S categorical groups are SAMPLED here first from smp_logit
  // [M][S][32] (dv3_onehot_sample_fwd_ex semantics incl. the next step's reset blend), and the gather reads the
  // blended class indices from LDS: the posterior sample of step t and the img_in layer of step t+1 in one launch.
  const float* smp_logit;
  const float* smp_noise;
  const unsigned long long* smp_rng;
  unsigned long long smp_off;
  float* smp_onehot;
  int* smp_idx;
  const int* smp_forced;
  unsigned int* smp_flips;
  float smp_unimix;
  int smp_mode;
  const float* nb_first;  // [M] is_first of the NEXT step
  const float* nb_init;   // [S][32] initial stoch (a one-hot: the mode of the initial prior)
  const int* nb_init_idx;  // [S]
  float* nb_out;          // [M][S][32] blended one-hot (next step's stoch input)
  int* nb_idx_out;        // [M][S] its class indices (= what the gather uses)
};

// N == 256 * NV4.  One workgroup (4 waves) per output row: wave w gathers the rows s = w, w+4, ... of its share in
// ONE batch of independent 16-byte loads (lane l owns columns 4*(l + 64 v) .. +3), the four partial rows meet in
// LDS, and every wave then sums them in the same fixed order and normalises -- one memory round trip and one
// barrier per row (a wave per row needed S / 8 dependent round trips at one wave per SIMD).
template <int NV4, bool SMP = false>
__global__ __launch_bounds__(SMP ? 1024 : 256) void onehot_linear_ln_vec_kernel(OLParams p) {
  __shared__ __attribute__((aligned(16))) float part[4][256 * NV4];
  __shared__ int sidx[64];
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const float inv_n = 1.f / (float)p.N;
  constexpr int MAXB = 8;  // gathered rows per wave and batch (S <= 32: one batch)
  for (long r = blockIdx.x; r < p.M; r += gridDim.x) {
    int my;
    if constexpr (SMP) {
      // one group of 32 classes per half-wave (16 waves); same arithmetic and draws as onehot_sample_kernel<32>.
      // Waves 4..15 only sample: the gather below is the 4-wave scheme of the plain kernel (they keep its barriers).
      const int d = threadIdx.x & 31;
      unsigned long long seed = 0, offset = 0;
      if (!p.smp_mode && !p.smp_noise) {
        seed = p.smp_rng[0];
        offset = p.smp_rng[1] + p.smp_off;
      }
      const float m = p.nb_first[r];
      __syncthreads();  // the previous row's gather is done with sidx
      for (int sg = threadIdx.x >> 5; sg < p.S; sg += 32) {  // 1024 threads: all S <= 32 groups in one pass
        const long gi = r * p.S + sg;
        const float lg = p.smp_logit[gi * 32 + d];
        float sm, ph;
        unimix_probs<32>(lg, true, 32, p.smp_unimix, sm, ph);
        float score = ph;
        if (!p.smp_mode) {
          float q;
          if (p.smp_noise) {
            q = p.smp_noise[gi * 32 + d];
          } else {
            uint32_t o4[4];
            const unsigned long long e = (unsigned long long)gi * 32 + d;
            Philox ph4(seed);
            ph4(offset + (e >> 2), 0x5eedULL, o4);
            q = fmaxf(-logf(u01(o4[e & 3])), 1e-30f);
          }
          score = ph / q;
        }
        float best = score;
        int bi = d;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          const float ob = __shfl_xor(best, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
          }
        }
        if (p.smp_forced) {
          const int f = p.smp_forced[gi];
          if (d == 0 && p.smp_flips && f != bi) atomicAdd(p.smp_flips, 1u);
          bi = f;
        }
        const float v = (d == bi) ? 1.f : 0.f;
        p.smp_onehot[gi * 32 + d] = v;
        p.nb_out[gi * 32 + d] = v * (1.f - m) + p.nb_init[sg * 32 + d] * m;
        if (d == 0) {
          if (p.smp_idx) p.smp_idx[gi] = bi;
          const int ni = (m != 0.f) ? p.nb_init_idx[sg] : bi;
          p.nb_idx_out[gi] = ni;
          sidx[sg] = ni;
        }
      }
      __syncthreads();
      my = (l < p.S) ? sidx[l] : 0;
    } else {
      my = (l < p.S) ? p.idx[r * p.S + l] : 0;
    }
    const bool worker = !SMP || wave < 4;  // wave-uniform
    f32x4 acc[NV4];
#pragma unroll
    for (int v = 0; v < NV4; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (worker) {
    if (wave == 0 && p.base) {
#pragma unroll
      for (int v = 0; v < NV4; ++v) acc[v] = *reinterpret_cast<const f32x4u*>(p.base + r * p.ldbase + 4 * (l + 64 * v));
    }
    for (int s0 = wave; s0 < p.S; s0 += 4 * MAXB) {
      f32x4 t[MAXB][NV4];
#pragma unroll
      for (int u = 0; u < MAXB; ++u) {
        const int s = s0 + 4 * u;
        if (s < p.S) {  // wave-uniform
          const int id = __builtin_amdgcn_readlane(my, s);
          const float* w = p.WT + (long)(s * p.D + id) * p.ldw + 4 * l;
#pragma unroll
          for (int v = 0; v < NV4; ++v) t[u][v] = *reinterpret_cast<const f32x4u*>(w + 256 * v);
        } else {
#pragma unroll
          for (int v = 0; v < NV4; ++v) t[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int u = 0; u < MAXB; ++u)
#pragma unroll
        for (int v = 0; v < NV4; ++v) acc[v] += t[u][v];
    }
    for (int a = wave; a < p.A2; a += 4) {
      const float xa = p.x2[r * p.ldx2 + a];
      const float* w = p.WT + (long)(p.S * p.D + a) * p.ldw + 4 * l;
#pragma unroll
      for (int v = 0; v < NV4; ++v) acc[v] += xa * *reinterpret_cast<const f32x4u*>(w + 256 * v);
    }
    }
    __syncthreads();  // the previous row's readers are done with `part`
    if (worker) {
#pragma unroll
      for (int v = 0; v < NV4; ++v) *reinterpret_cast<f32x4*>(&part[wave][4 * (l + 64 * v)]) = acc[v];
    }
    __syncthreads();
    if (!worker) continue;  // no barrier below
#pragma unroll
    for (int v = 0; v < NV4; ++v) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&part[0][4 * (l + 64 * v)]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&part[1][4 * (l + 64 * v)]);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(&part[2][4 * (l + 64 * v)]);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(&part[3][4 * (l + 64 * v)]);
      acc[v] = (a0 + a1) + (a2 + a3);
      if ((v & 3) == wave) *reinterpret_cast<f32x4u*>(p.pre + r * p.ldpre + 4 * (l + 64 * v)) = acc[v];
    }
    if (!p.y) continue;
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV4; ++v) s += (acc[v][0] + acc[v][1]) + (acc[v][2] + acc[v][3]);
    const float mean = group_sum<64>(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV4; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = acc[v][e] - mean;
        q += d * d;
      }
    const float rstd = rsqrtf(group_sum<64>(q) * inv_n + kLnEps);
#pragma unroll
    for (int v = 0; v < NV4; ++v) {
      if ((v & 3) != wave) continue;  // wave-uniform
      const f32x4 g = *reinterpret_cast<const f32x4u*>(p.gamma + 4 * (l + 64 * v));
      const f32x4 b = *reinterpret_cast<const f32x4u*>(p.beta + 4 * (l + 64 * v));
      f32x4 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = (acc[v][e] - mean) * rstd * g[e] + b[e];
        z[e] = p.act ? siluf_(t) : t;
      }
      *reinterpret_cast<f32x4u*>(p.y + r * p.ldy + 4 * (l + 64 * v)) = z;
    }
    if (threadIdx.x == 0) {
      if (p.mean) p.mean[r] = mean;
      if (p.rstd) p.rstd[r] = rstd;
    }
  }
}

// any N (the tiny test configs, odd widths): a workgroup per row, columns strided over its 256 threads; the
// pre-activation makes its passes through global memory (every thread re-reads only what it wrote)
__global__ __launch_bounds__(256) void onehot_linear_ln_generic_kernel(OLParams p) {
  __shared__ float red[4];
  __shared__ int sidx[64];
  const int tid = threadIdx.x;
  const float inv_n = 1.f / (float)p.N;
  for (long r = blockIdx.x; r < p.M; r += gridDim.x) {
    __syncthreads();
    if (tid < p.S) sidx[tid] = p.idx[r * p.S + tid];
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < p.N; c += 256) {
      float v = p.base ? p.base[r * p.ldbase + c] : 0.f;
      for (int k = 0; k < p.S; ++k) v += p.WT[(long)(k * p.D + sidx[k]) * p.ldw + c];
      for (int a = 0; a < p.A2; ++a) v += p.x2[r * p.ldx2 + a] * p.WT[(long)(p.S * p.D + a) * p.ldw + c];
      p.pre[r * p.ldpre + c] = v;
      s += v;
    }
    if (!p.y) continue;  // uniform
    const float mean = block_sum_256(s, red) * inv_n;
    float q = 0.f;
    for (int c = tid; c < p.N; c += 256) {
      const float d = p.pre[r * p.ldpre + c] - mean;
      q += d * d;
    }
    const float rstd = rsqrtf(block_sum_256(q, red) * inv_n + kLnEps);
    for (int c = tid; c < p.N; c += 256) {
      const float t = (p.pre[r * p.ldpre + c] - mean) * rstd * p.gamma[c] + p.beta[c];
      p.y[r * p.ldy + c] = p.act ? siluf_(t) : t;
    }
    if (tid == 0) {
      if (p.mean) p.mean[r] = mean;
      if (p.rstd) p.rstd[r] = rstd;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
struct AHParams {
  const float* pre;  // [M][U] pre-activation of the trunk's last Linear
  long ldpre;
  const float* gamma;
  const float* beta;
  float* y;  // SiLU(LN(pre)) out [M][U] (saved: the heads' weight gradients read it)
  long ldy;
  float* mean;
  float* rstd;
  const float* Wm;  // [A][U]
  const float* bm;
  const float* Ws;  // [A][U] std head (continuous actor) or null
  const float* bs;
  float* out_m;  // raw head outputs [M][A] (saved for the backward)
  float* out_s;
  const float* noise;  // [M][A]: N(0,1) (continuous) / Exp(1) (one-hot); null -> Philox
  const unsigned long long* rng;
  unsigned long long rng_off;
  float* eps_out;  // continuous: the noise used [M][A] (saved for the backward); may alias noise
  float* action;   // [M][A]
  float* ent;      // [M]
  int* act_idx;    // one-hot actor: chosen class [M] (optional)
  const int* forced;
  unsigned int* flips;
  long M;
  int U, A;
  float min_std, max_std, unimix;
  int onehot;
};

// The heads are 2A (or A) dot products of length U per row.  Every wave of the chip walks the same weight rows, so
// the loads of a batch of OB outputs are ALL issued before the first multiply (one memory round trip per batch, not
// one per output), and wave w starts its batch at output w % OB so that the 1024 waves do not hit the same cache
// lines at the same moment (from global memory in output order this took 26-36 us for 1024 rows).
template <int NV>
__global__ __launch_bounds__(256) void actor_head_kernel(AHParams p) {
  __shared__ float hsh[4 * 16 * 65];
  __shared__ float hres[4 * 128];
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const float inv_n = 1.f / (float)p.U;
  float g[NV], b[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = l + 64 * v;
    g[v] = c < p.U ? p.gamma[c] : 0.f;
    b[v] = c < p.U ? p.beta[c] : 0.f;
  }
  unsigned long long seed = 0, offset = 0;
  if (!p.noise) {
    seed = p.rng[0];
    offset = p.rng[1] + p.rng_off;
  }
  const int nout = p.Ws ? 2 * p.A : p.A;
  constexpr int OB = (NV <= 8) ? 12 : 6;  // OB * NV weight registers per batch of outputs
  const int rot = (int)((blockIdx.x * 4 + wave) % OB);
  auto load_w = [&](float (&wv)[OB][NV], int o0) {
    const int no = min(OB, nout - o0);
#pragma unroll
    for (int jj = 0; jj < OB; ++jj) {
      const int j = (jj + rot) % OB;
      const int o = o0 + j;
      if (j < no) {  // wave-uniform
        const float* w = (o < p.A) ? p.Wm + (long)o * p.U : p.Ws + (long)(o - p.A) * p.U;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int c = l + 64 * v;
          wv[jj][v] = c < p.U ? w[c] : 0.f;
        }
      } else {
#pragma unroll
        for (int v = 0; v < NV; ++v) wv[jj][v] = 0.f;
      }
    }
  };
  for (long r = (long)blockIdx.x * 4 + wave; r < p.M; r += (long)gridDim.x * 4) {
    float x[NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = l + 64 * v;
      x[v] = c < p.U ? p.pre[r * p.ldpre + c] : 0.f;
      s += x[v];
    }
    const float mean = group_sum<64>(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float d = (l + 64 * v < p.U) ? x[v] - mean : 0.f;
      q += d * d;
    }
    const float rstd = rsqrtf(group_sum<64>(q) * inv_n + kLnEps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = l + 64 * v;
      x[v] = c < p.U ? siluf_((x[v] - mean) * rstd * g[v] + b[v]) : 0.f;
      if (c < p.U) p.y[r * p.ldy + c] = x[v];
    }
    if (l == 0) {
      if (p.mean) p.mean[r] = mean;
      if (p.rstd) p.rstd[r] = rstd;
    }
    // heads: 2A (or A) dot products of length U.  Every lane writes its partial of each output to LDS and lane j
    // then sums the 64 partials of output j: two LDS passes instead of 2A dependent wave reductions.  Outputs go
    // through in chunks of 16; `res` collects them so that lane a ends up with mean[a] and std[a].
    float mr = 0.f, sr = 0.f;
    {
      float* sh = hsh + wave * (16 * 65);
      float* res = hres + wave * 128;
      for (int o0 = 0; o0 < nout; o0 += OB) {
        const int no = min(OB, nout - o0);
        // (hoisting the first batch above the row's own loads and LayerNorm was measured: 96 live registers across
        // the whole kernel, 15.6 -> 33 us)
        float wn[OB][NV];
        load_w(wn, o0);
#pragma unroll
        for (int jj = 0; jj < OB; ++jj) {
          const int j = (jj + rot) % OB;
          float d = 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v) d += x[v] * wn[jj][v];
          if (j < no) sh[j * 65 + l] = d;
        }
        __builtin_amdgcn_wave_barrier();
        if (l < no) {
          float t = 0.f;
#pragma unroll 16
          for (int k = 0; k < 64; ++k) t += sh[l * 65 + k];
          res[o0 + l] = t;
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (l < p.A) {
        mr = res[l] + p.bm[l];
        if (p.Ws) sr = res[p.A + l] + p.bs[l];
      }
      __builtin_amdgcn_wave_barrier();
    }
    const bool valid = l < p.A;
    const long e = r * p.A + l;
    if (valid) {
      p.out_m[e] = mr;
      if (p.out_s) p.out_s[e] = sr;
    }
    if (!p.onehot) {
      // Normal(tanh(mean), (max-min)*sigmoid(std+2)+min), rsample, absmax-1 rescale (tools.py:594-598)
      float en = 0.f;
      if (valid) {
        float eps;
        if (p.noise) {
          eps = p.noise[e];
        } else {  // the element fill_normal_kernel would have produced: Box-Muller over Philox counter e >> 2
          uint32_t o[4];
          Philox ph(seed);
          ph(offset + ((unsigned long long)e >> 2), 0x6e6f726dULL, o);
          const int j = (int)(e & 3);
          const float rad = sqrtf(-2.f * logf(u01(o[j & 2])));
          const float ang = 6.283185307179586f * u01(o[(j & 2) + 1]);
          eps = (j & 1) ? rad * sinf(ang) : rad * cosf(ang);
        }
        if (p.eps_out) p.eps_out[e] = eps;
        const float mu = tanhf(mr);
        const float sd = (p.max_std - p.min_std) * sigmoidf_(sr + 2.f) + p.min_std;
        const float pre = mu + sd * eps;
        p.action[e] = pre * (1.f / fmaxf(fabsf(pre), 1.f));
        en = 0.5f + 0.9189385332046727f + logf(sd);
      }
      en = group_sum<64>(en);
      if (l == 0 && p.ent) p.ent[r] = en;
    } else {
      // OneHotDist(logits, unimix): sample = onehot(argmax p_hat / q), entropy of p_hat (tools.py:436-460)
      const float m = group_max<64>(valid ? mr : -INFINITY);
      const float ex = valid ? expf(mr - m) : 0.f;
      const float sm = ex / group_sum<64>(ex);
      const float ph = valid ? sm * (1.f - p.unimix) + p.unimix / (float)p.A : 0.f;
      float qv = 1.f;
      if (valid) {
        if (p.noise) {
          qv = p.noise[e];
        } else {
          uint32_t o[4];
          Philox ph4(seed);
          ph4(offset + ((unsigned long long)e >> 2), 0x5eedULL, o);
          qv = fmaxf(-logf(u01(o[e & 3])), 1e-30f);
        }
      }
      float best = valid ? ph / qv : -INFINITY;
      int bi = l;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) {
          best = ob;
          bi = oi;
        }
      }
      if (p.forced) {
        const int f = p.forced[r];
        if (l == 0 && p.flips && f != bi) atomicAdd(p.flips, 1u);
        bi = f;
      }
      if (valid) p.action[e] = (l == bi) ? 1.f : 0.f;
      if (l == 0 && p.act_idx) p.act_idx[r] = bi;
      const float en = group_sum<64>(valid ? -ph * logf(ph) : 0.f);
      if (l == 0 && p.ent) p.ent[r] = en;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c]   (weights [N][K] -> [K][N] for the gather; 32x32 tiles through LDS)
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, long lds_, int R, int C,
                                                          float* __restrict__ dst, long ldd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int r = r0 + ty + j, c = c0 + tx;
    tile[ty + j][tx] = (r < R && c < C) ? src[(long)r * lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int c = c0 + ty + j, r = r0 + tx;
    if (c < C && r < R) dst[(long)c * ldd + r] = tile[tx][ty + j];
  }
}

// Up to kTrJobs transposes in ONE launch: the per-update re-packing of the weights the gathers and the reverse rollout
// read (engine.MLPEngine.pack_onehot, RSSMEngine.pack_img_in / pack_bwd) is 9-10 small launches otherwise.
constexpr int kTrJobs = 12;
struct TrJobs {
  const float* src[kTrJobs];
  float* dst[kTrJobs];
  long lds[kTrJobs], ldd[kTrJobs];
  int R[kTrJobs], C[kTrJobs], tiles_c[kTrJobs];
  int tile0[kTrJobs + 1];  // first workgroup of job k; tile0[n] = grid size
  int n;
};
__global__ __launch_bounds__(256) void transpose2d_many_kernel(TrJobs jb) {
  __shared__ float tile[32][33];
  int k = 0;
  while (k + 1 < jb.n && (int)blockIdx.x >= jb.tile0[k + 1]) ++k;
  const int t = blockIdx.x - jb.tile0[k];
  const float* __restrict__ src = jb.src[k];
  float* __restrict__ dst = jb.dst[k];
  const long lds_ = jb.lds[k], ldd = jb.ldd[k];
  const int R = jb.R[k], C = jb.C[k];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = (t % jb.tiles_c[k]) * 32, r0 = (t / jb.tiles_c[k]) * 32;
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int r = r0 + ty + j, c = c0 + tx;
    tile[ty + j][tx] = (r < R && c < C) ? src[(long)r * lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int c = c0 + ty + j, r = r0 + tx;
    if (c < C && r < R) dst[(long)c * ldd + r] = tile[tx][ty + j];
  }
}

// class index of every one-hot group: idx[g] = argmax_d x[g][d] (lowest index on ties).  G = pow2 >= D lanes per
// group, one class per lane (coalesced), shuffle argmax.
template <int G>
__global__ __launch_bounds__(256) void onehot_to_idx_kernel(const float* __restrict__ x, int* __restrict__ idx, long R,
                                                            int D) {
  constexpr int GPB = 256 / G;
  const int sub = threadIdx.x / G, d = threadIdx.x % G;
  for (long g0 = (long)blockIdx.x * GPB; g0 < R; g0 += (long)gridDim.x * GPB) {
    const long g = g0 + sub;
    float best = (g < R && d < D) ? x[g * D + d] : -INFINITY;
    int bi = d;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (g < R && d == 0) idx[g] = bi;
  }
}

static unsigned cap_grid(long n, long per_block, long cap) {
  long b = (n + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace dv3

using namespace dv3;

extern "C" int dv3_onehot_linear_ln_fwd(const int* idx, int S, int D, const float* x2, long ldx2, int A2,
                                        const float* WT, long ldw, const float* base, long ldbase, float* pre,
                                        long ldpre, const float* gamma, const float* beta, float* y, long ldy,
                                        float* mean, float* rstd, long M, int N, int act, void* stream) {
  if (M <= 0) return 0;
  if (!idx || !WT || !pre || S <= 0 || S > 64 || D <= 0 || N <= 0 || A2 < 0 || (A2 > 0 && !x2) || ldw < N ||
      ldpre < N || (base && ldbase < N))
    return DV3_ERR_ARG;
  if (y && (!gamma || !beta || ldy < N)) return DV3_ERR_ARG;
  OLParams p{idx, S, D, x2, ldx2, A2, WT, ldw, base, ldbase, pre, ldpre, gamma, beta, y, ldy, mean, rstd, M, N, act};
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (N % 256 == 0) && N <= 1024 && (ldw % 4 == 0) && (ldpre % 4 == 0) && (!base || ldbase % 4 == 0) &&
                   (!y || ldy % 4 == 0);
  if (vec) {
    const dim3 grid(cap_grid(M, 1, 32768)), block(256);
    switch (N / 256) {
      case 1: hipLaunchKernelGGL((onehot_linear_ln_vec_kernel<1>), grid, block, 0, s, p); break;
      case 2: hipLaunchKernelGGL((onehot_linear_ln_vec_kernel<2>), grid, block, 0, s, p); break;
      case 3: hipLaunchKernelGGL((onehot_linear_ln_vec_kernel<3>), grid, block, 0, s, p); break;
      default: hipLaunchKernelGGL((onehot_linear_ln_vec_kernel<4>), grid, block, 0, s, p); break;
    }
  } else {
    hipLaunchKernelGGL(onehot_linear_ln_generic_kernel, dim3(cap_grid(M, 1, 16384)), dim3(256), 0, s, p);
  }
  return (int)hipGetLastError();
}

// dv3_onehot_sample_fwd_ex (with the next step's reset blend) on logit [M][S][32] followed by dv3_onehot_linear_ln_fwd
// on the blended class indices, in one launch (row per workgroup): the last launch of observe step t and the first of
// step t+1.  N % 256 == 0, N <= 1024, S <= 32.
extern "C" int dv3_onehot_sample_linear_ln_fwd(const float* logit, const float* noise,
                                               const unsigned long long* rng_state, unsigned long long rng_offset,
                                               float* onehot, int* idx, const int* forced, unsigned int* flips,
                                               float unimix, int mode, const float* next_first, const float* init,
                                               const int* init_idx, float* next_out, int* next_idx, int S,
                                               const float* x2, long ldx2, int A2, const float* WT, long ldw,
                                               float* pre, long ldpre, const float* gamma, const float* beta, float* y,
                                               long ldy, float* mean, float* rstd, long M, int N, int act,
                                               void* stream) {
  if (M <= 0) return 0;
  if (!logit || !onehot || !next_first || !init || !init_idx || !next_out || !next_idx || !WT || !pre || !gamma ||
      !beta || !y || S <= 0 || S > 32 || N <= 0 || (N % 256) != 0 || N > 1024 || A2 < 0 || (A2 > 0 && !x2) ||
      ldw < N || ldpre < N || ldy < N || (ldw % 4) || (ldpre % 4) || (ldy % 4))
    return DV3_ERR_ARG;
  if (!mode && !noise && !rng_state) return DV3_ERR_ARG;
  OLParams p{nullptr, S, 32, x2, ldx2, A2, WT, ldw, nullptr, 0, pre, ldpre, gamma, beta, y, ldy, mean, rstd, M, N, act,
             logit, noise, rng_state, rng_offset, onehot, idx, forced, flips, unimix, mode, next_first, init, init_idx,
             next_out, next_idx};
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(cap_grid(M, 1, 32768)), block(1024);
  switch (N / 256) {
    case 1: hipLaunchKernelGGL((onehot_linear_ln_vec_kernel<1, true>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((onehot_linear_ln_vec_kernel<2, true>), grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL((onehot_linear_ln_vec_kernel<3, true>), grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((onehot_linear_ln_vec_kernel<4, true>), grid, block, 0, s, p); break;
  }
  return (int)hipGetLastError();
}

extern "C" int dv3_actor_head_fwd(const float* pre, long ldpre, const float* gamma, const float* beta, float* y,
                                  long ldy, float* mean, float* rstd, const float* Wm, const float* bm,
                                  const float* Ws, const float* bs, float* out_m, float* out_s, const float* noise,
                                  const unsigned long long* rng_state, unsigned long long rng_offset, float* eps_out,
                                  float* action, float* entropy, int* act_idx, const int* forced,
                                  unsigned int* flips, long M, int U, int A, float min_std, float max_std,
                                  float unimix, int onehot, void* stream) {
  if (M <= 0) return 0;
  if (!pre || !gamma || !beta || !y || !Wm || !bm || !out_m || !action || U <= 0 || U > 1024 || A <= 0 || A > 64 ||
      ldpre < U || ldy < U)
    return DV3_ERR_ARG;
  if (!onehot && (!Ws || !bs || !out_s)) return DV3_ERR_ARG;
  if (onehot && Ws) return DV3_ERR_ARG;
  if (!noise && !rng_state) return DV3_ERR_ARG;
  AHParams p{pre, ldpre, gamma, beta, y, ldy, mean, rstd, Wm, bm, Ws, bs, out_m, out_s, noise, rng_state, rng_offset,
             eps_out, action, entropy, act_idx, forced, flips, M, U, A, min_std, max_std, unimix, onehot};
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(cap_grid(M, 4, 4096)), block(256);
  if (U <= 64) hipLaunchKernelGGL((actor_head_kernel<1>), grid, block, 0, s, p);
  else if (U <= 256) hipLaunchKernelGGL((actor_head_kernel<4>), grid, block, 0, s, p);
  else if (U <= 512) hipLaunchKernelGGL((actor_head_kernel<8>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((actor_head_kernel<16>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

extern "C" int dv3_transpose2d(const float* src, long lds, int R, int C, float* dst, long ldd, void* stream) {
  if (R <= 0 || C <= 0) return 0;
  if (!src || !dst || lds < C || ldd < R) return DV3_ERR_ARG;
  hipLaunchKernelGGL(transpose2d_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, src,
                     lds, R, C, dst, ldd);
  return (int)hipGetLastError();
}

extern "C" int dv3_transpose2d_many(int njobs, const unsigned long long* jobs_host, void* stream) {
  // jobs_host: HOST array of njobs x 6 values {src, dst, lds, ldd, R, C} (device pointers as integers): consumed
  // before the call returns (the descriptors travel as kernel arguments)
  if (njobs <= 0) return 0;
  if (!jobs_host || njobs > kTrJobs) return DV3_ERR_ARG;
  TrJobs jb;
  int tiles = 0;
  for (int k = 0; k < njobs; ++k) {
    const unsigned long long* j = jobs_host + 6 * k;
    jb.src[k] = (const float*)(uintptr_t)j[0];
    jb.dst[k] = (float*)(uintptr_t)j[1];
    jb.lds[k] = (long)j[2];
    jb.ldd[k] = (long)j[3];
    jb.R[k] = (int)j[4];
    jb.C[k] = (int)j[5];
    if (!jb.src[k] || !jb.dst[k] || jb.R[k] <= 0 || jb.C[k] <= 0 || jb.lds[k] < jb.C[k] || jb.ldd[k] < jb.R[k]) return DV3_ERR_ARG;
    jb.tiles_c[k] = (jb.C[k] + 31) / 32;
    jb.tile0[k] = tiles;
    tiles += jb.tiles_c[k] * ((jb.R[k] + 31) / 32);
  }
  for (int k = njobs; k <= kTrJobs; ++k) jb.tile0[k] = tiles;
  jb.n = njobs;
  hipLaunchKernelGGL(transpose2d_many_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, jb);
  return (int)hipGetLastError();
}

extern "C" int dv3_onehot_to_idx(const float* onehot, int* idx, long R, int D, void* stream) {
  if (R <= 0) return 0;
  if (!onehot || !idx || D <= 0) return DV3_ERR_ARG;
  if (D > 64) return DV3_ERR_ARG;
  int G = 4;
  while (G < 64 && G < D) G <<= 1;
  hipStream_t s = (hipStream_t)stream;
  switch (G) {
    case 4: hipLaunchKernelGGL((onehot_to_idx_kernel<4>), dim3(cap_grid(R, 64, 8192)), dim3(256), 0, s, onehot, idx, R, D); break;
    case 8: hipLaunchKernelGGL((onehot_to_idx_kernel<8>), dim3(cap_grid(R, 32, 8192)), dim3(256), 0, s, onehot, idx, R, D); break;
    case 16: hipLaunchKernelGGL((onehot_to_idx_kernel<16>), dim3(cap_grid(R, 16, 8192)), dim3(256), 0, s, onehot, idx, R, D); break;
    case 32: hipLaunchKernelGGL((onehot_to_idx_kernel<32>), dim3(cap_grid(R, 8, 8192)), dim3(256), 0, s, onehot, idx, R, D); break;
    default: hipLaunchKernelGGL((onehot_to_idx_kernel<64>), dim3(cap_grid(R, 4, 8192)), dim3(256), 0, s, onehot, idx, R, D); break;
  }
  return (int)hipGetLastError();
}
