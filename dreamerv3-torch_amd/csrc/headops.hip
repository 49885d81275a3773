// Output heads and losses of the hot path (all HBM-bound row kernels).
//
// Reference: tools.DiscDist (tools.py:463-517) for the reward / value heads, tools.Bernoulli
// (tools.py:604-628) for the continue head, tools.MSEDist (tools.py:520-540) + the u8 -> f32/255
// image decode of WorldModel.preprocess (models.py:176-180), tools.ContDist around a Normal for the
// continuous actor (networks.py:693-700, tools.py:575-601), tools.lambda_return (tools.py:682-728)
// and the discount weights of ImagBehavior._compute_target (models.py:620-638).
#include "dv3_common.h"

namespace dv3 {

constexpr int kBuckets = 255;

// torch.linspace(-20, 20, 255) element k, as ATen computes it (symmetric halves)
__device__ __forceinline__ float bucket(int k) {
  const float step = (20.f - (-20.f)) / 254.f;
  return (k < kBuckets / 2) ? (-20.f + step * (float)k) : (20.f - step * (float)(kBuckets - 1 - k));
}
__device__ __forceinline__ float symlogf_(float x) {
  const float s = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
  return s * logf(fabsf(x) + 1.f);
}
__device__ __forceinline__ float symexpf_(float x) {
  const float s = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
  return s * (expf(fabsf(x)) - 1.f);
}

// one wave per row of 255 logits; lane owns classes lane + 64 v
struct DiscRow {
  float l[4], sm[4];
  float lse;
  __device__ __forceinline__ void load(const float* row, int lane) {
    float m = -INFINITY;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = lane + 64 * v;
      l[v] = (k < kBuckets) ? row[k] : -INFINITY;
      m = fmaxf(m, l[v]);
    }
    m = group_max<64>(m);
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      sm[v] = (lane + 64 * v < kBuckets) ? expf(l[v] - m) : 0.f;
      s += sm[v];
    }
    s = group_sum<64>(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int v = 0; v < 4; ++v) sm[v] *= inv;
    lse = m + logf(s);
  }
  __device__ __forceinline__ float mean(int lane) const {
    float a = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = lane + 64 * v;
      if (k < kBuckets) a += sm[v] * bucket(k);
    }
    return group_sum<64>(a);
  }
};

// two-hot target of tools.py:490-508 for symlog(x): indices and weights
__device__ __forceinline__ void twohot(float x, int lane, int& below, int& above, float& wb, float& wa) {
  const float y = symlogf_(x);
  int le = 0, gt = 0;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int k = lane + 64 * v;
    if (k < kBuckets) {
      const float b = bucket(k);
      le += (b <= y) ? 1 : 0;
      gt += (b > y) ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    le += __shfl_xor(le, o, 64);
    gt += __shfl_xor(gt, o, 64);
  }
  below = min(max(le - 1, 0), kBuckets - 1);
  above = min(max(kBuckets - gt, 0), kBuckets - 1);
  const bool eq = below == above;
  const float db = eq ? 1.f : fabsf(bucket(below) - y);
  const float da = eq ? 1.f : fabsf(bucket(above) - y);
  const float tot = db + da;
  wb = da / tot;
  wa = db / tot;
}

__global__ __launch_bounds__(256) void disc_mode_fwd_kernel(const float* __restrict__ logits,
                                                            float* __restrict__ out, long R) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
    DiscRow d;
    d.load(logits + r * kBuckets, lane);
    const float m = d.mean(lane);
    if (lane == 0) out[r] = symexpf_(m);
  }
}

// dlogits_k (+)= up[r] * exp(|m|) * sm_k * (b_k - m)
__global__ __launch_bounds__(256) void disc_mode_bwd_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ up,
                                                            float* __restrict__ dlogits, long R, int accumulate) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
    DiscRow d;
    d.load(logits + r * kBuckets, lane);
    const float m = d.mean(lane);
    const float c = up[r] * expf(fabsf(m));
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = lane + 64 * v;
      if (k < kBuckets) {
        const float g = c * d.sm[v] * (bucket(k) - m);
        float* o = dlogits + r * kBuckets + k;
        *o = accumulate ? (*o + g) : g;
      }
    }
  }
}

// out[r] = sum_k target_k (l_k - lse)
__global__ __launch_bounds__(256) void disc_logprob_fwd_kernel(const float* __restrict__ logits,
                                                               const float* __restrict__ x, float* __restrict__ out,
                                                               long R) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
    DiscRow d;
    d.load(logits + r * kBuckets, lane);
    int below, above;
    float wb, wa;
    twohot(x[r], lane, below, above, wb, wa);
    float a = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = lane + 64 * v;
      if (k < kBuckets) {
        const float t = (k == below ? wb : 0.f) + (k == above ? wa : 0.f);
        a += t * (d.l[v] - d.lse);
      }
    }
    a = group_sum<64>(a);
    if (lane == 0) out[r] = a;
  }
}

// d(logprob)/dlogits_k = target_k - sm_k ; dlogits (+)= up[r] * that
__global__ __launch_bounds__(256) void disc_logprob_bwd_kernel(const float* __restrict__ logits,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ up,
                                                               float* __restrict__ dlogits, long R, int accumulate) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
    DiscRow d;
    d.load(logits + r * kBuckets, lane);
    int below, above;
    float wb, wa;
    twohot(x[r], lane, below, above, wb, wa);
    const float u = up[r];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = lane + 64 * v;
      if (k < kBuckets) {
        const float t = (k == below ? wb : 0.f) + (k == above ? wa : 0.f);
        const float g = u * (t - d.sm[v]);
        float* o = dlogits + r * kBuckets + k;
        *o = accumulate ? (*o + g) : g;
      }
    }
  }
}

// Bernoulli(logits=l).log_prob(x) = -softplus(l)(1-x) - softplus(-l) x      (tools.py:622-627)
__device__ __forceinline__ float softplusf_(float z) { return (z > 20.f) ? z : log1pf(expf(z)); }
__global__ void bernoulli_logprob_fwd_kernel(const float* __restrict__ l, const float* __restrict__ x,
                                             float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = -softplusf_(l[i]) * (1.f - x[i]) - softplusf_(-l[i]) * x[i];
}
// d/dl = x - sigmoid(l)
__global__ void bernoulli_logprob_bwd_kernel(const float* __restrict__ l, const float* __restrict__ x,
                                             const float* __restrict__ up, float* __restrict__ dl, long n,
                                             int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = up[i] * (x[i] - sigmoidf_(l[i]));
    dl[i] = accumulate ? dl[i] + g : g;
  }
}

// Image reconstruction loss on raw u8 pixels: target = u8/255 (models.py:180), loss[img] = sum (recon-t)^2
// (tools.py:531-540), drecon = 2*up*(recon-t) written in the same pass when drecon != null.
// recon image index n' = t*B + b reads replay image b*T + t when permB > 0 (time-major activations)
__device__ __forceinline__ long src_image(long n, int permB, int permT) {
  if (permB <= 0) return n;
  const long t = n / permB, b = n - t * permB;
  return b * permT + t;
}
__global__ __launch_bounds__(256) void mse_image_kernel(const float* __restrict__ recon,
                                                        const unsigned char* __restrict__ image0,
                                                        float* __restrict__ loss, float* __restrict__ drecon, int P,
                                                        float up, int permB, int permT) {
  __shared__ float red[4];
  const long base = (long)blockIdx.x * P;
  const unsigned char* image = image0 + (src_image(blockIdx.x, permB, permT) - (long)blockIdx.x) * P;
  float a = 0.f;
  for (int i = threadIdx.x * 4; i < P; i += 256 * 4) {
    if (i + 3 < P) {
      const float4 r = *reinterpret_cast<const float4*>(recon + base + i);
      const uchar4 u = *reinterpret_cast<const uchar4*>(image + base + i);
      const float d0 = r.x - (float)u.x / 255.f, d1 = r.y - (float)u.y / 255.f;
      const float d2 = r.z - (float)u.z / 255.f, d3 = r.w - (float)u.w / 255.f;
      a += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      if (drecon) {
        float4 g = {2.f * up * d0, 2.f * up * d1, 2.f * up * d2, 2.f * up * d3};
        *reinterpret_cast<float4*>(drecon + base + i) = g;
      }
    } else {
      for (int e = i; e < P; ++e) {
        const float d = recon[base + e] - (float)image[base + e] / 255.f;
        a += d * d;
        if (drecon) drecon[base + e] = 2.f * up * d;
      }
    }
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) loss[blockIdx.x] = a;
}

// u8 image -> f32 (u8/255 - 0.5), the encoder's input (models.py:180 + networks.py:487)
__global__ void image_to_f32_kernel(const unsigned char* __restrict__ img, float* __restrict__ out, long n_images,
                                    int P, int permB, int permT) {
  const long per = P / 4;  // P % 4 == 0 (checked on the host)
  const long total = n_images * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / per, j = (i - n * per) * 4;
    const uchar4 u = *reinterpret_cast<const uchar4*>(img + src_image(n, permB, permT) * P + j);
    float4 o = {(float)u.x / 255.f - 0.5f, (float)u.y / 255.f - 0.5f, (float)u.z / 255.f - 0.5f,
                (float)u.w / 255.f - 0.5f};
    *reinterpret_cast<float4*>(out + n * P + j) = o;
  }
}

// [B][T][k] -> [T][B][k] (action / reward / is_first / proprio keys into time-major)
__global__ void transpose01_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, int k) {
  const long total = (long)B * T * k;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % k);
    const long r = i / k;
    const int b = (int)(r % B);
    const int t = (int)(r / B);
    y[i] = x[((long)b * T + t) * k + j];
  }
}

// out[n] (+)= sum_r x[r][n]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out,
                                                     long R, int N, int accumulate) {
  // block handles 64 columns x a slab of rows; 4 waves split the slab, lanes own columns
  __shared__ float red[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + lane;
  const long rows_per = (R + gridDim.y - 1) / gridDim.y;
  const long rb = (long)blockIdx.y * rows_per;
  long re = rb + rows_per;
  if (re > R) re = R;
  float a = 0.f;
  if (c < N)
    for (long r = rb + wave; r < re; r += 4) a += x[r * ldx + c];
  red[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && c < N) {
    const float s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (gridDim.y == 1 && !accumulate) out[c] = s;
    else atomicAdd(out + c, s);
  }
}

// Grouped form: the bias gradients of one cluster of weight gradients (ops.gemm_group) as ONE grid -- each is a 2 MB
// read behind a launch boundary of its own otherwise (11-19 us per launch on a 128-CU lane, ten per update).
constexpr int kMaxColGroup = 48;
struct ColsumItem {
  const float* x;
  float* out;
  long ldx;
  int R, N, blk0, by;
};
struct ColsumGroup {
  int n, total;
  ColsumItem it[kMaxColGroup];
};
__global__ __launch_bounds__(256) void colsum_grouped_kernel(ColsumGroup p) {
  __shared__ float red[4][64];
  int g = 0;
  while (g + 1 < p.n && (int)blockIdx.x >= p.it[g + 1].blk0) ++g;  // (uniform: scalar loads of the kernel arguments)
  const ColsumItem it = p.it[g];
  const int local = (int)blockIdx.x - it.blk0;
  const int nbx = (it.N + 63) / 64;
  const int bx = local % nbx, by = local / nbx;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = bx * 64 + lane;
  const long rows_per = ((long)it.R + it.by - 1) / it.by;
  const long rb = (long)by * rows_per;
  long re = rb + rows_per;
  if (re > it.R) re = it.R;
  float a = 0.f;
  if (c < it.N)
    for (long r = rb + wave; r < re; r += 4) a += it.x[r * it.ldx + c];
  red[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && c < it.N) atomicAdd(it.out + c, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// narrow matrices (N <= 32, e.g. the 3-channel image bias or the 1-wide continue head): every thread
// keeps one column (its global index modulo N is fixed because the stride is a multiple of N)
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float* __restrict__ x, long ldx,
                                                            float* __restrict__ out, long R, int N) {
  __shared__ float red[32];
  if (threadIdx.x < 32) red[threadIdx.x] = 0.f;
  __syncthreads();
  const long total = R * N;
  const long stride = (long)gridDim.x * blockDim.x;  // host guarantees stride % N == 0
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(i0 % N);
  float a = 0.f;
  if (ldx == N) {
    for (long i = i0; i < total; i += stride) a += x[i];
  } else {
    for (long i = i0; i < total; i += stride) a += x[(i / N) * ldx + c];
  }
  atomicAdd(&red[c], a);
  __syncthreads();
  if (threadIdx.x < N) atomicAdd(out + threadIdx.x, red[threadIdx.x]);
}

// One launch for the three is_first blends of an observe step (networks.py:181-193): stoch / deter
// against the learned initial state, action against zero.
__global__ void obs_blend_kernel(const float* __restrict__ ps, const float* __restrict__ s0,
                                 const float* __restrict__ pd, const float* __restrict__ d0,
                                 const float* __restrict__ act, const float* __restrict__ first,
                                 float* __restrict__ os, float* __restrict__ od, float* __restrict__ oa, int B,
                                 int SD, int De, int A) {
  const int W = SD + De + A;
  const long total = (long)B * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / W), j = (int)(i % W);
    const float m = first[b];
    if (j < SD) os[(long)b * SD + j] = (ps ? ps[(long)b * SD + j] : 0.f) * (1.f - m) + s0[j] * m;
    else if (j < SD + De) {
      const int k = j - SD;
      od[(long)b * De + k] = (pd ? pd[(long)b * De + k] : 0.f) * (1.f - m) + d0[k] * m;
    } else {
      const int k = j - SD - De;
      oa[(long)b * A + k] = act[(long)b * A + k] * (1.f - m);
    }
  }
}
// backward of the stoch / deter blends: gs_prev += dsin*(1-m), gd_prev += ddin*(1-m) (skipped when NULL),
// dstoch0 += sum_b dsin*m, ddeter0 += sum_b ddin*m
__global__ void obs_blend_bwd_kernel(const float* __restrict__ dsin, long ld_dsin, const float* __restrict__ ddin,
                                     long ld_ddin, const float* __restrict__ first, float* __restrict__ gs_prev,
                                     float* __restrict__ gd_prev, float* __restrict__ ds0,
                                     float* __restrict__ dd0, int B, int SD, int De) {
  // one thread per (row, column): the carry is a plain read-modify-write; the initial-state gradient only
  // receives anything on reset rows (rare), so its atomics almost never execute
  const int W = SD + De;
  const long total = (long)B * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / W), j = (int)(i % W);
    const bool is_s = j < SD;
    const int k = is_s ? j : j - SD;
    const int w = is_s ? SD : De;
    const float m = first[b];
    const float g = is_s ? dsin[(long)b * ld_dsin + k] : ddin[(long)b * ld_ddin + k];
    float* prev = is_s ? gs_prev : gd_prev;
    if (prev) prev[(long)b * w + k] += g * (1.f - m);
    if (m != 0.f) atomicAdd((is_s ? ds0 : dd0) + k, g * m);
  }
}

__global__ void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = tanhf(x[i]);
}
__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                long n, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = dy[i] * (1.f - y[i] * y[i]);
    dx[i] = accumulate ? dx[i] + g : g;
  }
}

// symlog-MSE (tools.SymlogDist, tools.py:558-572): loss[row] = sum_j d_j, d = (mode - symlog(x))^2 zeroed
// below 1e-8; dmode = 2*up*(mode - symlog(x)) where d >= 1e-8.
__global__ __launch_bounds__(256) void symlog_mse_kernel(const float* __restrict__ mode, const float* __restrict__ x,
                                                         float* __restrict__ loss, float* __restrict__ dmode,
                                                         long R, int W, float up) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
    float a = 0.f;
    for (int j = lane; j < W; j += 64) {
      const float df = mode[r * W + j] - symlogf_(x[r * W + j]);
      float d = df * df;
      const bool keep = !(d < 1e-8f);
      if (!keep) d = 0.f;
      a += d;
      if (dmode) dmode[r * W + j] = keep ? 2.f * up * df : 0.f;
    }
    a = group_sum<64>(a);
    if (lane == 0) loss[r] = a;
  }
}
// symlog of the proprio encoder input (networks.py:659-660)
__global__ void symlog_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = symlogf_(x[i]);
}

// ---- continuous actor: Normal(tanh(mean), (max-min)*sigmoid(std+2)+min), absmax=1 ------------------
// One thread per row (A is tiny).  Outputs action (sampled, rescaled), entropy, log-prob of the action.
__global__ void actor_normal_fwd_kernel(const float* __restrict__ mean_raw, const float* __restrict__ std_raw,
                                        const float* __restrict__ eps, float* __restrict__ action,
                                        float* __restrict__ ent, long M, int A, float min_std, float max_std) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += (long)gridDim.x * blockDim.x) {
    float e = 0.f;
    for (int a = 0; a < A; ++a) {
      const float mu = tanhf(mean_raw[r * A + a]);
      const float sd = (max_std - min_std) * sigmoidf_(std_raw[r * A + a] + 2.f) + min_std;
      if (action) {
        const float pre = mu + sd * eps[r * A + a];
        action[r * A + a] = pre * (1.f / fmaxf(fabsf(pre), 1.f));
      }
      e += 0.5f + 0.9189385332046727f + logf(sd);
    }
    if (ent) ent[r] = e;
  }
}
// log_prob of a given action under the same Normal (reinforce branch, models.py:667)
__global__ void actor_normal_logp_kernel(const float* __restrict__ mean_raw, const float* __restrict__ std_raw,
                                         const float* __restrict__ action, float* __restrict__ logp, long M, int A,
                                         float min_std, float max_std) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int a = 0; a < A; ++a) {
      const float mu = tanhf(mean_raw[r * A + a]);
      const float sd = (max_std - min_std) * sigmoidf_(std_raw[r * A + a] + 2.f) + min_std;
      const float d = action[r * A + a] - mu;
      s += -(d * d) / (2.f * sd * sd) - logf(sd) - 0.9189385332046727f;
    }
    logp[r] = s;
  }
}
// Backward of the three actor outputs w.r.t. (mean_raw, std_raw):
//   daction (through the rsample; the absmax rescale factor is a constant), dent, dlogp (log-prob of the sampled
//   action: logp_of_sample = 1: the action is the rsample of this very (mean, std) through eps, as in models.py:667;
//   0: the action is a constant).
// Any of daction / dent / dlogp may be null.
__global__ void actor_normal_bwd_kernel(const float* __restrict__ mean_raw, const float* __restrict__ std_raw,
                                        const float* __restrict__ eps, const float* __restrict__ action,
                                        const float* __restrict__ daction, const float* __restrict__ dent,
                                        const float* __restrict__ dlogp, float* __restrict__ dmean_raw,
                                        float* __restrict__ dstd_raw, long M, int A, float min_std,
                                        float max_std, int logp_of_sample) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += (long)gridDim.x * blockDim.x) {
    for (int a = 0; a < A; ++a) {
      const long i = r * A + a;
      const float mu = tanhf(mean_raw[i]);
      const float sg = sigmoidf_(std_raw[i] + 2.f);
      const float sd = (max_std - min_std) * sg + min_std;
      float dmu = 0.f, dsd = 0.f;
      if (daction) {
        const float pre = mu + sd * eps[i];
        const float g = daction[i] * (1.f / fmaxf(fabsf(pre), 1.f));
        dmu += g;
        dsd += g * eps[i];
      }
      if (dent) dsd += dent[r] / sd;
      if (dlogp) {
        // log N(a; mu, sd) with a = (mu + sd eps) * c the rsampled action itself (ContDist.sample, tools.py:594-598;
        // c = 1 / max(|mu + sd eps|, 1) is detached): the explicit dependence on (mu, sd) plus the path through a
        const float d = action[i] - mu;
        float dmu_l = d / (sd * sd), dsd_l = d * d / (sd * sd * sd) - 1.f / sd;
        if (logp_of_sample) {
          const float c = 1.f / fmaxf(fabsf(mu + sd * eps[i]), 1.f);
          const float dla = -d / (sd * sd);
          dmu_l += dla * c;
          dsd_l += dla * c * eps[i];
        }
        dmu += dlogp[r] * dmu_l;
        dsd += dlogp[r] * dsd_l;
      }
      dmean_raw[i] = dmu * (1.f - mu * mu);
      dstd_raw[i] = dsd * (max_std - min_std) * sg * (1.f - sg);
    }
  }
}

// ---- lambda-return and discount weights over the imagination horizon -------------------------------
// disc_t = gamma * sigmoid(cont_logit_t);  R_t = r_{t+1} + disc_{t+1}((1-lam) v_{t+1} + lam R_{t+1}),
// R_{H-1} := v_{H-1};  weights_t = prod_{s<t} disc_s.   All tensors [H][N].  One thread per column.
__global__ void lambda_return_fwd_kernel(const float* __restrict__ reward, const float* __restrict__ value,
                                         const float* __restrict__ cont_logit, float* __restrict__ target,
                                         float* __restrict__ weights, float* __restrict__ disc_out, int H, long N,
                                         float gamma, float lam) {
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
    float w = 1.f;
    for (int t = 0; t < H; ++t) {
      const float d = gamma * sigmoidf_(cont_logit[t * N + n]);
      weights[t * N + n] = w;
      if (disc_out) disc_out[t * N + n] = d;
      w *= d;
    }
    float agg = value[(long)(H - 1) * N + n];
    for (int t = H - 2; t >= 0; --t) {
      const float d = gamma * sigmoidf_(cont_logit[(t + 1) * N + n]);
      const float inp = reward[(t + 1) * N + n] + d * value[(t + 1) * N + n] * (1.f - lam);
      agg = inp + d * lam * agg;
      target[t * N + n] = agg;
    }
  }
}
// Given dtarget [H-1][N]: dreward [H][N] (row 0 = 0), dcont_logit [H][N] (row 0 = 0).  value carries no
// gradient on this path (models.py:626 evaluates it on the detached feats).
__global__ void lambda_return_bwd_kernel(const float* __restrict__ dtarget, const float* __restrict__ value,
                                         const float* __restrict__ cont_logit, const float* __restrict__ target,
                                         float* __restrict__ dreward, float* __restrict__ dcont_logit, int H, long N,
                                         float gamma, float lam) {
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
    dreward[n] = 0.f;
    dcont_logit[n] = 0.f;
    float G = 0.f;  // dL/dR_t, carried forward in t
    for (int t = 0; t <= H - 2; ++t) {
      const float sg = sigmoidf_(cont_logit[(t + 1) * N + n]);
      const float d = gamma * sg;
      G = dtarget[t * N + n] + ((t > 0) ? G : 0.f);
      const float rnext = (t + 1 <= H - 2) ? target[(t + 1) * N + n] : value[(long)(H - 1) * N + n];
      dreward[(t + 1) * N + n] = G;
      dcont_logit[(t + 1) * N + n] = G * ((1.f - lam) * value[(t + 1) * N + n] + lam * rnext) * gamma * sg * (1.f - sg);
      G = G * d * lam;  // flows into R_{t+1}
    }
  }
}

// ---- is_first reset blend (networks.py:181-193): out = x*(1-m) + init*m, m per row -------------------
__global__ void reset_blend_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ init,
                                   const float* __restrict__ first, float* __restrict__ out, long ldo, int B,
                                   int n) {
  const long total = (long)B * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / n), j = (int)(i % n);
    const float m = first[b];
    const float xv = x ? x[b * ldx + j] : 0.f;
    const float iv = init ? init[j] : 0.f;
    out[b * ldo + j] = xv * (1.f - m) + iv * m;
  }
}
// dx = dout*(1-m);  dinit[j] += sum_b dout[b][j]*m_b
__global__ void reset_blend_bwd_kernel(const float* __restrict__ dout, long ldo, const float* __restrict__ first,
                                       float* __restrict__ dx, long ldx, float* __restrict__ dinit, int B, int n) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      const float m = first[b];
      const float g = dout[b * ldo + j];
      if (dx) dx[b * ldx + j] = g * (1.f - m);
      acc += g * m;
    }
    if (dinit) dinit[j] += acc;
  }
}

// out[0] += sum_i f(x_i) * w_i ; f = max(., cmin) when use_clip.  (loss / metric reductions)
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ x, const float* __restrict__ w, long n,
                                                  float* __restrict__ out, int use_clip, float cmin, float scale) {
  __shared__ float red[4];
  float a = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (use_clip) v = fmaxf(v, cmin);
    a += w ? v * w[i] : v;
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) atomicAdd(out, a * scale);
}

// Actor loss of ImagBehavior._compute_actor_loss + the entropy bonus (models.py:406-407, 640-681), with
// its gradients, in one pass.  All [H][N] except target/base grads which use rows 0..H-2.
//   offset = ema[0], scale = max(ema[1]-ema[0], 1)                                  (models.py:24-25)
//   dynamics : loss_t = -w_t ((target_t-offset)/scale - (base_t-offset)/scale) - c ent_t ; dtarget = -w_t/(scale*cnt)
//   reinforce: loss_t = -w_t logp_t (target_t - base_t) - c ent_t                ; dlogp   = -w_t (target_t-base_t)/cnt
//   both     : loss_t = -w_t (mix target_t + (1-mix) logp_t sg(target_t - base_t)) - c ent_t; both gradients
// loss_out[0] += mean over the (H-1)*N entries; dent = -c/cnt on rows < H-1, 0 on the last row.
__global__ void actor_loss_kernel(const float* __restrict__ target, const float* __restrict__ value,
                                  const float* __restrict__ weights, const float* __restrict__ ent,
                                  const float* __restrict__ logp, const float* __restrict__ ema,
                                  float* __restrict__ loss_out, float* __restrict__ dtarget,
                                  float* __restrict__ dlogp, float* __restrict__ dent, int H, long N, float ent_coef,
                                  int mode, float mix) {
  __shared__ float red[4];
  const long cnt = (long)(H - 1) * N;
  const float inv = 1.f / (float)cnt;
  const float offset = ema[0], scale = fmaxf(ema[1] - ema[0], 1.f);
  float a = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)H * N; i += (long)gridDim.x * blockDim.x) {
    if (i < cnt) {
      const float w = weights[i];
      float tgt;
      if (mode == 1) {  // reinforce
        const float adv = target[i] - value[i];
        tgt = logp[i] * adv;
        dlogp[i] = -w * adv * inv;
      } else if (mode == 2) {  // both: mix * target + (1 - mix) * logp * sg(target - value)   (models.py:670-676)
        const float adv = target[i] - value[i];
        tgt = mix * target[i] + (1.f - mix) * logp[i] * adv;
        dlogp[i] = -w * (1.f - mix) * adv * inv;
        dtarget[i] = -w * mix * inv;
      } else {
        tgt = (target[i] - offset) / scale - (value[i] - offset) / scale;
        dtarget[i] = -w * inv / scale;
      }
      a += -w * tgt - ent_coef * ent[i];
      dent[i] = -ent_coef * inv;
    } else {
      dent[i] = 0.f;
      if (mode != 0) dlogp[i] = 0.f;
    }
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) atomicAdd(loss_out, a * inv);
}

// up[i] = -w[i] * inv_count  (the upstream of both critic log-prob terms, models.py:424-429)
__global__ void scale_neg_kernel(const float* __restrict__ w, float* __restrict__ up, long n, float s) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) up[i] = -w[i] * s;
}

static unsigned nblk(long n, int per, long cap) {
  long b = (n + per - 1) / per;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace dv3

using namespace dv3;
#define S_ ((hipStream_t)stream)

extern "C" int dv3_disc_mode_fwd(const float* logits, float* out, long R, void* stream) {
  if (R <= 0) return 0;
  if (!logits || !out) return DV3_ERR_ARG;
  hipLaunchKernelGGL(disc_mode_fwd_kernel, dim3(nblk(R, 4, 4096)), dim3(256), 0, S_, logits, out, R);
  return (int)hipGetLastError();
}
extern "C" int dv3_disc_mode_bwd(const float* logits, const float* up, float* dlogits, long R, int accumulate,
                                 void* stream) {
  if (R <= 0) return 0;
  if (!logits || !up || !dlogits) return DV3_ERR_ARG;
  hipLaunchKernelGGL(disc_mode_bwd_kernel, dim3(nblk(R, 4, 4096)), dim3(256), 0, S_, logits, up, dlogits, R, accumulate);
  return (int)hipGetLastError();
}
extern "C" int dv3_disc_logprob_fwd(const float* logits, const float* x, float* out, long R, void* stream) {
  if (R <= 0) return 0;
  if (!logits || !x || !out) return DV3_ERR_ARG;
  hipLaunchKernelGGL(disc_logprob_fwd_kernel, dim3(nblk(R, 4, 4096)), dim3(256), 0, S_, logits, x, out, R);
  return (int)hipGetLastError();
}
extern "C" int dv3_disc_logprob_bwd(const float* logits, const float* x, const float* up, float* dlogits, long R,
                                    int accumulate, void* stream) {
  if (R <= 0) return 0;
  if (!logits || !x || !up || !dlogits) return DV3_ERR_ARG;
  hipLaunchKernelGGL(disc_logprob_bwd_kernel, dim3(nblk(R, 4, 4096)), dim3(256), 0, S_, logits, x, up, dlogits, R,
                     accumulate);
  return (int)hipGetLastError();
}
extern "C" int dv3_bernoulli_logprob_fwd(const float* logit, const float* x, float* out, long n, void* stream) {
  if (n <= 0) return 0;
  if (!logit || !x || !out) return DV3_ERR_ARG;
  hipLaunchKernelGGL(bernoulli_logprob_fwd_kernel, dim3(nblk(n, 256, 2048)), dim3(256), 0, S_, logit, x, out, n);
  return (int)hipGetLastError();
}
extern "C" int dv3_bernoulli_logprob_bwd(const float* logit, const float* x, const float* up, float* dlogit, long n,
                                         int accumulate, void* stream) {
  if (n <= 0) return 0;
  if (!logit || !x || !up || !dlogit) return DV3_ERR_ARG;
  hipLaunchKernelGGL(bernoulli_logprob_bwd_kernel, dim3(nblk(n, 256, 2048)), dim3(256), 0, S_, logit, x, up, dlogit, n,
                     accumulate);
  return (int)hipGetLastError();
}
static bool perm_ok(long n_images, int permB, int permT) {
  return (permB <= 0 && permT <= 0) || (permB > 0 && permT > 0 && (long)permB * permT == n_images);
}
extern "C" int dv3_mse_image(const float* recon, const unsigned char* image_u8, float* loss, float* drecon,
                             long n_images, int pixels, float upstream, int perm_B, int perm_T, void* stream) {
  if (n_images <= 0) return 0;
  if (!recon || !image_u8 || !loss || pixels <= 0 || (pixels % 4) != 0 || !perm_ok(n_images, perm_B, perm_T))
    return DV3_ERR_ARG;
  hipLaunchKernelGGL(mse_image_kernel, dim3((unsigned)n_images), dim3(256), 0, S_, recon, image_u8, loss, drecon, pixels,
                     upstream, perm_B, perm_T);
  return (int)hipGetLastError();
}
extern "C" int dv3_image_to_f32(const unsigned char* image_u8, float* out, long n_images, int pixels, int perm_B,
                                int perm_T, void* stream) {
  if (n_images <= 0) return 0;
  if (!image_u8 || !out || pixels <= 0 || (pixels % 4) != 0 || !perm_ok(n_images, perm_B, perm_T)) return DV3_ERR_ARG;
  hipLaunchKernelGGL(image_to_f32_kernel, dim3(nblk(n_images * (pixels / 4), 256, 8192)), dim3(256), 0, S_, image_u8, out,
                     n_images, pixels, perm_B, perm_T);
  return (int)hipGetLastError();
}
extern "C" int dv3_transpose01(const float* x, float* y, int B, int T, int k, void* stream) {
  if (B <= 0 || T <= 0 || k <= 0) return 0;
  if (!x || !y) return DV3_ERR_ARG;
  hipLaunchKernelGGL(transpose01_kernel, dim3(nblk((long)B * T * k, 256, 2048)), dim3(256), 0, S_, x, y, B, T, k);
  return (int)hipGetLastError();
}
extern "C" int dv3_colsum(const float* x, long ldx, float* out, long R, int N, int accumulate, void* stream) {
  if (R <= 0 || N <= 0) return 0;
  if (!x || !out) return DV3_ERR_ARG;
  if (N <= 32 && R >= 4096) {
    if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * N, S_);
    long blocks = (R * N + 256L * 64 - 1) / (256L * 64);
    if (blocks > 1024) blocks = 1024;
    blocks = ((blocks + N - 1) / N) * N;  // stride = blocks*256 must be a multiple of N ... 256*blocks % N == 0
    hipLaunchKernelGGL(colsum_narrow_kernel, dim3((unsigned)blocks), dim3(256), 0, S_, x, ldx, out, R, N);
    return (int)hipGetLastError();
  }
  const unsigned bx = (unsigned)((N + 63) / 64);
  // row slabs of 64 (16 rows per wave): a 1024 x 1024 bias gradient then runs on 256 workgroups instead of 64
  // (11 -> ~6 us; these launches sit between the weight-gradient GEMMs)
  long by = (R + 63) / 64;
  if (by > 64) by = 64;
  if (by < 1) by = 1;
  if (by > 1 && !accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * N, S_);
  hipLaunchKernelGGL(colsum_kernel, dim3(bx, (unsigned)by), dim3(256), 0, S_, x, ldx, out, R, N, accumulate);
  return (int)hipGetLastError();
}
extern "C" int dv3_colsum_grouped(int n, const float* const* x, const long* ldx, float* const* out, const long* R,
                                  const int* N, void* stream) {
  if (n <= 0) return 0;
  if (n > kMaxColGroup || !x || !ldx || !out || !R || !N) return DV3_ERR_ARG;
  ColsumGroup p{};
  int total = 0;
  for (int g = 0; g < n; ++g) {
    if (R[g] <= 0 || R[g] > 0x7fffffffL || N[g] <= 0 || !x[g] || !out[g] || ldx[g] < N[g]) return DV3_ERR_ARG;
    ColsumItem& it = p.it[p.n++];
    it.x = x[g]; it.out = out[g]; it.ldx = ldx[g];
    it.R = (int)R[g]; it.N = N[g];
    long by = (R[g] + 63) / 64;
    if (by > 64) by = 64;
    it.by = (int)by;
    it.blk0 = total;
    total += ((N[g] + 63) / 64) * it.by;
  }
  p.total = total;
  hipLaunchKernelGGL(colsum_grouped_kernel, dim3((unsigned)total), dim3(256), 0, S_, p);
  return (int)hipGetLastError();
}
extern "C" int dv3_tanh_fwd(const float* x, float* y, long n, void* stream) {
  if (n <= 0) return 0;
  if (!x || !y) return DV3_ERR_ARG;
  hipLaunchKernelGGL(tanh_fwd_kernel, dim3(nblk(n, 256, 2048)), dim3(256), 0, S_, x, y, n);
  return (int)hipGetLastError();
}
extern "C" int dv3_tanh_bwd(const float* y, const float* dy, float* dx, long n, int accumulate, void* stream) {
  if (n <= 0) return 0;
  if (!y || !dy || !dx) return DV3_ERR_ARG;
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(nblk(n, 256, 2048)), dim3(256), 0, S_, y, dy, dx, n, accumulate);
  return (int)hipGetLastError();
}
extern "C" int dv3_symlog_mse(const float* mode, const float* x, float* loss, float* dmode, long R, int W,
                              float upstream, void* stream) {
  if (R <= 0) return 0;
  if (!mode || !x || !loss || W <= 0) return DV3_ERR_ARG;
  hipLaunchKernelGGL(symlog_mse_kernel, dim3(nblk(R, 4, 4096)), dim3(256), 0, S_, mode, x, loss, dmode, R, W, upstream);
  return (int)hipGetLastError();
}
extern "C" int dv3_symlog(const float* x, float* y, long n, void* stream) {
  if (n <= 0) return 0;
  if (!x || !y) return DV3_ERR_ARG;
  hipLaunchKernelGGL(symlog_kernel, dim3(nblk(n, 256, 2048)), dim3(256), 0, S_, x, y, n);
  return (int)hipGetLastError();
}
extern "C" int dv3_actor_normal_fwd(const float* mean_raw, const float* std_raw, const float* eps, float* action,
                                    float* entropy, long M, int A, float min_std, float max_std, void* stream) {
  if (M <= 0) return 0;
  if (!mean_raw || !std_raw || A <= 0 || (action && !eps)) return DV3_ERR_ARG;
  hipLaunchKernelGGL(actor_normal_fwd_kernel, dim3(nblk(M, 256, 2048)), dim3(256), 0, S_, mean_raw, std_raw, eps, action,
                     entropy, M, A, min_std, max_std);
  return (int)hipGetLastError();
}
extern "C" int dv3_actor_normal_logp(const float* mean_raw, const float* std_raw, const float* action, float* logp,
                                     long M, int A, float min_std, float max_std, void* stream) {
  if (M <= 0) return 0;
  if (!mean_raw || !std_raw || !action || !logp || A <= 0) return DV3_ERR_ARG;
  hipLaunchKernelGGL(actor_normal_logp_kernel, dim3(nblk(M, 256, 2048)), dim3(256), 0, S_, mean_raw, std_raw, action,
                     logp, M, A, min_std, max_std);
  return (int)hipGetLastError();
}
extern "C" int dv3_actor_normal_bwd(const float* mean_raw, const float* std_raw, const float* eps,
                                    const float* action, const float* daction, const float* dent, const float* dlogp,
                                    float* dmean_raw, float* dstd_raw, long M, int A, float min_std, float max_std,
                                    int logp_of_sample, void* stream) {
  if (M <= 0) return 0;
  if (!mean_raw || !std_raw || !dmean_raw || !dstd_raw || A <= 0) return DV3_ERR_ARG;
  if ((daction && !eps) || (dlogp && !action) || (dlogp && logp_of_sample && !eps)) return DV3_ERR_ARG;
  hipLaunchKernelGGL(actor_normal_bwd_kernel, dim3(nblk(M, 256, 2048)), dim3(256), 0, S_, mean_raw, std_raw, eps, action,
                     daction, dent, dlogp, dmean_raw, dstd_raw, M, A, min_std, max_std, logp_of_sample);
  return (int)hipGetLastError();
}
extern "C" int dv3_lambda_return_fwd(const float* reward, const float* value, const float* cont_logit, float* target,
                                     float* weights, float* disc, int H, long N, float gamma, float lam,
                                     void* stream) {
  if (N <= 0) return 0;
  if (H < 2 || !reward || !value || !cont_logit || !target || !weights) return DV3_ERR_ARG;
  hipLaunchKernelGGL(lambda_return_fwd_kernel, dim3(nblk(N, 256, 2048)), dim3(256), 0, S_, reward, value, cont_logit,
                     target, weights, disc, H, N, gamma, lam);
  return (int)hipGetLastError();
}
extern "C" int dv3_lambda_return_bwd(const float* dtarget, const float* value, const float* cont_logit,
                                     const float* target, float* dreward, float* dcont_logit, int H, long N,
                                     float gamma, float lam, void* stream) {
  if (N <= 0) return 0;
  if (H < 2 || !dtarget || !value || !cont_logit || !target || !dreward || !dcont_logit) return DV3_ERR_ARG;
  hipLaunchKernelGGL(lambda_return_bwd_kernel, dim3(nblk(N, 256, 2048)), dim3(256), 0, S_, dtarget, value, cont_logit,
                     target, dreward, dcont_logit, H, N, gamma, lam);
  return (int)hipGetLastError();
}
extern "C" int dv3_reset_blend(const float* x, long ldx, const float* init, const float* is_first, float* out,
                               long ldo, int B, int n, void* stream) {
  if (B <= 0 || n <= 0) return 0;
  if (!is_first || !out) return DV3_ERR_ARG;
  hipLaunchKernelGGL(reset_blend_kernel, dim3(nblk((long)B * n, 256, 2048)), dim3(256), 0, S_, x, ldx, init, is_first,
                     out, ldo, B, n);
  return (int)hipGetLastError();
}
extern "C" int dv3_reset_blend_bwd(const float* dout, long ldo, const float* is_first, float* dx, long ldx,
                                   float* dinit, int B, int n, void* stream) {
  if (B <= 0 || n <= 0) return 0;
  if (!dout || !is_first) return DV3_ERR_ARG;
  hipLaunchKernelGGL(reset_blend_bwd_kernel, dim3(nblk(n, 256, 2048)), dim3(256), 0, S_, dout, ldo, is_first, dx, ldx,
                     dinit, B, n);
  return (int)hipGetLastError();
}

extern "C" int dv3_dot_accumulate(const float* x, const float* w, long n, float* out, int use_clip_min, float clip_min,
                                  float scale, void* stream) {
  if (n <= 0) return 0;
  if (!x || !out) return DV3_ERR_ARG;
  hipLaunchKernelGGL(dot_kernel, dim3(nblk(n, 1024, 512)), dim3(256), 0, S_, x, w, n, out, use_clip_min, clip_min, scale);
  return (int)hipGetLastError();
}
extern "C" int dv3_actor_loss(const float* target, const float* value, const float* weights, const float* entropy,
                              const float* logp, const float* ema_vals, float* loss_out, float* dtarget, float* dlogp,
                              float* dentropy, int H, long N, float entropy_coef, int mode, float mix, void* stream) {
  if (N <= 0) return 0;
  if (H < 2 || !target || !value || !weights || !entropy || !ema_vals || !loss_out || !dentropy) return DV3_ERR_ARG;
  if (mode < 0 || mode > 2) return DV3_ERR_ARG;
  if (mode != 0 && (!logp || !dlogp)) return DV3_ERR_ARG;
  if (mode != 1 && !dtarget) return DV3_ERR_ARG;
  hipLaunchKernelGGL(actor_loss_kernel, dim3(nblk((long)H * N, 1024, 512)), dim3(256), 0, S_, target, value, weights,
                     entropy, logp, ema_vals, loss_out, dtarget, dlogp, dentropy, H, N, entropy_coef, mode, mix);
  return (int)hipGetLastError();
}
extern "C" int dv3_scale_neg(const float* w, float* out, long n, float s, void* stream) {
  if (n <= 0) return 0;
  if (!w || !out) return DV3_ERR_ARG;
  hipLaunchKernelGGL(scale_neg_kernel, dim3(nblk(n, 256, 2048)), dim3(256), 0, S_, w, out, n, s);
  return (int)hipGetLastError();
}

extern "C" int dv3_obs_blend(const float* prev_stoch, const float* init_stoch, const float* prev_deter,
                             const float* init_deter, const float* action, const float* is_first, float* out_stoch,
                             float* out_deter, float* out_action, int B, int SD, int De, int A, void* stream) {
  if (B <= 0) return 0;
  if (!init_stoch || !init_deter || !action || !is_first || !out_stoch || !out_deter || !out_action) return DV3_ERR_ARG;
  if ((prev_stoch == nullptr) != (prev_deter == nullptr)) return DV3_ERR_ARG;
  hipLaunchKernelGGL(obs_blend_kernel, dim3(nblk((long)B * (SD + De + A), 256, 1024)), dim3(256), 0, S_, prev_stoch,
                     init_stoch, prev_deter, init_deter, action, is_first, out_stoch, out_deter, out_action, B, SD, De, A);
  return (int)hipGetLastError();
}
extern "C" int dv3_obs_blend_bwd(const float* dsin, long ld_dsin, const float* ddin, long ld_ddin, const float* is_first,
                                 float* gs_prev, float* gd_prev, float* dstoch0, float* ddeter0, int B, int SD, int De,
                                 void* stream) {
  if (B <= 0) return 0;
  if (!dsin || !ddin || !is_first || !dstoch0 || !ddeter0 || ld_dsin < SD || ld_ddin < De) return DV3_ERR_ARG;
  if ((gs_prev == nullptr) != (gd_prev == nullptr)) return DV3_ERR_ARG;
  hipLaunchKernelGGL(obs_blend_bwd_kernel, dim3(nblk((long)B * (SD + De), 256, 1024)), dim3(256), 0, S_, dsin, ld_dsin,
                     ddin, ld_ddin, is_first, gs_prev, gd_prev, dstoch0, ddeter0, B, SD, De);
  return (int)hipGetLastError();
}
