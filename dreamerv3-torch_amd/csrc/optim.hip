// Optimizer step on flat fp32 buckets: global grad-norm, clip, Adam -- one launch each per
// optimizer, no host synchronisation (the step counter and the norm live in device memory so the
// whole update can sit inside one hipGraph).
//
// Reference: tools.Optimizer.__call__ (tools.py:760-776): clip_grad_norm_(params, clip) then
// torch.optim.Adam(lr, eps).step(), betas (0.9, 0.999); slow critic EMA models.py:683-689.
// The flat bucket is also the unit of the data-parallel all-reduce (one RCCL call per optimizer).
#include "dv3_common.h"

namespace dv3 {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  const long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) {
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) a += x[i] * x[i];
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) atomicAdd(out, a);
}

// The same sum in a FIXED order: block b leaves its partial sum in partial[b] (no atomics), one workgroup adds the
// partials up by index.  The gradient norm -- and with it the clipping coefficient -- is then a pure function of the
// gradient: data-parallel replicas that hold the same all-reduced gradient take bit-identical Adam steps (with the
// atomic form their norms differed in the last bit and the replicas drifted apart by 1e-9 per clipped update).
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float a = 0.f;
  const long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) {
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) a += x[i] * x[i];
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 256) a += partial[i];
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) out[0] += a;
}

// state[0] = step count (float, exact up to 2^24), state[1] = sum of squared grads (input),
// state[2] = grad norm (output, for the `*_grad_norm` metric)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   float* __restrict__ state, double lr, double b1d, double b2d,
                                                   double epsd, float clip, float wd, float gscale) {
  const float step = state[0] + 1.f;
  const float norm = sqrtf(state[1]) * gscale;
  float coef = 1.f;
  if (clip > 0.f) coef = fminf(clip / (norm + 1e-6f), 1.f);
  // per-step scalars in double, rounded to float once -- exactly what torch.optim.Adam's single-tensor path does
  // (bias_correction = 1 - beta ** step, step_size = lr / bias_correction1, value = 1 - beta2, ...)
  const float w1 = (float)(1.0 - b1d), w2 = (float)(1.0 - b2d), b2 = (float)b2d, eps = (float)epsd;
  const float bc2s = (float)sqrt(1.0 - pow(b2d, (double)step));
  const float step_size = (float)(lr / (1.0 - pow(b1d, (double)step)));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * (coef * gscale);
    float pi = p[i];
    if (wd > 0.f) pi *= (1.f - wd);  // tools.py:778-783 (applied before the step, as the reference does)
    const float mi = m[i] + w1 * (gi - m[i]);
    const float vi = v[i] * b2 + w2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}
// runs after adam_kernel on the same stream: bump the step, publish the norm, clear the accumulator
__global__ void adam_finish_kernel(float* __restrict__ state, float gscale) {
  state[0] += 1.f;
  state[2] = sqrtf(state[1]) * gscale;
  state[1] = 0.f;
}

// y = a*x + b*y   (slow-critic EMA: a = mix, b = 1 - mix; gradient averaging etc.)
__global__ void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float a, float b) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = a * x[i] + b * y[i];
}

// N(0,1) fill (Box-Muller over Philox): the actor's rsample noise (torch _standard_normal in the
// reference, networks.py:697-699) when the caller injects none.  One counter per 4 outputs.
__global__ void fill_normal_kernel(float* __restrict__ out, long n, const unsigned long long* __restrict__ st,
                                   unsigned long long offset_add) {
  const unsigned long long seed = st[0], offset = st[1] + offset_add;
  const long n4 = (n + 3) >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    uint32_t o[4];
    Philox ph(seed);
    ph(offset + (unsigned long long)i, 0x6e6f726dULL, o);
    const float r0 = sqrtf(-2.f * logf(u01(o[0]))), r1 = sqrtf(-2.f * logf(u01(o[2])));
    const float a0 = 6.283185307179586f * u01(o[1]), a1 = 6.283185307179586f * u01(o[3]);
    const float z[4] = {r0 * cosf(a0), r0 * sinf(a0), r1 * cosf(a1), r1 * sinf(a1)};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * i + e < n) out[4 * i + e] = z[e];
  }
}


// Two quantiles of n values by exact radix selection + the return-normalisation EMA (models.RewardEMA, models.py:11-26:
// torch.quantile(x, [0.05, 0.95]) -- linear interpolation between the two neighbouring order statistics at
// q (n - 1) -- then ema = alpha q + (1 - alpha) ema).  One workgroup: the four order statistics (floor / ceil rank of
// each quantile) are selected together, one key byte per pass (4 passes over x, LDS histograms), so the result is
// the exact value torch.sort would give; n = (H-1) B T is 14 k .. 460 k on the BASELINE configs.
__device__ __forceinline__ uint32_t fkey(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// One LDS atomic per DISTINCT bucket among a wave's lanes instead of one per lane: the lanes that share the leader's
// bucket are counted with a ballot and the leader adds the count.  Returns concentrate in a narrow range, so in the first
// passes nearly every lane hits the same bucket (64 serialised LDS atomics per wave and batch: 106 us at 14 k values).
__device__ __forceinline__ void wave_hist_add(unsigned int* h, uint32_t b, bool valid) {
  unsigned long long active = __ballot(valid);
  const int lane = threadIdx.x & 63;
  // the two most contended buckets are added once each (leader + ballot count); what is left is spread over many
  // buckets (later passes: the matching values differ in the byte being counted) and goes through plain LDS atomics,
  // which do not conflict there -- walking every distinct bucket with ballots cost 65 us at 14 k values
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (!active) return;  // wave-uniform
    const int leader = __ffsll((long long)active) - 1;
    const uint32_t lb = (uint32_t)__shfl((int)b, leader, 64);
    const unsigned long long m = __ballot(valid && b == lb);
    if (lane == leader) atomicAdd(&h[lb], (unsigned int)__popcll(m));
    active &= ~m;
    valid = valid && b != lb;
  }
  if (valid) atomicAdd(&h[b], 1u);
}
// torch.lerp (ATen/native/Lerp.h), the interpolation torch.quantile applies between the two neighbouring order statistics
__device__ __forceinline__ float torch_lerp(float a, float b, float w) {
  return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);
}
__global__ __launch_bounds__(1024) void quantile2_ema_kernel(const float* __restrict__ x, long n, double q0, double q1,
                                                             float* __restrict__ ema, float alpha,
                                                             float* __restrict__ out_q) {
  constexpr int kU = 8, kR = 16;
  __shared__ unsigned int hist[4][256];
  __shared__ unsigned int whist[16][256];  // per-wave histogram for the passes in which the four prefixes coincide
  __shared__ uint32_t prefix[4];
  __shared__ long rank[4];
  __shared__ float frac[2];
  const int tid = threadIdx.x, wave = tid >> 6;
  if (tid == 0) {
    // ranks in float32 as torch.quantile computes them (q is a float32 tensor, models.py:16: ranks = q * (n - 1))
    const float p0 = (float)q0 * (float)(n - 1), p1 = (float)q1 * (float)(n - 1);
    rank[0] = (long)floorf(p0); rank[1] = (long)ceilf(p0);
    rank[2] = (long)floorf(p1); rank[3] = (long)ceilf(p1);
    frac[0] = p0 - floorf(p0); frac[1] = p1 - floorf(p1);
    for (int t = 0; t < 4; ++t) prefix[t] = 0u;
  }
  uint32_t mask = 0u;
  int has_nan = 0;
  // up to 16 k values (cfg 1 / cfg 2: 14 x 1024 returns) stay in registers for all four passes: ONE batch of loads
  const bool cached = n <= 1024L * kR;
  float vc[kR];
  if (cached) {
#pragma unroll
    for (int u = 0; u < kR; ++u) {
      const long i = (long)u * 1024 + tid;
      vc[u] = i < n ? x[i] : 0.f;
      has_nan |= (vc[u] != vc[u]);
    }
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = tid; i < 4 * 256; i += 1024) (&hist[0][0])[i] = 0u;
    for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t p0 = prefix[0], p1 = prefix[1], p2 = prefix[2], p3 = prefix[3];
    // Returns concentrate in a narrow range, so in the first passes (nearly) every element lands in ONE bucket: four
    // shared-histogram atomics per element serialised the whole workgroup on it (79 us at 14k values).  While the four
    // order statistics still share their prefix one count serves all of them, taken in wave-private histograms.
    const bool same = (p0 == p1) && (p1 == p2) && (p2 == p3);  // uniform
    auto count = [&](float v, bool in) {
      const uint32_t k = fkey(v);
      const uint32_t km = k & mask, bkt = (k >> shift) & 255u;
      if (same) {
        wave_hist_add(whist[wave], bkt, in && km == p0);
      } else {
        wave_hist_add(hist[0], bkt, in && km == p0);
        wave_hist_add(hist[1], bkt, in && km == p1);
        wave_hist_add(hist[2], bkt, in && km == p2);
        wave_hist_add(hist[3], bkt, in && km == p3);
      }
    };
    if (cached) {  // the values sit in registers since before the first pass
#pragma unroll
      for (int u = 0; u < kR; ++u) count(vc[u], (long)u * 1024 + tid < n);
    } else {
      // kU independent loads in flight per lane and chunk: one load per loop trip exposed a full memory round trip
      // per 1024 values (14 trips x 4 passes ~ 100 us at 14 k values: that WAS the cost of this kernel)
      for (long i0 = 0; i0 < n; i0 += 1024 * kU) {  // uniform trip count: the ballots inside need every lane
        float v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const long i = i0 + (long)u * 1024 + tid;
          v[u] = i < n ? x[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (pass == 0) has_nan |= (v[u] != v[u]);
          count(v[u], i0 + (long)u * 1024 + tid < n);
        }
      }
    }
    if (same) {
      __syncthreads();
      if (tid < 256) {
        unsigned int c = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) c += whist[w][tid];
        hist[0][tid] = hist[1][tid] = hist[2][tid] = hist[3][tid] = c;
      }
    }
    __syncthreads();
    if (wave < 4) {
      // wave t locates statistic t's bucket: lane l owns buckets 4l .. 4l+3, an inclusive scan over the lanes gives the
      // count below each lane's first bucket (a serial walk over 255 buckets by one thread cost ~12 us per pass)
      const int l = tid & 63;
      const unsigned int c0 = hist[wave][4 * l], c1 = hist[wave][4 * l + 1], c2 = hist[wave][4 * l + 2],
                         c3 = hist[wave][4 * l + 3];
      long incl = (long)c0 + c1 + c2 + c3;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const long up = __shfl_up(incl, o, 64);
        if (l >= o) incl += up;
      }
      const long r = rank[wave];
      const long below = incl - ((long)c0 + c1 + c2 + c3);  // elements in buckets before this lane's
      const bool mine = (r >= below && r < incl) || (l == 63 && r >= incl);  // (r >= total cannot happen: guard only)
      if (mine) {
        long rr = r - below;
        unsigned int b = 4 * l;
        if (rr >= (long)c0) { rr -= c0; ++b;
          if (rr >= (long)c1) { rr -= c1; ++b;
            if (rr >= (long)c2) { rr -= c2; ++b; } } }
        rank[wave] = rr;
        prefix[wave] |= b << shift;
      }
    }
    mask |= 0xFFu << shift;
    __syncthreads();
  }
  // a NaN anywhere makes both quantiles NaN, as torch.quantile does (the EMA then turns NaN too instead of training
  // on silently wrong statistics)
  const int any_nan = __syncthreads_or(has_nan);
  if (tid == 0) {
    const float a0 = fkey_inv(prefix[0]), a1 = fkey_inv(prefix[1]), b0 = fkey_inv(prefix[2]), b1 = fkey_inv(prefix[3]);
    float v0 = torch_lerp(a0, a1, frac[0]), v1 = torch_lerp(b0, b1, frac[1]);
    if (any_nan) v0 = v1 = NAN;
    if (out_q) { out_q[0] = v0; out_q[1] = v1; }
    if (ema) {
      ema[0] = alpha * v0 + (1.f - alpha) * ema[0];
      ema[1] = alpha * v1 + (1.f - alpha) * ema[1];
    }
  }
}

// advance the Philox offset kept in device memory: rng_state = {seed, offset}
__global__ void rng_advance_kernel(unsigned long long* st, unsigned long long inc) { st[1] += inc; }

}  // namespace dv3

using namespace dv3;

static unsigned blocks_for(long n, long cap) {
  long b = (n + 1023) / 1024;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

extern "C" int dv3_sumsq_accumulate(const float* x, long n, float* out, void* stream) {
  if (n <= 0) return 0;
  if (!x || !out || ((uintptr_t)x & 15)) return DV3_ERR_ARG;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, n, out);
  return (int)hipGetLastError();
}

extern "C" int dv3_sumsq_ordered(const float* x, long n, float* out, float* partial, int partial_len, void* stream) {
  if (n <= 0) return 0;
  if (!x || !out || !partial || partial_len < 1 || ((uintptr_t)x & 15)) return DV3_ERR_ARG;
  const unsigned blocks = blocks_for(n, partial_len < 1024 ? partial_len : 1024);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, s, x, n, partial);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, partial, (int)blocks, out);
  return (int)hipGetLastError();
}

extern "C" int dv3_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float* state,
                             double lr, double beta1, double beta2, double eps, float clip, float weight_decay,
                             float grad_scale, void* stream) {
  if (n <= 0) return 0;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !state) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n, 2048)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, state,
                     lr, beta1, beta2, eps, clip, weight_decay, grad_scale);
  hipLaunchKernelGGL(adam_finish_kernel, dim3(1), dim3(1), 0, s, state, grad_scale);
  return (int)hipGetLastError();
}

extern "C" int dv3_axpby(const float* x, float* y, long n, float a, float b, void* stream) {
  if (n <= 0) return 0;
  if (!x || !y) return DV3_ERR_ARG;
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, x, y, n, a, b);
  return (int)hipGetLastError();
}

extern "C" int dv3_rng_advance(unsigned long long* rng_state, unsigned long long increment, void* stream) {
  if (!rng_state) return DV3_ERR_ARG;
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state, increment);
  return (int)hipGetLastError();
}

extern "C" int dv3_fill_normal(float* out, long n, const unsigned long long* rng_state, unsigned long long rng_offset,
                               void* stream) {
  if (n <= 0) return 0;
  if (!out || !rng_state) return DV3_ERR_ARG;
  hipLaunchKernelGGL(fill_normal_kernel, dim3(blocks_for((n + 3) / 4, 2048)), dim3(256), 0, (hipStream_t)stream, out, n,
                     rng_state, rng_offset);
  return (int)hipGetLastError();
}

extern "C" int dv3_quantile2_ema(const float* x, long n, double q0, double q1, float* ema, float alpha, float* out_q,
                                 void* stream) {
  if (n <= 0 || !x || (!ema && !out_q) || q0 < 0.0 || q0 > 1.0 || q1 < 0.0 || q1 > 1.0) return DV3_ERR_ARG;
  hipLaunchKernelGGL(quantile2_ema_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, q0, q1, ema, alpha, out_q);
  return (int)hipGetLastError();
}

// tools.tensorstats (tools.py:949-958): out[0..3] = mean, std (unbiased, torch.std), min, max of (x[i] - shift) * scale
// in one launch (the reference's four reductions per logged tensor).  shift / scale (optional device scalars) fold
// the normed_target expression (models.py:412-414) into the read.
struct StatsJobs {
  const float* x[6];
  long n[6];
  const float* shift[6];
  const float* scale[6];
};
__device__ void tensorstats_body(const float* __restrict__ x, long n, const float* __restrict__ shift,
                                 const float* __restrict__ inv_scale, float* __restrict__ out);
__global__ __launch_bounds__(1024) void tensorstats_kernel(const float* __restrict__ x, long n,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ inv_scale, float* __restrict__ out) {
  tensorstats_body(x, n, shift, inv_scale, out);
}
// one workgroup per logged tensor: the five statistics groups of a behaviour update (models.py:431-445) in ONE launch
__global__ __launch_bounds__(1024) void tensorstats_multi_kernel(StatsJobs j, float* __restrict__ out) {
  const int b = blockIdx.x;
  tensorstats_body(j.x[b], j.n[b], j.shift[b], j.scale[b], out + 4 * b);
}
__device__ void tensorstats_body(const float* __restrict__ x, long n, const float* __restrict__ shift,
                                 const float* __restrict__ inv_scale, float* __restrict__ out) {
  // one pass: sum and sum of squares in double (n <= ~1e6 logged values of O(1..100): the cancellation in
  // sum(x^2) - n mean^2 stays far below float resolution), min / max in float
  __shared__ double red[2][16];
  __shared__ float redmin[16], redmax[16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float sh = shift ? shift[0] : 0.f;
  const float dv = inv_scale ? inv_scale[0] : 1.f;
  double s = 0.0, q = 0.0;
  float mn = INFINITY, mx = -INFINITY;
  auto take = [&](float raw) {
    const float v = inv_scale ? (raw - sh) / dv : (raw - sh);
    s += (double)v;
    q += (double)v * (double)v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  };
  const long n4 = (((uintptr_t)x & 15) == 0) ? (n >> 2) : 0;
  for (long i = tid; i < n4; i += 1024) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    take(v.x), take(v.y), take(v.z), take(v.w);
  }
  for (long i = 4 * n4 + tid; i < n; i += 1024) take(x[i]);
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = q;
    redmin[wave] = mn;
    redmax[wave] = mx;
  }
  __syncthreads();
  if (tid == 0) {
    double ts = 0.0, tq = 0.0;
    float a = INFINITY, b = -INFINITY;
    for (int w = 0; w < 16; ++w) {
      ts += red[0][w];
      tq += red[1][w];
      a = fminf(a, redmin[w]);
      b = fmaxf(b, redmax[w]);
    }
    const double mean = ts / (double)n;
    double var = (tq - (double)n * mean * mean) / (double)(n - 1);
    if (var < 0.0) var = 0.0;
    out[0] = (float)mean;
    out[1] = n > 1 ? (float)sqrt(var) : NAN;
    out[2] = a;
    out[3] = b;
  }
}

extern "C" int dv3_tensorstats(const float* x, long n, const float* shift, const float* scale, float* out4,
                               void* stream) {
  if (n <= 0 || !x || !out4) return DV3_ERR_ARG;
  hipLaunchKernelGGL(tensorstats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, shift, scale, out4);
  return (int)hipGetLastError();
}

extern "C" int dv3_tensorstats_multi(int count, const float* x0, long n0, const float* shift0, const float* scale0,
                                     const float* x1, long n1, const float* shift1, const float* scale1,
                                     const float* x2, long n2, const float* shift2, const float* scale2,
                                     const float* x3, long n3, const float* shift3, const float* scale3,
                                     const float* x4, long n4, const float* shift4, const float* scale4,
                                     const float* x5, long n5, const float* shift5, const float* scale5, float* out,
                                     void* stream) {
  if (count < 1 || count > 6 || !out) return DV3_ERR_ARG;
  StatsJobs j;
  const float* xs[6] = {x0, x1, x2, x3, x4, x5};
  const long ns[6] = {n0, n1, n2, n3, n4, n5};
  const float* sh[6] = {shift0, shift1, shift2, shift3, shift4, shift5};
  const float* sc[6] = {scale0, scale1, scale2, scale3, scale4, scale5};
  for (int i = 0; i < 6; ++i) {
    if (i < count && (!xs[i] || ns[i] <= 0)) return DV3_ERR_ARG;
    j.x[i] = xs[i]; j.n[i] = ns[i]; j.shift[i] = sh[i]; j.scale[i] = sc[i];
  }
  hipLaunchKernelGGL(tensorstats_multi_kernel, dim3(count), dim3(1024), 0, (hipStream_t)stream, j, out);
  return (int)hipGetLastError();
}

extern "C" int dv3_version(void) { return 3; }
extern "C" int dv3_dev_switches(void) {
#ifdef DV3_DEV_SWITCHES
  return 1;
#else
  return 0;
#endif
}
