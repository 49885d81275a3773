// Shared device helpers and error codes for libdv3hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DV3_OK 0
#define DV3_ERR_ARG 10001  // bad shape / pointer / unsupported size; nothing was launched

// Kernel-selection switches for A/B measurements (tools/*_bench.py) exist only in a development build
// (`build.py --dev` defines DV3_DEV_SWITCHES); the shipped library reads no environment variable.
#ifdef DV3_DEV_SWITCHES
#include <stdlib.h>
#define DV3_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define DV3_ENV_INT(name, dflt) (dflt)
#endif

// hipFuncSetAttribute (dynamic LDS above 64 KB) is a PER-DEVICE setting: a launcher remembers per device, not per
// process, that it has made the call -- a process that drives a second GPU gets the attribute there too.
// `seen`: a zero-initialised static of the call site (256 device bits).
static inline bool dv3_first_on_device(unsigned long long (&seen)[4]) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 256) return true;
  const unsigned long long bit = 1ull << (d & 63);
  if (seen[d >> 6] & bit) return false;
  seen[d >> 6] |= bit;
  return true;
}

namespace dv3 {

constexpr float kLnEps = 1e-3f;  // every LayerNorm on the path: networks.py:55,66,75,631,754,802

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// d/dx silu(x) = s + x s (1 - s)
__device__ __forceinline__ float dsiluf_(float x) {
  const float s = sigmoidf_(x);
  return s * (1.f + x * (1.f - s));
}

// reductions over a power-of-two group of G consecutive lanes (G <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// One-hot categorical with uniform mixing (tools.py:436-442), one class per lane in a group of G lanes:
// p_hat = (1-u) softmax(l) + u/D for the lane's class; also returns the plain softmax value
template <int G>
__device__ __forceinline__ void unimix_probs(float l, bool valid, int D, float unimix, float& sm, float& ph) {
  const float m = group_max<G>(valid ? l : -INFINITY);
  const float e = valid ? expf(l - m) : 0.f;
  const float s = group_sum<G>(e);
  sm = e / s;
  ph = valid ? sm * (1.f - unimix) + unimix / (float)D : 0.f;
}

// block-wide sum for 256-thread workgroups (4 waves); `red` is >= 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = group_sum<64>(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Philox4x32-10 counter RNG (used when the caller passes no explicit noise tensor)
struct Philox {
  uint32_t k0, k1;
  __device__ __forceinline__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  __device__ __forceinline__ void operator()(uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) const {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n1 = (uint32_t)p1;
      const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b, n3 = (uint32_t)p0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};
// uniform in (0,1]
__device__ __forceinline__ float u01(uint32_t x) { return ((x >> 8) + 1) * (1.0f / 16777216.0f); }

}  // namespace dv3
