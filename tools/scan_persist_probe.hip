// Bounded experiment (VERDICT r03 "Next" #6): what does ONE persistent launch of the forward observe scan cost per
// step on MI355X, with the weights resident in LDS and a device-wide hand-off between layers?  Not part of libdv3hip:
// tools/scan_persist_probe.py builds this file on its own and times it.
//
// The skeleton has the scan's data flow and nothing else: per step PH dependent phases; in a phase every workgroup
//   (1) reads the FULL 16 x K input vector the previous phase produced (other CUs wrote it: sc1 loads, served by L2),
//   (2) multiplies it with its own slice of the layer's weights, which sits in LDS for the whole launch
//       (16 output columns per workgroup, K split over the 4 waves, v_mfma_f32_16x16x4_f32, LDS reduce),
//   (3) writes its 16 x cols slice of the output vector (sc1 stores, vmcnt(0)), and
//   (4) arrives at a device-wide barrier (one monotonic counter per launch, or per-XCD counters + a top counter).
// The row operations of the real scan (LayerNorm, gates, sampling) are left out: they add work, never remove a seam, so
// the time measured here is a floor for the persistent form.  verify=1 makes every phase write values the readers can
// check (a stale or early read is counted): the protocol is exercised, not only timed.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ProbeParams {
  float* act[2];        // ping-pong activation vectors [16][KMAX]
  unsigned* counters;   // [0] flat barrier counter, [1] error flag, [2] stale-read count, [8..15] per-XCD, [16] top, [24] generation
  int steps, phases, G, K, cols, mode, verify;
  int variant;          // hand-off form: 0 sc1 loads; 1 sc0 sc1 loads; 2 acquire fence + plain loads; 3 release + acquire fences, plain stores / loads
  int nbuf;             // activation buffers in rotation (2: ping-pong; phases * steps + 1: every phase reads fresh addresses)
  float* ring;          // [nbuf][16][K] when nbuf > 2
  const float* wsrc;    // [16][K] weights every workgroup copies into LDS (values irrelevant)
  long spin_limit;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// device-wide barrier on a monotonic counter: every workgroup adds 1 after its stores have drained; everybody polls.
// Bounded: a poll loop that runs out sets the error flag, and every later barrier returns at once.
__device__ __forceinline__ bool barrier_flat(unsigned* counters, unsigned target, long limit, int variant) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (variant == 3) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (every byte handed off was stored sc1 and has drained -- vmcnt(0) in front of the barrier -- so the arrival itself
    // needs no release fence: MI355X_MICROARCH.md, hand-offs with sc1 loads in place of the acquire)
    __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long n = 0;
    while (ld_sc1(&counters[0]) < target) {
      if (++n > limit || ld_sc1(&counters[1]) != 0) {
        __hip_atomic_store(&counters[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (variant >= 2) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  return ok;
}

// XCD-hierarchical: arrivals go to the counter of the workgroup's XCD; the last arriver of an XCD adds to the top
// counter; the last XCD publishes the generation; everybody polls the generation word.
__device__ __forceinline__ bool barrier_xcd(unsigned* counters, unsigned gen, int G, int xcd, int per_xcd, long limit,
                                            int variant) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (variant == 3) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned a = __hip_atomic_fetch_add(&counters[8 + xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == (unsigned)per_xcd * gen) {
      const unsigned t = __hip_atomic_fetch_add(&counters[16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned nx = (unsigned)((G + per_xcd - 1) / per_xcd);
      if (t + 1 == nx * gen) __hip_atomic_store(&counters[24], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    long n = 0;
    while (ld_sc1(&counters[24]) < gen) {
      if (++n > limit || ld_sc1(&counters[1]) != 0) {
        __hip_atomic_store(&counters[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (variant >= 2) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  return ok;
}

extern "C" __global__ __launch_bounds__(256) void persist_scan_probe(ProbeParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // weights [16][K + 8] | reduce [4][256]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int LDW = p.K + 8;
  float* red = lds + 16 * LDW;
  for (int e = tid; e < 16 * p.K; e += 256) lds[(e / p.K) * LDW + (e % p.K)] = p.wsrc[e];
  __syncthreads();
  int xcd = 0, per_xcd = 1;
  if (p.mode == 1) {
    // (blocks are dealt round-robin over the XCDs: blockIdx % 8 names the XCD group; placement is speed only)
    xcd = blockIdx.x & 7;
    per_xcd = (p.G + 7) / 8;
  }
  const int kq = p.K / 4;          // K share of a wave
  const int chunks = kq / 16;      // 16-k chunks per wave
  unsigned gen = 0;
  bool alive = true;
  for (int s = 0; s < p.steps && alive; ++s) {
    for (int ph = 0; ph < p.phases && alive; ++ph) {
      const int n = s * p.phases + ph;
      const float* in = p.nbuf > 2 ? p.ring + (long)(n % p.nbuf) * 16 * p.K : p.act[n & 1];
      float* out = p.nbuf > 2 ? p.ring + (long)((n + 1) % p.nbuf) * 16 * p.K : p.act[(n + 1) & 1];
      // (1) the wave's K share of the 16 input rows: row i, k = wave*kq + 16c + 4q .. +3  (sc1: never from this CU's L1)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* arow = in + (long)i * p.K + wave * kq + 4 * q;
      f32x4 a[8] = {};
      unsigned stale = 0;
      for (int c0 = 0; c0 < chunks; c0 += 8) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c0 + c < chunks) {
            const float* src = arow + 16 * (c0 + c);
            if (p.variant == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(a[c]) : "v"(src) : "memory");
            else if (p.variant == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(a[c]) : "v"(src) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[c]) : "v"(src) : "memory");
          }
        // (the loads above are invisible to the compiler's wait-count pass: wait for all of them, and tie the wait to
        // the registers so that no use is scheduled in front of it)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     :
                     : "memory");
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c0 + c < chunks) {
            if (p.verify && (s > 0 || ph > 0)) {
              const float want = (float)((s * p.phases + ph) & 1023);
#pragma unroll
              for (int e = 0; e < 4; ++e) stale += a[c][e] != want;
            }
            const f32x4 b = *reinterpret_cast<const f32x4*>(&lds[i * LDW + wave * kq + 16 * (c0 + c) + 4 * q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][e], b[e], acc, 0, 0, 0);
          }
      }
      if (stale) atomicAdd(&p.counters[2], stale);
      // (2) reduce the four waves' partial 16 x 16 tiles
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * q + r) * 16 + i] = acc[r];
      __syncthreads();
      // (3) this workgroup's slice of the output vector: rows 0..15, columns [blockIdx * cols, +cols)
      if (tid < 16 * p.cols) {
        const int r = tid / p.cols, c = tid % p.cols;
        float v = red[r * 16 + c] + red[256 + r * 16 + c] + red[512 + r * 16 + c] + red[768 + r * 16 + c];
        if (p.verify) v = (float)((s * p.phases + ph + 1) & 1023);
        const long col = (long)blockIdx.x * p.cols + c;
        if (col < p.K) {
          if (p.variant == 3) out[(long)r * p.K + col] = v;
          else __hip_atomic_store(out + (long)r * p.K + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // (4) the seam
      ++gen;
      alive = p.mode == 1 ? barrier_xcd(p.counters, gen, p.G, xcd, per_xcd, p.spin_limit, p.variant)
                          : barrier_flat(p.counters, gen * (unsigned)p.G, p.spin_limit, p.variant);
    }
  }
}

extern "C" int probe_launch(ProbeParams* p, void* stream) {
  const size_t ldsb = (size_t)(16 * (p->K + 8) + 4 * 256) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)persist_scan_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL(persist_scan_probe, dim3(p->G), dim3(256), ldsb, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
