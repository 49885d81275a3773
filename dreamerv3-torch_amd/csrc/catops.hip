// One-hot categorical latents with 1% uniform mixing: sampling, straight-through backward, KL and
// entropy.  Reference: tools.OneHotDist (tools.py:436-460) as used by RSSM.get_dist
// (networks.py:161-166), RSSM.kl_loss (networks.py:272-290) and the 'onehot' actor
// (networks.py:713-714).
//
// A categorical group has D classes (32 for the RSSM, num_actions for the discrete actor) and is
// owned by G = pow2 >= D consecutive lanes, one class per lane; 64/G groups per wave.
#include "dv3_common.h"

namespace dv3 {

// sample: idx = argmax_d p_hat[d] / q[d], q ~ Exp(1) (the single-draw path of torch.multinomial);
// mode: idx = argmax_d p_hat[d].  Ties resolve to the lowest class index, as torch.argmax does.
// forced (optional, [R]): the class to emit instead of the kernel's own draw; flips counts the disagreements.
template <int G>
__global__ __launch_bounds__(256) void onehot_sample_kernel(const float* __restrict__ logit,
                                                            const float* __restrict__ noise,
                                                            const unsigned long long* __restrict__ rng_state,
                                                            float* __restrict__ out, int* __restrict__ idx_out,
                                                            const int* __restrict__ forced,
                                                            unsigned int* __restrict__ flips,
                                                            long R, int D, float unimix, int mode,
                                                            unsigned long long offset_add,
                                                            const float* __restrict__ nb_first,
                                                            const float* __restrict__ nb_init,
                                                            float* __restrict__ nb_out, int nb_groups,
                                                            const int* __restrict__ nb_init_idx,
                                                            int* __restrict__ nb_idx_out) {
  constexpr int GPB = 256 / G;
  const int sub = threadIdx.x / G, d = threadIdx.x % G;
  const bool valid = d < D;
  unsigned long long seed = 0, offset = 0;
  if (!mode && !noise) {
    seed = rng_state[0];
    offset = rng_state[1] + offset_add;
  }
  for (long r0 = (long)blockIdx.x * GPB; r0 < R; r0 += (long)gridDim.x * GPB) {
    const long r = r0 + sub;
    const bool rv = r < R;
    const float l = (rv && valid) ? logit[r * D + d] : 0.f;
    float sm, ph;
    unimix_probs<G>(l, valid, D, unimix, sm, ph);
    float score = ph;
    if (!mode) {
      float q;
      if (noise) {
        q = (rv && valid) ? noise[r * D + d] : 1.f;
      } else {
        uint32_t o[4];
        const unsigned long long e = (unsigned long long)r * D + d;
        Philox ph4(seed);
        ph4(offset + (e >> 2), 0x5eedULL, o);
        q = -logf(u01(o[e & 3]));
        q = fmaxf(q, 1e-30f);
      }
      score = ph / q;
    }
    if (!valid) score = -INFINITY;
    // argmax with lowest-index tie-break
    float best = score;
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
    if (forced) {
      // teacher forcing (parity tests): the state follows the given draw; a draw of this kernel that differs is
      // counted (argmax(p/q) flips when two ratios are within an ulp)
      const int f = rv ? forced[r] : 0;
      if (rv && d == 0 && flips && f != bi) atomicAdd(flips, 1u);
      bi = f;
    }
    if (rv && valid) {
      const float v = (d == bi) ? 1.f : 0.f;
      out[r * D + d] = v;
      if (nb_out) {  // next observe step's reset blend of this sample: row b = r / groups, init [groups][D]
        const float m = nb_first[r / nb_groups];
        nb_out[r * D + d] = v * (1.f - m) + nb_init[(r % nb_groups) * D + d] * m;
      }
    }
    if (rv && d == 0 && idx_out) idx_out[r] = bi;
    // class index of the blended next-step state (a one-hot either way: the sample or the initial state's mode)
    if (rv && d == 0 && nb_idx_out) nb_idx_out[r] = (nb_first[r / nb_groups] != 0.f) ? nb_init_idx[r % nb_groups] : bi;
  }
}

// Straight-through backward.  sample: forward = onehot + (p - sg(p))  -> t = g.
// mode: forward = onehot + (log p - sg(log p))                        -> t = g / p_hat.
// dlogit_k (+)= (1-u) sm_k (t_k - sum_j sm_j t_j)      (normalisation constants cancel)
template <int G>
__global__ __launch_bounds__(256) void onehot_st_bwd_kernel(const float* __restrict__ logit,
                                                            const float* __restrict__ g, float* __restrict__ dlogit,
                                                            long R, int D, float unimix, int mode, int accumulate) {
  constexpr int GPB = 256 / G;
  const int sub = threadIdx.x / G, d = threadIdx.x % G;
  const bool valid = d < D;
  for (long r0 = (long)blockIdx.x * GPB; r0 < R; r0 += (long)gridDim.x * GPB) {
    const long r = r0 + sub;
    const bool rv = r < R;
    const float l = (rv && valid) ? logit[r * D + d] : 0.f;
    float sm, ph;
    unimix_probs<G>(l, valid, D, unimix, sm, ph);
    float t = (rv && valid) ? g[r * D + d] : 0.f;
    if (mode && valid) t = t / ph;
    const float dot = group_sum<G>(sm * t);
    if (rv && valid) {
      const float v = (1.f - unimix) * sm * (t - dot);
      float* o = dlogit + r * D + d;
      *o = accumulate ? (*o + v) : v;
    }
  }
}

// Reverse observe scan, the two row operations between step t's last data-gradient GEMM and step t-1's first one in ONE
// launch (obs_blend_bwd + onehot_st_bwd):
//   carry   gs_prev += dsin (1 - m),  gd_prev += ddin (1 - m),  dstoch0 / ddeter0 += sum_b (.) m   (networks.py:183-191)
//   ST      dpost_logit_prev += (1-u) sm (t - sum_j sm_j t_j),  t = the just-updated gs_prev      (tools.py:452-460)
// Blocks [0, nb_s) take the B*S categorical groups (one class per lane), blocks [nb_s, ...) the B*De deter elements.
template <int G>
__global__ __launch_bounds__(256) void obs_carry_st_bwd_kernel(const float* __restrict__ dsin, long ld_dsin,
                                                               const float* __restrict__ ddin, long ld_ddin,
                                                               const float* __restrict__ first,
                                                               float* __restrict__ gs_prev, float* __restrict__ gd_prev,
                                                               float* __restrict__ ds0, float* __restrict__ dd0,
                                                               const float* __restrict__ logit_prev,
                                                               float* __restrict__ dlogit_prev, int B, int S, int D,
                                                               int De, float unimix, int nb_s) {
  if ((int)blockIdx.x < nb_s) {
    constexpr int GPB = 256 / G;
    const int sub = threadIdx.x / G, d = threadIdx.x % G;
    const bool valid = d < D;
    const long R = (long)B * S;
    const int SD = S * D;
    for (long r0 = (long)blockIdx.x * GPB; r0 < R; r0 += (long)nb_s * GPB) {
      const long r = r0 + sub;
      const bool rv = r < R;
      const int b = rv ? (int)(r / S) : 0, sg = rv ? (int)(r % S) : 0;
      const int j = sg * D + d;
      float t = 0.f;
      if (rv && valid) {
        const float m = first[b];
        const float g = dsin[(long)b * ld_dsin + j];
        t = gs_prev[(long)b * SD + j] + g * (1.f - m);
        gs_prev[(long)b * SD + j] = t;
        if (m != 0.f) atomicAdd(ds0 + j, g * m);
      }
      const float l = (rv && valid) ? logit_prev[r * D + d] : 0.f;
      float sm, ph;
      unimix_probs<G>(l, valid, D, unimix, sm, ph);
      const float dot = group_sum<G>(sm * t);
      if (rv && valid) dlogit_prev[r * D + d] += (1.f - unimix) * sm * (t - dot);
    }
  } else {
    const long total = (long)B * De;
    const long nb_d = gridDim.x - nb_s;
    for (long e = (long)(blockIdx.x - nb_s) * 256 + threadIdx.x; e < total; e += nb_d * 256) {
      const int b = (int)(e / De), k = (int)(e % De);
      const float m = first[b];
      const float g = ddin[(long)b * ld_ddin + k];
      gd_prev[e] += g * (1.f - m);
      if (m != 0.f) atomicAdd(dd0 + k, g * m);
    }
  }
}

// KL(post || prior) summed over the S groups of one state row, plus both entropies.
//   kl = sum p (log p - log q),  ent = -sum p log p      (p, q unimixed)
template <int G>
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ post, const float* __restrict__ prior,
                                                     float* __restrict__ kl, float* __restrict__ ent_post,
                                                     float* __restrict__ ent_prior, long rows, int S, int D,
                                                     float unimix) {
  constexpr int GPW = 64 / G;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / G, d = lane % G;
  const bool valid = d < D;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float a_kl = 0.f, a_ep = 0.f, a_eq = 0.f;
    for (int s0 = 0; s0 < S; s0 += GPW) {
      const int sg = s0 + sub;
      const bool gv = sg < S;
      const long off = (row * S + sg) * D + d;
      const float lp = (gv && valid) ? post[off] : 0.f;
      const float lq = (gv && valid) ? prior[off] : 0.f;
      float smp, pp, smq, pq;
      unimix_probs<G>(lp, valid, D, unimix, smp, pp);
      unimix_probs<G>(lq, valid, D, unimix, smq, pq);
      if (gv && valid) {
        const float lgp = logf(pp), lgq = logf(pq);
        a_kl += pp * (lgp - lgq);
        a_ep -= pp * lgp;
        a_eq -= pq * lgq;
      }
    }
    a_kl = group_sum<64>(a_kl);
    a_ep = group_sum<64>(a_ep);
    a_eq = group_sum<64>(a_eq);
    if (lane == 0) {
      kl[row] = a_kl;
      if (ent_post) ent_post[row] = a_ep;
      if (ent_prior) ent_prior[row] = a_eq;
    }
  }
}

// loss_row = dyn_scale*max(KL(sg(post)||prior), free) + rep_scale*max(KL(post||sg(prior)), free)
// d/dpost  = rep_scale * up * (1-u) sm_p (d - sum sm_p d),  d = log p - log q
// d/dprior = dyn_scale * up * (1-u) sm_q (h - sum sm_q h),  h = -p/q
// gradient passes the clip where kl >= free.  `up` is a scalar (the 1/(B*T) of torch.mean).
template <int G>
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ post, const float* __restrict__ prior,
                                                     const float* __restrict__ kl, float* __restrict__ dpost,
                                                     float* __restrict__ dprior, long rows, int S, int D,
                                                     float unimix, float free_nats, float dyn_scale,
                                                     float rep_scale, float up, int acc_post, int acc_prior) {
  constexpr int GPW = 64 / G;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / G, d = lane % G;
  const bool valid = d < D;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float pass = (kl[row] >= free_nats) ? up : 0.f;
    for (int s0 = 0; s0 < S; s0 += GPW) {
      const int sg = s0 + sub;
      const bool gv = sg < S;
      const long off = (row * S + sg) * D + d;
      const float lp = (gv && valid) ? post[off] : 0.f;
      const float lq = (gv && valid) ? prior[off] : 0.f;
      float smp, pp, smq, pq;
      unimix_probs<G>(lp, valid, D, unimix, smp, pp);
      unimix_probs<G>(lq, valid, D, unimix, smq, pq);
      const float dd = valid ? (logf(pp) - logf(pq)) : 0.f;
      const float hh = valid ? (-pp / pq) : 0.f;
      const float dotp = group_sum<G>(smp * dd);
      const float dotq = group_sum<G>(smq * hh);
      if (gv && valid) {
        const float gp = rep_scale * pass * (1.f - unimix) * smp * (dd - dotp);
        const float gq = dyn_scale * pass * (1.f - unimix) * smq * (hh - dotq);
        dpost[off] = acc_post ? dpost[off] + gp : gp;
        dprior[off] = acc_prior ? dprior[off] + gq : gq;
      }
    }
  }
}

// per-group entropy / log-prob of a one-hot action (discrete actor)
template <int G>
__global__ __launch_bounds__(256) void onehot_ent_logp_kernel(const float* __restrict__ logit,
                                                              const float* __restrict__ x, float* __restrict__ ent,
                                                              float* __restrict__ logp, long R, int D,
                                                              float unimix) {
  constexpr int GPB = 256 / G;
  const int sub = threadIdx.x / G, d = threadIdx.x % G;
  const bool valid = d < D;
  for (long r0 = (long)blockIdx.x * GPB; r0 < R; r0 += (long)gridDim.x * GPB) {
    const long r = r0 + sub;
    const bool rv = r < R;
    const float l = (rv && valid) ? logit[r * D + d] : 0.f;
    float sm, ph;
    unimix_probs<G>(l, valid, D, unimix, sm, ph);
    const float lg = valid ? logf(ph) : 0.f;
    const float e = group_sum<G>(valid ? -ph * lg : 0.f);
    const float xv = (rv && valid && x) ? x[r * D + d] : 0.f;
    const float lp = group_sum<G>(xv * lg);
    if (rv && d == 0) {
      if (ent) ent[r] = e;
      if (logp) logp[r] = lp;
    }
  }
}

// backward of entropy and log-prob w.r.t. the logits:
//   dH/dp_j = -(log p_j + 1);  dlogp/dp_j = x_j / p_j;  through p = (1-u) sm + u/D
template <int G>
__global__ __launch_bounds__(256) void onehot_ent_logp_bwd_kernel(const float* __restrict__ logit,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ dent,
                                                                  const float* __restrict__ dlogp,
                                                                  float* __restrict__ dlogit, long R, int D,
                                                                  float unimix, int accumulate) {
  constexpr int GPB = 256 / G;
  const int sub = threadIdx.x / G, d = threadIdx.x % G;
  const bool valid = d < D;
  for (long r0 = (long)blockIdx.x * GPB; r0 < R; r0 += (long)gridDim.x * GPB) {
    const long r = r0 + sub;
    const bool rv = r < R;
    const float l = (rv && valid) ? logit[r * D + d] : 0.f;
    float sm, ph;
    unimix_probs<G>(l, valid, D, unimix, sm, ph);
    float t = 0.f;
    if (rv && valid) {
      if (dent) t += dent[r] * -(logf(ph) + 1.f);
      if (dlogp && x) t += dlogp[r] * x[r * D + d] / ph;
    }
    const float dot = group_sum<G>(sm * t);
    if (rv && valid) {
      const float v = (1.f - unimix) * sm * (t - dot);
      float* o = dlogit + r * D + d;
      *o = accumulate ? (*o + v) : v;
    }
  }
}

static int pick_g(int D) {
  int g = 4;
  while (g < 64 && g < D) g <<= 1;
  return g;
}
static unsigned cap_blocks(long n, long per_block, long cap) {
  long b = (n + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

#define DV3_G_DISPATCH(D, CALL)                   \
  switch (pick_g(D)) {                            \
    case 4: { constexpr int G = 4; CALL; } break;   \
    case 8: { constexpr int G = 8; CALL; } break;   \
    case 16: { constexpr int G = 16; CALL; } break; \
    case 32: { constexpr int G = 32; CALL; } break; \
    default: { constexpr int G = 64; CALL; } break; \
  }

}  // namespace dv3

using namespace dv3;

extern "C" int dv3_onehot_sample_fwd(const float* logit, const float* noise, const unsigned long long* rng_state,
                                     unsigned long long rng_offset, float* onehot, int* idx, long R, int D,
                                     float unimix, int mode, void* stream) {
  if (R <= 0) return 0;
  if (D <= 0 || D > 64 || !logit || !onehot) return DV3_ERR_ARG;
  if (!mode && !noise && !rng_state) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DV3_G_DISPATCH(D, hipLaunchKernelGGL((onehot_sample_kernel<G>), dim3(cap_blocks(R, 256 / G, 8192)), dim3(256), 0, s,
                                       logit, noise, rng_state, onehot, idx, (const int*)nullptr,
                                       (unsigned int*)nullptr, R, D, unimix, mode, rng_offset, nullptr, nullptr,
                                       nullptr, 1, (const int*)nullptr, (int*)nullptr));
  return (int)hipGetLastError();
}

// General form: optional class indices out (idx), optional teacher forcing (forced [R] int32: emit that class;
// flips += number of rows whose own draw differs), optional fused reset blend (next_first != NULL, as
// dv3_onehot_sample_fwd_blend).
extern "C" int dv3_onehot_sample_fwd_ex(const float* logit, const float* noise, const unsigned long long* rng_state,
                                        unsigned long long rng_offset, float* onehot, int* idx, const int* forced,
                                        unsigned int* flips, long R, int D, float unimix, int mode,
                                        const float* next_first, const float* init, float* next_out, int groups,
                                        const int* init_idx, int* next_idx, void* stream) {
  if (R <= 0) return 0;
  if (D <= 0 || D > 64 || !logit || !onehot) return DV3_ERR_ARG;
  if (!mode && !noise && !rng_state) return DV3_ERR_ARG;
  if (next_first && (!init || !next_out || groups <= 0 || R % groups)) return DV3_ERR_ARG;
  if (next_idx && (!next_first || !init_idx)) return DV3_ERR_ARG;
  if (!next_first) { init = nullptr; next_out = nullptr; groups = 1; next_idx = nullptr; }
  hipStream_t s = (hipStream_t)stream;
  DV3_G_DISPATCH(D, hipLaunchKernelGGL((onehot_sample_kernel<G>), dim3(cap_blocks(R, 256 / G, 8192)), dim3(256), 0, s,
                                       logit, noise, rng_state, onehot, idx, forced, flips, R, D, unimix, mode,
                                       rng_offset, next_first, init, next_out, groups, init_idx, next_idx));
  return (int)hipGetLastError();
}

// as dv3_onehot_sample_fwd, plus next_out[r][d] = onehot*(1 - next_first[r / groups]) + init[r % groups][d] *
// next_first[...]: the following observe step's reset blend of the sampled state (networks.py:183-191)
extern "C" int dv3_onehot_sample_fwd_blend(const float* logit, const float* noise, const unsigned long long* rng_state,
                                           unsigned long long rng_offset, float* onehot, long R, int D, float unimix,
                                           int mode, const float* next_first, const float* init, float* next_out,
                                           int groups, void* stream) {
  if (R <= 0) return 0;
  if (D <= 0 || D > 64 || !logit || !onehot || !next_first || !init || !next_out || groups <= 0 || R % groups)
    return DV3_ERR_ARG;
  if (!mode && !noise && !rng_state) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DV3_G_DISPATCH(D, hipLaunchKernelGGL((onehot_sample_kernel<G>), dim3(cap_blocks(R, 256 / G, 8192)), dim3(256), 0, s,
                                       logit, noise, rng_state, onehot, (int*)nullptr, (const int*)nullptr,
                                       (unsigned int*)nullptr, R, D, unimix, mode, rng_offset, next_first, init,
                                       next_out, groups, (const int*)nullptr, (int*)nullptr));
  return (int)hipGetLastError();
}

extern "C" int dv3_onehot_st_bwd(const float* logit, const float* dstoch, float* dlogit, long R, int D, float unimix,
                                 int mode, int accumulate, void* stream) {
  if (R <= 0) return 0;
  if (D <= 0 || D > 64 || !logit || !dstoch || !dlogit) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DV3_G_DISPATCH(D, hipLaunchKernelGGL((onehot_st_bwd_kernel<G>), dim3(cap_blocks(R, 256 / G, 8192)), dim3(256), 0, s,
                                       logit, dstoch, dlogit, R, D, unimix, mode, accumulate));
  return (int)hipGetLastError();
}

extern "C" int dv3_obs_carry_st_bwd(const float* dsin, long ld_dsin, const float* ddin, long ld_ddin,
                                    const float* is_first, float* gs_prev, float* gd_prev, float* dstoch0,
                                    float* ddeter0, const float* logit_prev, float* dlogit_prev, int B, int S, int D,
                                    int De, float unimix, void* stream) {
  if (B <= 0) return 0;
  if (!dsin || !ddin || !is_first || !gs_prev || !gd_prev || !dstoch0 || !ddeter0 || !logit_prev || !dlogit_prev ||
      S <= 0 || D <= 0 || D > 64 || De <= 0 || ld_dsin < (long)S * D || ld_ddin < De)
    return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long R = (long)B * S;
  DV3_G_DISPATCH(D, {
    const int nb_s = (int)cap_blocks(R, 256 / G, 4096);
    long nb_d = ((long)B * De + 255) / 256;
    if (nb_d > 1024) nb_d = 1024;
    hipLaunchKernelGGL((obs_carry_st_bwd_kernel<G>), dim3((unsigned)(nb_s + nb_d)), dim3(256), 0, s, dsin, ld_dsin, ddin,
                       ld_ddin, is_first, gs_prev, gd_prev, dstoch0, ddeter0, logit_prev, dlogit_prev, B, S, D, De,
                       unimix, nb_s);
  });
  return (int)hipGetLastError();
}

extern "C" int dv3_kl_fwd(const float* post_logit, const float* prior_logit, float* kl, float* ent_post,
                          float* ent_prior, long rows, int S, int D, float unimix, void* stream) {
  if (rows <= 0) return 0;
  if (D <= 0 || D > 64 || S <= 0 || !post_logit || !prior_logit || !kl) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DV3_G_DISPATCH(D, hipLaunchKernelGGL((kl_fwd_kernel<G>), dim3(cap_blocks(rows, 4, 4096)), dim3(256), 0, s, post_logit,
                                       prior_logit, kl, ent_post, ent_prior, rows, S, D, unimix));
  return (int)hipGetLastError();
}

extern "C" int dv3_kl_bwd(const float* post_logit, const float* prior_logit, const float* kl, float* dpost,
                          float* dprior, long rows, int S, int D, float unimix, float free_nats, float dyn_scale,
                          float rep_scale, float upstream, int acc_post, int acc_prior, void* stream) {
  if (rows <= 0) return 0;
  if (D <= 0 || D > 64 || S <= 0 || !post_logit || !prior_logit || !kl || !dpost || !dprior) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DV3_G_DISPATCH(D, hipLaunchKernelGGL((kl_bwd_kernel<G>), dim3(cap_blocks(rows, 4, 4096)), dim3(256), 0, s, post_logit,
                                       prior_logit, kl, dpost, dprior, rows, S, D, unimix, free_nats, dyn_scale,
                                       rep_scale, upstream, acc_post, acc_prior));
  return (int)hipGetLastError();
}

extern "C" int dv3_onehot_ent_logp_fwd(const float* logit, const float* x, float* ent, float* logp, long R, int D,
                                       float unimix, void* stream) {
  if (R <= 0) return 0;
  if (D <= 0 || D > 64 || !logit) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DV3_G_DISPATCH(D, hipLaunchKernelGGL((onehot_ent_logp_kernel<G>), dim3(cap_blocks(R, 256 / G, 8192)), dim3(256), 0, s,
                                       logit, x, ent, logp, R, D, unimix));
  return (int)hipGetLastError();
}

extern "C" int dv3_onehot_ent_logp_bwd(const float* logit, const float* x, const float* dent, const float* dlogp,
                                       float* dlogit, long R, int D, float unimix, int accumulate, void* stream) {
  if (R <= 0) return 0;
  if (D <= 0 || D > 64 || !logit || !dlogit) return DV3_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  DV3_G_DISPATCH(D, hipLaunchKernelGGL((onehot_ent_logp_bwd_kernel<G>), dim3(cap_blocks(R, 256 / G, 8192)), dim3(256), 0,
                                       s, logit, x, dent, dlogp, dlogit, R, D, unimix, accumulate));
  return (int)hipGetLastError();
}
