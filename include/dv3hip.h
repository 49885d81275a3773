/* libdv3hip -- C ABI of the MI355X-native (gfx950) DreamerV3 world-model training hot path.
 *
 * The reference (ChenFengTsai/dreamerv3-torch) is pure Python on PyTorch: it has no FFI, plugin or
 * operator registry.  The "interface each entry point replaces" is therefore the ATen op sequence
 * a reference Python method issues; every declaration below cites that method (file:line in the
 * reference checkout).  The Python class surface that sits on top of this ABI (networks.RSSM,
 * models.WorldModel, ...) mirrors the reference's names and signatures; see INTEGRATION.md.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (fp32 unless typed otherwise), caller-allocated; no entry
 *     point allocates, frees, synchronises or keeps state between calls.
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); every launch is
 *     asynchronous on it and is hipGraph-capturable.
 *   - Return value: 0 on success, DV3_ERR_ARG (10001) when an argument is rejected before any
 *     launch, otherwise the hipError_t of the failed launch.  Nothing throws across the ABI.
 *   - Row-major everywhere; `ld*` are leading dimensions in elements.
 *   - "accumulate" flags: 0 = overwrite the output, 1 = add into it.
 */
#ifndef DV3HIP_H_
#define DV3HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define DV3_ERR_ARG 10001

int dv3_version(void);
/* 1 when the library was built with -DDV3_DEV_SWITCHES (`build.py --dev`): only then do the DV3_* A/B environment
 * switches of the launchers (and of the Python host code, dv3hip/_dev.py) have any effect.  The shipped build
 * returns 0 and reads no environment variable. */
int dv3_dev_switches(void);

/* ---- dense layers -------------------------------------------------------------------------------
 * C[M,N] (+)= [A | A2][M,K] * op(B)[K,N] + bias[N] on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
 *   transA=0: A[m*lda+k]   transA=1: A[k*lda+m]      transB=1: B[n*ldb+k]   transB=0: B[k*ldb+n]
 *   A2/K1: optional second K segment (columns K1..K-1 come from A2[m*lda2 + k-K1]); K1 % 32 == 0,
 *          transA must be 0.  Pass A2=NULL for a single operand.
 *   tile: -1 = choose, 0 = 128x128, 1 = 64x64, 2 = 32x128, 3 = skinny (M <= 32, transA = 0: one 16-column
 *         tile per workgroup, K split over its waves, operands straight to the 16x16x4 MFMA registers),
 *         4 = 128x128x32, 5 = 64x64x64, 6 = 32x64x64 with K split inside the workgroup, 7 = narrow output
 *         (N <= 32, transA=0, transB=1, no A2: the skinny kernel on the transposed problem), 8 = 32x32x64 with
 *         K split over all four waves (two workgroups per CU when M*N is ~512k), 9 = register-direct (transA=0:
 *         32 x 64 outputs per workgroup, operands straight from global memory to 16x16x4 MFMA registers, K split
 *         over the waves; the default for <= 2M outputs), 10 = register-direct weight gradient (transA=1,
 *         transB=0, no A2 / bias; K split over workgroups when accumulating).
 *   accumulate: 0 = overwrite C, 1 = C += product, 2 = C += product with the summation order left free: the
 *         skinny kernel then also splits K over workgroups and adds the partial tiles with atomics (fills the
 *         chip at M <= 32).  The tile kernels treat 1 and 2 alike (they split long-K accumulating products,
 *         i.e. weight gradients, atomically in either case).
 * Replaces nn.Linear forward / its autograd transposes in RSSM.img_step, obs_step
 * (networks.py:195-233), GRUCell.forward (networks.py:762), MLP.forward (networks.py:657-681),
 * ConvDecoder._linear_layer (networks.py:569), and the torch.cat in front of them
 * (networks.py:154-159, 196, 216, 762). */
int dv3_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* A2,
                 long lda2, int K1, const float* B, long ldb, float* C, long ldc, const float* bias,
                 int accumulate, int tile, void* stream);

/* y = A B^T (A [M,K], B [N,K], both k-contiguous; K % 32 == 0, 16-byte aligned rows) with the output columns split
 * over two destinations: [0, nsplit) -> C (ldc, accumulate), [nsplit, N) -> C2 (ldc2, accumulate2); nsplit % 16 == 0.
 * One launch for the two data gradients of GRUCell's Linear over cat([x, h]) (networks.py:760-763): B = W^T,
 * C = dx, C2 = dh (accumulating into the gradient the successor state already carries). */
int dv3_gemm_split_f32(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                       int accumulate, float* C2, long ldc2, int nsplit, int accumulate2, void* stream);

/* n <= 48 independent weight gradients C_g (+)= A_g^T B_g in ONE grid: A_g [K_g, M_g] (row stride lda[g]), B_g
 * [K_g, N_g] (ldb[g]), C_g [M_g, N_g] (ldc[g]); 64 x 64 output tiles, every tile walks its whole K range (no split-K, no
 * atomics: reproducible sums).  The ARRAYS are host memory (n entries each), the pointers inside A / B / C device
 * memory.  The C_g must not overlap.  Replaces the per-parameter `.grad` accumulation of loss.backward() for the
 * nn.Linear weights of one backward pass (tools.py:765; networks.py:657-681 MLP layers, :195-233 RSSM layers) where the
 * reduction is the replay batch (K = B*T rows): seventeen launches of 20-60 us become one. */
int dv3_gemm_tn_grouped_f32(int n, const float* const* A, const long* lda, const float* const* B, const long* ldb,
                            float* const* C, const long* ldc, const int* M, const int* N, const int* K,
                            const int* accumulate, void* stream);

/* dv3_gemm_f32 (transA=0, transB=1, register-direct kernel) with the one-hot categorical sampling of its output fused
 * into the epilogue: C [M,N] = the logits of N/32 groups of 32 classes per row (N % 64 == 0), sampled exactly as
 * dv3_onehot_sample_fwd_ex would (same noise layout [M*N/32, 32], same Philox counters).  Replaces
 * RSSM._suff_stats_layer + get_dist(...).sample() in img_step / obs_step (networks.py:226-229, 241-250). */
int dv3_gemm_sample_f32(int M, int N, int K, const float* A, long lda, const float* A2, long lda2, int K1,
                        const float* B, long ldb, float* C, long ldc, const float* bias, const float* noise,
                        const unsigned long long* rng_state, unsigned long long rng_offset, float* onehot, int* idx,
                        const int* forced, unsigned int* flips, float unimix, int mode, const float* ln_gamma,
                        const float* ln_beta, float* ln_mean, float* ln_rstd, void* stream);
/* ln_gamma != NULL: A holds pre-activations and SiLU(LayerNorm(A) * gamma + beta) (eps 1e-3, over the K columns) is
 * applied while the operand is loaded -- the `_img_out_layers` LayerNorm + SiLU of networks.py:62-69 without a launch
 * of its own; ln_mean / ln_rstd [M] (optional) receive the row statistics for dv3_ln_act_bwd.  No A2, K % 4 == 0. */

/* ---- LayerNorm(eps 1e-3) [+ SiLU] ----------------------------------------------------------------
 * y = act(LN(x) * gamma + beta), rows of length N <= 2048; mean/rstd [R] are saved for the backward
 * (may be NULL in fwd).  act: 0 none, 1 SiLU.  chw_group G > 0 addresses y (fwd) / dy (bwd) as the
 * (C,H,W) flatten of G-pixel images: element (r,c) at (r/G)*N*G + c*G + r%G (networks.py:494).
 * Replaces nn.LayerNorm + SiLU (networks.py:55-56, 66-67, 75-76, 631-633) and ImgChLayerNorm + SiLU
 * (networks.py:476-477, 552-554, 801-810).  bwd ACCUMULATES dgamma/dbeta (both or neither NULL). */
int dv3_ln_act_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy,
                   float* mean, float* rstd, long R, int N, int act, int chw_group, void* stream);
int dv3_ln_act_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma, const float* beta,
                   const float* mean, const float* rstd, float* dx, long lddx, float* dgamma, float* dbeta,
                   long R, int N, int act, int chw_group, int accumulate_dx, void* stream);
/* dv3_ln_act_bwd with a workspace for the parameter gradients (two launches): ws (device, >= 4096 N floats, contents
 * irrelevant) receives one row of 2 N column sums per row block, written with plain stores; a second, column-parallel
 * launch adds the rows up into dgamma / dbeta (+=).  Without it every row block adds its sums onto the same N addresses:
 * up to 2048 blocks x 2 N device-scope atomics on a few cache lines, +18-40 us per launch on the conv stacks' channel
 * LayerNorms (37 us without the parameter gradients at 262 k x 64).  One workspace per stream (launches that may overlap
 * must not share it).  Same reference seams as dv3_ln_act_bwd: the .grad accumulation of nn.LayerNorm / ImgChLayerNorm
 * weights under loss.backward() (tools.py:765; networks.py:631-633, 801-810). */
int dv3_ln_act_bwd_ws(const float* dy, long lddy, const float* x, long ldx, const float* gamma, const float* beta,
                      const float* mean, const float* rstd, float* dx, long lddx, float* dgamma, float* dbeta,
                      long R, int N, int act, int chw_group, int accumulate_dx, float* ws, long ws_floats,
                      void* stream);

/* ---- LayerNorm-GRU gates -- GRUCell.forward (networks.py:760-768) ---------------------------------
 * p [M,3*De] is the output of the fused Linear on cat[x,h]; LN over all 3*De, then
 * r=sigmoid, c=tanh(r*c), u=sigmoid(u-1), h' = u*c + (1-u)*h.  Any De forward; backward De <= 2133 or
 * De in {2048, 3072, 4096} (wide cells: a workgroup per row, LayerNorm rows up to 12288). */
int dv3_gru_fwd(const float* p, long ldp, const float* gamma, const float* beta, const float* h, long ldh,
                float* h_new, long ldhn, float* mean, float* rstd, int M, int De, void* stream);
/* dv3_gru_fwd plus the NEXT observe step's reset blend of the new state (networks.py:183-191), fused:
 * next_out[r] = h_new[r]*(1 - next_first[r]) + init*next_first[r].  De % 256 == 0 and De <= 1024, or
 * De in {2048, 3072, 4096}. */
int dv3_gru_fwd_blend(const float* p, long ldp, const float* gamma, const float* beta, const float* h, long ldh,
                      float* h_new, long ldhn, float* mean, float* rstd, int M, int De, const float* next_first,
                      const float* init, float* next_out, long ld_next, void* stream);
int dv3_gru_bwd(const float* dh_new, long lddhn, const float* p, long ldp, const float* gamma, const float* beta,
                const float* h, long ldh, const float* mean, const float* rstd, float* dp, long lddp, float* dh,
                long lddh, float* dgamma, float* dbeta, int M, int De, int accumulate_dh, void* stream);

/* ---- one-hot categorical with unimix -- tools.OneHotDist (tools.py:436-460) -----------------------
 * R groups of D <= 64 classes.  sample: onehot(argmax p_hat/q), q = noise[R,D] ~ Exp(1) when given,
 * else Philox draws from rng_state = {seed, offset} (device memory).  mode=1: onehot(argmax p_hat).
 * idx (optional) receives the chosen class.  st_bwd is the straight-through gradient
 * (tools.py:446-450, 456-459): dlogit (+)= J^T dstoch.  Used by RSSM.get_dist/get_stoch
 * (networks.py:161-166, 235-239) and the discrete actor (networks.py:713-714). */
int dv3_onehot_sample_fwd(const float* logit, const float* noise, const unsigned long long* rng_state,
                          unsigned long long rng_offset, float* onehot, int* idx, long R, int D, float unimix,
                          int mode, void* stream);
int dv3_onehot_sample_fwd_blend(const float* logit, const float* noise, const unsigned long long* rng_state,
                                unsigned long long rng_offset, float* onehot, long R, int D, float unimix, int mode,
                                const float* next_first, const float* init, float* next_out, int groups,
                                void* stream);
/* ... plus next_out[r][d] = onehot*(1 - f) + init[r % groups][d]*f, f = next_first[r / groups]: the next observe
 * step's reset blend of the sampled state, fused (groups = stoch groups per batch row). */
/* General form of the two above.  forced (optional, int32 [R]): teacher forcing for the parity tests -- the class
 * emitted is forced[r]; *flips (optional, device uint32) is incremented once per row whose own draw differs
 * (argmax(p/q) can flip on an ulp).  next_first == NULL disables the fused blend. */
int dv3_onehot_sample_fwd_ex(const float* logit, const float* noise, const unsigned long long* rng_state,
                             unsigned long long rng_offset, float* onehot, int* idx, const int* forced,
                             unsigned int* flips, long R, int D, float unimix, int mode, const float* next_first,
                             const float* init, float* next_out, int groups, const int* init_idx, int* next_idx,
                             void* stream);
/* init_idx [groups] / next_idx [R] (optional, with the fused blend): class index of the blended next-step state,
 * next_idx[r] = next_first ? init_idx[r % groups] : idx[r] (the input of dv3_onehot_linear_ln_fwd at that step). */
/* rng_offset is added to rng_state's offset for this call (R*D/4+1 counters are consumed): the caller lays
 * the calls of one update out on disjoint counter ranges and advances rng_state once, so the launch
 * sequence stays static under hipGraph replay. */
int dv3_onehot_st_bwd(const float* logit, const float* dstoch, float* dlogit, long R, int D, float unimix,
                      int mode, int accumulate, void* stream);
/* entropy [R] and log-prob [R] of one-hot x (either output may be NULL), and their backward */
int dv3_onehot_ent_logp_fwd(const float* logit, const float* x, float* ent, float* logp, long R, int D,
                            float unimix, void* stream);
int dv3_onehot_ent_logp_bwd(const float* logit, const float* x, const float* dent, const float* dlogp,
                            float* dlogit, long R, int D, float unimix, int accumulate, void* stream);

/* ---- RSSM.kl_loss (networks.py:272-290) ----------------------------------------------------------
 * rows = B*T state rows of S groups x D classes.  fwd: kl[rows] = KL(post||prior) (the un-clipped
 * `value`; dyn and rep share this forward value), plus both entropies (metrics models.py:158-163).
 * bwd: gradient of  upstream * (dyn_scale*max(KL(sg(post)||prior),free) + rep_scale*max(KL(post||sg(prior)),free)). */
int dv3_kl_fwd(const float* post_logit, const float* prior_logit, float* kl, float* ent_post, float* ent_prior,
               long rows, int S, int D, float unimix, void* stream);
int dv3_kl_bwd(const float* post_logit, const float* prior_logit, const float* kl, float* dpost, float* dprior,
               long rows, int S, int D, float unimix, float free_nats, float dyn_scale, float rep_scale,
               float upstream, int acc_post, int acc_prior, void* stream);

/* ---- 255-bucket symlog two-hot head -- tools.DiscDist (tools.py:463-517) --------------------------
 * logits [R,255].  mode: symexp(sum softmax*linspace(-20,20,255)).  logprob: two-hot cross-entropy of
 * symlog(x[r]).  bwd: dlogits (+)= up[r] * d(out[r])/dlogits. */
int dv3_disc_mode_fwd(const float* logits, float* out, long R, void* stream);
int dv3_disc_mode_bwd(const float* logits, const float* up, float* dlogits, long R, int accumulate, void* stream);
int dv3_disc_logprob_fwd(const float* logits, const float* x, float* out, long R, void* stream);
int dv3_disc_logprob_bwd(const float* logits, const float* x, const float* up, float* dlogits, long R,
                         int accumulate, void* stream);

/* ---- continue head -- tools.Bernoulli.log_prob (tools.py:622-627) --------------------------------- */
int dv3_bernoulli_logprob_fwd(const float* logit, const float* x, float* out, long n, void* stream);
int dv3_bernoulli_logprob_bwd(const float* logit, const float* x, const float* up, float* dlogit, long n,
                              int accumulate, void* stream);

/* ---- image decode + reconstruction loss -----------------------------------------------------------
 * dv3_image_to_f32: u8 -> u8/255 - 0.5 (WorldModel.preprocess models.py:180 + ConvEncoder networks.py:487).
 * dv3_mse_image: loss[img] = sum_pixels (recon - u8/255)^2 (tools.MSEDist.log_prob, tools.py:531-540,
 * negated), and, when drecon != NULL, drecon = 2*upstream*(recon - u8/255) in the same pass. */
int dv3_image_to_f32(const unsigned char* image_u8, float* out, long n_images, int pixels, int perm_B, int perm_T,
                     void* stream);
int dv3_mse_image(const float* recon, const unsigned char* image_u8, float* loss, float* drecon, long n_images,
                  int pixels, float upstream, int perm_B, int perm_T, void* stream);
/* perm_B/perm_T > 0: activations are time-major -- image n' = t*B + b of out/recon pairs with replay image
 * b*T + t of image_u8 (the [B,T] -> [T,B] swap of RSSM.observe, networks.py:128-130, folded into the load). */

/* ---- small layout / reduction helpers ---------------------------------------------------------------
 * dv3_transpose01: [B,T,k] -> [T,B,k] (the same swap for action / reward / is_first / proprio keys).
 * dv3_colsum: out[N] (+)= sum over rows of x[R,N] (bias gradients of the stat/head Linears).
 * dv3_tanh_*: RSSM.initial's deter = tanh(W) (networks.py:121) and its backward (from y = tanh x). */
int dv3_transpose01(const float* x, float* y, int B, int T, int k, void* stream);
/* dv3_concat6: dst = s0 | s1 | ... | s5 (flat; n_j elements each, n_j = 0 skips): the acting step's outputs
 * (action, logprob, stoch, deter, logit: dreamer.py:183-188) packed in one launch for a single device-to-host hop. */
int dv3_concat6(const float* s0, long n0, const float* s1, long n1, const float* s2, long n2, const float* s3, long n3,
                const float* s4, long n4, const float* s5, long n5, float* dst, void* stream);
int dv3_colsum(const float* x, long ldx, float* out, long R, int N, int accumulate, void* stream);
/* n <= 48 column sums out_g[N_g] += sum_r x_g[r][:] in ONE grid (the bias gradients of a cluster of weight gradients,
 * see dv3_gemm_tn_grouped_f32).  The ARRAYS are host memory, the pointers inside x / out device memory; the out_g must
 * not overlap.  Same seam as dv3_colsum: nn.Linear bias .grad under loss.backward() (tools.py:765). */
int dv3_colsum_grouped(int n, const float* const* x, const long* ldx, float* const* out, const long* R, const int* N,
                       void* stream);
int dv3_tanh_fwd(const float* x, float* y, long n, void* stream);
int dv3_tanh_bwd(const float* y, const float* dy, float* dx, long n, int accumulate, void* stream);

/* ---- 64x64 CNN stacks as implicit GEMMs (NHWC activations) ---------------------------------------------
 * dv3_pack_conv_weight: reference weight [A][B][4][4] -> MFMA-friendly image.  transposed=0: Conv2d
 *   weight [Co][Ci][4][4] -> [Co][(ky,kx,ci)] for dv3_conv_s2_fwd.  transposed=1: ConvTranspose2d weight
 *   [Ci][Co][4][4] -> [4 parity classes][Co][(a,b,ci)] for dv3_convT_s2_fwd.  (A Conv2d weight passed with
 *   transposed=1 yields the image its dgrad needs, and vice versa: the two ops are adjoint.)
 * dv3_conv_s2_fwd : y[N,H/2,W/2,Co] = Conv2d(k4,s2,pad 1)(x[N,H,W,Ci])      ConvEncoder layers, networks.py:466-474, 771-798
 * dv3_convT_s2_fwd: y[N,2IH,2IW,Co] = ConvTranspose2d(k4,s2,p1)(x[N,IH,IW,Ci]) + bias + out_add
 *                                                                          ConvDecoder layers, networks.py:540-550, 584
 * dv3_conv_s2_wgrad: dw[Ccoarse][Cfine][4][4] += sum coarse[m,:]^T (x) gather(fine)  -- the weight gradient of
 *   either op (conv: coarse=dY, fine=x; convT: coarse=layer input, fine=dOut), in the reference layout. */
int dv3_pack_conv_weight(const float* w, float* w_packed, int Co, int Ci, int transposed, void* stream);

/*
 * dv3_im2col_s2: cols [Nimg*(H/2)*(W/2)][16*C] = the 4x4 stride-2 "same"-padded patches of x [Nimg][H][W][C] (NHWC),
 * column order (ci, ky, kx) = the Conv2d weight's own [Co][Ci][4][4] order, so the convolution of a few images is
 * y = cols * W.view(Co, 16 C)^T through dv3_gemm_f32 (ConvEncoder on the acting path, networks.py:398-440,
 * 771-798; dreamer.py:116-188).
 */
int dv3_im2col_s2(const float* x, float* cols, int Nimg, int H, int W, int C, void* stream);
int dv3_conv_s2_fwd(const float* x, const float* w_packed, float* y, int Nimg, int H, int W, int Ci, int Co,
                    int accumulate, void* stream);
int dv3_convT_s2_fwd(const float* x, const float* w_packed, const float* bias, float out_add, float* y, int Nimg,
                     int IH, int IW, int Ci, int Co, int accumulate, void* stream);
/* dv3_conv_s2_wgrad_tile: the same weight gradient for the narrow layer pair next to the image layers (Cfine 32,
 * Ccoarse 64; H, W multiples of 16): a workgroup walks 8 x 8 coarse-pixel tiles with the whole [64][16 x 32] gradient in
 * its accumulators; per-workgroup partial sums go to `partial` (dv3_conv_s2_wgrad_tile_scratch(...) floats; 0 = shape
 * not supported) and a second launch adds them into dw.  dv3_conv_s2_wgrad remains the general entry point. */
int dv3_conv_s2_wgrad_tile_scratch(int Nimg, int H, int W, int Cfine, int Ccoarse);
int dv3_conv_s2_wgrad_tile(const float* coarse, const float* fine, float* partial, float* dw, int Nimg, int H, int W,
                           int Cfine, int Ccoarse, void* stream);
int dv3_conv_s2_wgrad(const float* coarse, const float* fine, float* dw_packed, float* dw, int Nimg, int H, int W,
                      int Cfine, int Ccoarse, void* stream);
/* dw_packed: caller-owned scratch of Ccoarse*16*Cfine floats that must be ZERO on entry; partial sums land there
 * with coalesced fp32 atomics, then are added into dw (reference layout) and the scratch is cleared again. */
/* The 3-channel image-side layers (first encoder conv 3->CW, last decoder transposed conv CW->3, and each
 * other's input gradient), as HBM-bound VALU kernels with wave-uniform weights read in the REFERENCE layout
 * ([CW][3][4][4] for both; CW in {32, 96} = cnn_depth of the shipped configs). */
int dv3_conv_s2_c3_fwd(const float* x, const float* w, float* y, int Nimg, int H, int W, int CW, int accumulate,
                       void* stream);
int dv3_convT_s2_c3_fwd(const float* x, const float* w, const float* bias, float out_add, float* y, int Nimg,
                        int IH, int IW, int CW, int accumulate, void* stream);

/* ---- proprio inputs/outputs -- tools.symlog (tools.py:22-23), tools.SymlogDist (tools.py:543-572) --- */
int dv3_symlog(const float* x, float* y, long n, void* stream);
int dv3_symlog_mse(const float* mode, const float* x, float* loss, float* dmode, long R, int W, float upstream,
                   void* stream);

/* ---- continuous actor -- MLP.dist 'normal' (networks.py:693-700) + ContDist absmax (tools.py:594-598)
 * mean_raw/std_raw [M,A] are the two head Linears.  fwd: action = rescale(tanh(mean) + std*eps)
 * (action may be NULL), entropy [M] (may be NULL).  logp: Normal log_prob of a given action.
 * bwd: any of daction / dent / dlogp may be NULL. */
int dv3_actor_normal_fwd(const float* mean_raw, const float* std_raw, const float* eps, float* action,
                         float* entropy, long M, int A, float min_std, float max_std, void* stream);
int dv3_actor_normal_logp(const float* mean_raw, const float* std_raw, const float* action, float* logp, long M,
                          int A, float min_std, float max_std, void* stream);
int dv3_actor_normal_bwd(const float* mean_raw, const float* std_raw, const float* eps, const float* action,
                         const float* daction, const float* dent, const float* dlogp, float* dmean_raw,
                         float* dstd_raw, long M, int A, float min_std, float max_std,
                         int logp_of_sample, void* stream);
/* logp_of_sample = 1: dlogp is the gradient on log N(action) of the action rsampled from this very (mean, std)
 * through eps (policy.log_prob(imag_action), models.py:667: the path through the action is included); 0: the action
 * is a constant. */

/* ---- lambda-return + discount weights -- tools.lambda_return (tools.py:682-728),
 * ImagBehavior._compute_target (models.py:620-638).  reward/value/cont_logit/weights/disc [H,N],
 * target [H-1,N]; disc = gamma*sigmoid(cont_logit) (disc may be NULL). */
int dv3_lambda_return_fwd(const float* reward, const float* value, const float* cont_logit, float* target,
                          float* weights, float* disc, int H, long N, float gamma, float lam, void* stream);
int dv3_lambda_return_bwd(const float* dtarget, const float* value, const float* cont_logit, const float* target,
                          float* dreward, float* dcont_logit, int H, long N, float gamma, float lam, void* stream);

/* ---- loss assembly --------------------------------------------------------------------------------------
 * dv3_dot_accumulate: out[0] += scale * sum_i f(x_i) * w_i, f = max(., clip_min) if use_clip_min (w may be
 *   NULL = ones) -- the torch.mean / torch.clip(min=free) reductions of models.py:147-148, networks.py:286-288.
 * dv3_actor_loss: ImagBehavior._compute_actor_loss + entropy bonus (models.py:406-407, 640-681) and its
 *   gradients in one pass; target/value/weights/entropy/logp [H,N] (target: H-1 rows used), ema_vals [2].
 *   mode 0 ('dynamics'): writes dtarget [H-1,N]; mode 1 ('reinforce'): writes dlogp [H,N]; mode 2 ('both',
 *   models.py:670-676: mix*target + (1-mix)*logp*sg(target-value)): writes both.  loss_out[0] += loss.
 * dv3_scale_neg: out = -s*w (upstream of the two critic log-prob terms, models.py:424-429). */
int dv3_dot_accumulate(const float* x, const float* w, long n, float* out, int use_clip_min, float clip_min,
                       float scale, void* stream);
int dv3_actor_loss(const float* target, const float* value, const float* weights, const float* entropy,
                   const float* logp, const float* ema_vals, float* loss_out, float* dtarget, float* dlogp,
                   float* dentropy, int H, long N, float entropy_coef, int mode, float mix, void* stream);
int dv3_scale_neg(const float* w, float* out, long n, float s, void* stream);

/* ---- is_first reset -- RSSM.obs_step (networks.py:176-193), branch-free -----------------------------
 * out[b,:] = x[b,:]*(1-m_b) + init[:]*m_b  (x or init may be NULL = zeros).  bwd: dx = dout*(1-m)
 * (dx may be NULL), dinit += sum_b dout[b,:]*m_b (dinit may be NULL). */
int dv3_reset_blend(const float* x, long ldx, const float* init, const float* is_first, float* out, long ldo,
                    int B, int n, void* stream);
int dv3_reset_blend_bwd(const float* dout, long ldo, const float* is_first, float* dx, long ldx, float* dinit,
                        int B, int n, void* stream);

/* One-launch forms for an observe step: the three blends (stoch/deter against the learned initial state,
 * action against zero; prev_* NULL = zeros for the first step), and their backward with the carry folded
 * in: gs_prev += dsin*(1-m), gd_prev += ddin*(1-m) (NULL on the first step), dstoch0/ddeter0 += sum_b (.)*m.
 * dsin / ddin are row-strided (ld_*), everything else dense. */
int dv3_obs_blend(const float* prev_stoch, const float* init_stoch, const float* prev_deter, const float* init_deter,
                  const float* action, const float* is_first, float* out_stoch, float* out_deter, float* out_action,
                  int B, int SD, int De, int A, void* stream);
int dv3_obs_blend_bwd(const float* dsin, long ld_dsin, const float* ddin, long ld_ddin, const float* is_first,
                      float* gs_prev, float* gd_prev, float* dstoch0, float* ddeter0, int B, int SD, int De,
                      void* stream);

/* dv3_obs_blend_bwd of step t followed by dv3_onehot_st_bwd (sample form, accumulate) of step t-1 in one launch: the
 * carry gs_prev / gd_prev (+)= as above, then dlogit_prev [B,S,D] += the straight-through gradient of the posterior
 * sample of step t-1 (logit_prev [B,S,D]) with t = the updated gs_prev (RSSM.observe backward: networks.py:183-191,
 * tools.py:452-460).  dsin [B, S*D] / ddin [B, De] row-strided. */
int dv3_obs_carry_st_bwd(const float* dsin, long ld_dsin, const float* ddin, long ld_ddin, const float* is_first,
                         float* gs_prev, float* gd_prev, float* dstoch0, float* ddeter0, const float* logit_prev,
                         float* dlogit_prev, int B, int S, int D, int De, float unimix, void* stream);

/* ---- optimizer -- tools.Optimizer.__call__ (tools.py:760-776) on a flat fp32 bucket ----------------
 * state[0] = step count, state[1] = sum of squares accumulator, state[2] = last grad norm.
 * dv3_sumsq_accumulate adds sum(x^2) into *out (x 16-byte aligned).  dv3_adam_step clips by
 * clip/(norm+1e-6) (clip <= 0: no clipping), applies torch.optim.Adam's update, bumps the step and
 * clears the accumulator -- all on device.  dv3_axpby: y = a*x + b*y (slow critic, models.py:683-689). */
int dv3_sumsq_accumulate(const float* x, long n, float* out, void* stream);
/* The same sum in a fixed order (per-workgroup partial sums in partial[0 .. min(partial_len, 1024)), added up by index by
 * one workgroup; two launches, no atomics): the clipping norm of tools.py:768 as a pure function of the gradient, so that
 * data-parallel replicas holding the same all-reduced gradient take bit-identical steps. */
int dv3_sumsq_ordered(const float* x, long n, float* out, float* partial, int partial_len, void* stream);
int dv3_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float* state,
                  double lr, double beta1, double beta2, double eps, float clip, float weight_decay,
                  float grad_scale, void* stream);
/* lr / betas / eps are doubles because torch.optim.Adam derives its per-step scalars (1-beta, the bias
 * corrections, lr / (1 - beta1^t)) in double before rounding them to float: the kernel does the same, so that
 * its moments and parameters track a torch.optim.Adam run (and its saved state_dict) to fp32 rounding. */
/* grad_scale multiplies every gradient (and the norm) first: 1/world_size after a SUM all-reduce. */
int dv3_axpby(const float* x, float* y, long n, float a, float b, void* stream);
int dv3_rng_advance(unsigned long long* rng_state, unsigned long long increment, void* stream);
/* out[n] ~ N(0,1) from Philox(rng_state) -- the actor's rsample noise when none is injected
 * (torch.distributions.utils._standard_normal behind networks.py:697-699); consumes ceil(n/4) counters. */
int dv3_fill_normal(float* out, long n, const unsigned long long* rng_state, unsigned long long rng_offset,
                    void* stream);

/* Two quantiles q0, q1 of x[n] (torch.quantile's linear interpolation between the neighbouring order statistics,
 * selected exactly by a 4-pass radix select) and, when ema != NULL, ema[i] = alpha*q_i + (1-alpha)*ema[i]:
 * models.RewardEMA.__call__ (models.py:19-26) in one launch instead of a sort.  out_q (optional) receives q0, q1. */
int dv3_quantile2_ema(const float* x, long n, double q0, double q1, float* ema, float alpha, float* out_q,
                      void* stream);

/* tools.tensorstats (tools.py:949-958) in one launch: out4 = {mean, std (unbiased), min, max} of (x[i] - shift[0]) /
 * scale[0] over x[n]; shift / scale are optional DEVICE scalars (NULL: 0 / 1) -- the normed_target statistics of
 * models.py:412-414 read (target - ema[0]) / (ema[1] - ema[0] clipped) without materialising it. */
int dv3_tensorstats(const float* x, long n, const float* shift, const float* scale, float* out4, void* stream);
/* The same for up to six tensors in ONE launch (one workgroup each): out[4 * i ..] = statistics of tensor i.  The five
 * tensorstats groups ImagBehavior._train logs per update (value, target, imag_reward, imag_action, normed_target:
 * models.py:431-445, 659-661) were five launches of 20 us. */
int dv3_tensorstats_multi(int count, const float* x0, long n0, const float* shift0, const float* scale0,
                          const float* x1, long n1, const float* shift1, const float* scale1,
                          const float* x2, long n2, const float* shift2, const float* scale2,
                          const float* x3, long n3, const float* shift3, const float* scale3,
                          const float* x4, long n4, const float* shift4, const float* scale4,
                          const float* x5, long n5, const float* shift5, const float* scale5, float* out, void* stream);

/* dv3_onehot_sample_fwd_ex (D = 32, with the next observe step's reset blend: next_first [M], init [S][32], init_idx [S]
 * -> next_out [M][S][32], next_idx [M][S]) followed by dv3_onehot_linear_ln_fwd on the blended indices, in ONE launch:
 * the posterior sample of obs_step t (networks.py:199-204) and the _img_in_layers of obs_step t+1 (networks.py:216-218).
 * Same draws as the stand-alone sampler (bit-equal).  N % 256 == 0, N <= 1024, S <= 32. */
int dv3_onehot_sample_linear_ln_fwd(const float* logit, const float* noise, const unsigned long long* rng_state,
                                    unsigned long long rng_offset, float* onehot, int* idx, const int* forced,
                                    unsigned int* flips, float unimix, int mode, const float* next_first,
                                    const float* init, const int* init_idx, float* next_out, int* next_idx, int S,
                                    const float* x2, long ldx2, int A2, const float* WT, long ldw, float* pre,
                                    long ldpre, const float* gamma, const float* beta, float* y, long ldy, float* mean,
                                    float* rstd, long M, int N, int act, void* stream);

/* ---- few-row launches of the observe scan with the preceding row operation in the GEMM's prologue
 * (csrc/scanops.hip; RSSM.obs_step, networks.py:174-206) ------------------------------------------------
 * Every workgroup of the few-row GEMM recomputes the row sums of the (<= 16 per row block) complete rows and
 * transforms only the A fragments it multiplies; workgroup column 0 stores the transformed rows.
 * dv3_scan_ln_gemm_fwd: y = SiLU(LN(x)) (= dv3_ln_act_fwd), then C[M,N] (+)= y W^T + bias (RSSM._obs_stat_layer on
 *   the output of _obs_out_layers, networks.py:197-200); bit-identical to the two launches it replaces.
 *   K in {256, 512, 1024}; N % 16 == 0; y / mean / rstd optional. */
int dv3_scan_ln_gemm_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy,
                         float* mean, float* rstd, const float* W, long ldw, const float* bias, float* C, long ldc,
                         int M, int K, int N, int accumulate, void* stream);

/* Reverse observe scan (same file): the data-gradient few-row GEMMs dX = dY W (W [K][N] row-major; K split over
 * workgroups, partial tiles added with atomics as dv3_gemm_f32 accumulate = 2 does) with the row operation that
 * produces dY in their prologue.
 * dv3_scan_ln_factors: xhat = (x - mean) rstd and jac = SiLU'(xhat gamma + beta), [R, K] contiguous: the factors of a
 *   LayerNorm + SiLU layer's backward that depend on forward data only -- computed once per update for all T*B rows
 *   in front of the reverse scan, so that the per-step prologue below is multiply-add only.
 * dv3_scan_lnbwd_gemm: dx = rstd (dy jac gamma - mean_c(.) - xhat mean_c(. xhat)) (= dv3_ln_act_bwd, act = 1;
 *   d-gamma / d-beta accumulated), then C[M,N] += dx W.  K in {256, 512, 1024}, N % 64 == 0.  Reverse of
 *   RSSM._obs_out_layers / _img_in_layers inside obs_step (networks.py:195-197, 216-218).
 * dv3_scan_carry_st_gemm: t = gs + dsin (1 - first) (the carry through the next step's reset blend,
 *   networks.py:183-191; dstoch0 += sum_b dsin first; dsin == NULL: no carry, the scan's last step),
 *   dlogit_out = dlogit + straight-through gradient of the posterior sample at t (tools.py:452-460; = dv3_onehot_st_bwd),
 *   C[B,N] += dlogit_out W; extra workgroups add the deter carry gd += ddin (1 - first), ddeter0 += sum_b ddin first
 *   (= dv3_obs_carry_st_bwd).  D == 32, S % 8 == 0, N % 64 == 0; dlogit_out must not alias dlogit. */
/* dv3_scan_gru_factors / dv3_scan_grubwd_gemm: the same for the GRU cell (GRUCell.forward reversed,
 *   networks.py:760-768; = dv3_gru_bwd followed by the data-gradient GEMM of its Linear).  Factors per row, computed
 *   once per update: xhat [R, 3 De], afac [R, 3 De] (d gate pre-activation / d new state), p1, p2 [R, De] (the two
 *   LayerNorm row sums as dot products with the state gradient), ah [R, De] (1 - update gate).  The per-step launch
 *   writes dp [M, 3 De] (gradient on the Linear's output), adds g * ah onto dh (atomic) and C[M,N] += dp W (atomic);
 *   d-gamma / d-beta accumulated.  De in {256, 512, 1024}, N % 64 == 0. */
int dv3_scan_gru_factors(const float* p, long ldp, const float* gamma, const float* beta, const float* h, long ldh,
                         const float* mean, const float* rstd, float* xhat, float* afac, float* p1, float* p2,
                         float* ah, long R, int De, void* stream);
int dv3_scan_grubwd_gemm(const float* g, long ldg, const float* xhat, const float* afac, const float* p1,
                         const float* p2, const float* ah, const float* gamma, const float* rstd, float* dp, float* dh,
                         long lddh, float* dgamma, float* dbeta, const float* W, long ldb, float* C, long ldc, int M,
                         int De, int N, void* stream);
int dv3_scan_ln_factors(const float* x, long ldx, const float* gamma, const float* beta, const float* mean,
                        const float* rstd, float* xhat, float* jac, long R, int K, void* stream);
int dv3_scan_lnbwd_gemm(const float* dy, long lddy, const float* xhat, const float* jac, const float* gamma,
                        const float* rstd, float* dx, long lddx, float* dgamma, float* dbeta, const float* W, long ldb,
                        float* C, long ldc, int M, int K, int N, void* stream);
int dv3_scan_carry_st_gemm(const float* dsin, long ld_dsin, const float* ddin, long ld_ddin, const float* is_first,
                           const float* gs, float* gd, float* dstoch0, float* ddeter0, const float* logit,
                           const float* dlogit, float* dlogit_out, const float* W, long ldb, float* C, long ldc, int B,
                           int S, int D, int De, int N, float unimix, void* stream);

/* ---- row-fused layers of the imagination step (csrc/fusedops.hip) -----------------------------------
 * dv3_onehot_linear_ln_fwd: pre[M,N] = base + sum_s WT[s*D + idx[m][s]] + sum_a x2[m][a] * WT[S*D + a], then
 * y = act(LN(pre)) (y == NULL: pre only).  The Linear + LayerNorm + SiLU whose input is cat[stoch.flat, tail]
 * (RSSM._img_in_layers on cat[stoch, action], networks.py:216-218; the first layer of every MLP head on
 * get_feat = cat[stoch, deter], networks.py:154-159, 657-668) with the stoch columns an exact one-hot per group
 * (tools.py:452-460): S rows of the TRANSPOSED weight WT [S*D + A2][N] are gathered by class index instead of
 * multiplying the one-hot through an MFMA GEMM.  base [M,N] (optional, may alias pre): the product of the other
 * input columns (deter) from dv3_gemm_f32.  idx int32 [M,S], S <= 64.  mean/rstd [M] saved for dv3_ln_act_bwd.
 * dv3_actor_head_fwd: y = SiLU(LN(pre)) of the actor trunk's last layer [M,U<=1024], its heads
 * out_m = y Wm^T + bm (and out_s = y Ws^T + bs for the continuous actor), the action sample and the entropy:
 * MLP.forward tail + dist 'normal' / 'onehot' (networks.py:672-681, 693-700, 713-714), ContDist.sample with
 * absmax 1 (tools.py:594-598) / OneHotDist.sample (tools.py:452-460).  noise [M,A] (N(0,1) / Exp(1)) or, when
 * NULL, the Philox draws dv3_fill_normal / dv3_onehot_sample_fwd would make at rng_offset; eps_out receives the
 * N(0,1) draws used.  onehot=1: Ws/bs/out_s NULL, act_idx/forced/flips as in dv3_onehot_sample_fwd_ex.
 * dv3_transpose2d: dst[c*ldd + r] = src[r*lds + c] (weights [N][K] -> WT [K][N]).
 * dv3_onehot_to_idx: idx[g] = argmax_d onehot[g][d] for R groups of D. */
int dv3_onehot_linear_ln_fwd(const int* idx, int S, int D, const float* x2, long ldx2, int A2, const float* WT,
                             long ldw, const float* base, long ldbase, float* pre, long ldpre, const float* gamma,
                             const float* beta, float* y, long ldy, float* mean, float* rstd, long M, int N, int act,
                             void* stream);
int dv3_actor_head_fwd(const float* pre, long ldpre, const float* gamma, const float* beta, float* y, long ldy,
                       float* mean, float* rstd, const float* Wm, const float* bm, const float* Ws, const float* bs,
                       float* out_m, float* out_s, const float* noise, const unsigned long long* rng_state,
                       unsigned long long rng_offset, float* eps_out, float* action, float* entropy, int* act_idx,
                       const int* forced, unsigned int* flips, long M, int U, int A, float min_std, float max_std,
                       float unimix, int onehot, void* stream);
int dv3_transpose2d(const float* src, long lds, int R, int C, float* dst, long ldd, void* stream);
/* Up to 12 transposes in one launch (the per-update re-packing of the weights the one-hot gathers and the reverse
 * imagination rollout read).  jobs_host: HOST array of njobs x 6 unsigned 64-bit values {src, dst, lds, ldd, R, C}
 * (device pointers as integers), consumed before the call returns. */
int dv3_transpose2d_many(int njobs, const unsigned long long* jobs_host, void* stream);
int dv3_onehot_to_idx(const float* onehot, int* idx, long R, int D, void* stream);

/* ---- compute-unit partitioned streams (csrc/streams.hip) --------------------------------------------------------
 * The MI355X-native counterpart of nothing in the reference (its update is one stream of ATen launches,
 * models.py:105-169): the reverse observe scan and the weight gradients that nothing reads before the optimizer run
 * on two HIP streams whose hardware queues own complementary halves of the compute units.
 * mask_words: HOST array, bit b of word w = compute unit 32*w + b may be used; stream_out: HOST 64-bit slot that
 * receives the hipStream_t.  The stream is a blocking stream (it synchronises implicitly with the NULL stream). */
int dv3_device_cu_count(int* count_out);
int dv3_stream_create_cu_masked(int n_words, const unsigned int* mask_words, unsigned long long* stream_out);
int dv3_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DV3HIP_H_ */
