/* cd360_hip.h -- C ABI of libcd360_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the pose-conditioned denoising hot path of custom-diffusion360
 * (SURVEY.md §8b).  The reference is pure Python; the native arithmetic it reaches on this path
 * lives in third-party operators.  Each entry point below names the reference call site(s) it
 * replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates; nothing is allocated here);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no hidden synchronisation; re-entrant across
 *     streams.  No entry point reads the environment.  State: (1) a process-wide DEFAULT tuning struct (cd360_set_tuning: an explicit
 *     A/B interface whose defaults, -1 everywhere, are the measured-best choices); (2) keyed BY STREAM and mutex-protected, so that two
 *     samplers / captures on two streams never see each other's: tuning overrides (cd360_set_stream_tuning) and weight-prefetch arms
 *     (cd360_prefetch_arm_on).  A launch consults only the entries of the stream it is given;
 *   - workspace sizes come from the *_workspace_bytes() queries, the caller provides the buffer;
 *   - return 0 on success, <0 on error (CD360_ERR_*): the Python side raises;
 *   - bf16 tensors are raw uint16 storage; "fp32"/"int32" as named;
 *   - cameras are packed rows of 16 fp32: R (row-major 9) | T (3) | focal (2) | principal point (2), PyTorch3D
 *     NDC convention (X_view = X_world R + T, +X left, +Y up), index 0 = target view, 1..n = reference views.
 */
#ifndef CD360_HIP_H
#define CD360_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CD360_OK 0
#define CD360_ERR_ARG (-1)    /* null / misaligned pointer, non-positive size */
#define CD360_ERR_SHAPE (-2)  /* unsupported shape or stride */
#define CD360_ERR_LAUNCH (-3) /* HIP launch failure */

/* ---- tuning / A-B interface --------------------------------------------------------------------------------
 * Kernel and tiling choices are made per call from the shapes.  A harness that wants to compare two tilings in one process
 * overrides them here (field = -1: choose by shape).  `size` must be sizeof(cd360_tuning).  cd360_set_tuning(NULL) restores the
 * defaults.  Set the default between launches, not concurrently with them (per-stream overrides: cd360_set_stream_tuning).  `whatif` (timing experiments that produce wrong results) is
 * ignored unless the library was built with -DCD360_WHATIF (cd360_whatif_build() == 1); the product build has no such code. */
typedef struct cd360_tuning {
  int32_t size;
  int32_t gemm_cfg;         /* 1..8: tiling of cd360_gemm_bf16 */
  int32_t gemm_group_m;     /* > 0: token tiles per tile group */
  int32_t gemm_movers;      /* 0 | 4: mover waves off / on */
  int32_t gemm_ksplit;      /* 0 | 1 | 2: wave arrangement of the 128 x 128 four-buffer tiling */
  int32_t conv_cfg;         /* 1..6: tiling of cd360_conv3x3_dma_bf16 */
  int32_t conv_dma;         /* 0: convolutions on the register-staged kernel */
  int32_t conv_kgroup;      /* > 0: K-order group size (before weights are packed); process-wide only: cd360_set_stream_tuning returns CD360_ERR_ARG for a different value */
  int32_t conv_wide;        /* 0: no 160-channel tiles in the register-staged kernel */
  int32_t conv_wmajor;      /* 0 | 1: tile order of the register-staged kernel */
  int32_t conv_split;       /* 1 | 2: in-workgroup split-K of the register-staged kernel */
  int32_t attn_smallk;      /* 0: <= 96-key attention on the tiled kernel */
  int32_t attn_smallk_wgs;  /* > 0: workgroup target of the register-resident kernel */
  int32_t attn_self;        /* 0 | 1 | 2 | 3: self-attention kernel generation / tiling */
  int32_t attn_fast;        /* 0: guarded path of the first-generation kernel */
  int32_t nerf_kernel;      /* 0 | 1: FeatureNeRF render kernel with register gathers / full-line gathers */
  int32_t qattn_cfg;        /* 1..4: tile of cd360_qproj_attn_bf16 (256 x 256 / 128 x 128 + movers / 128 x 128 x 2 WGs / 256 x 128) */
  int32_t whatif;           /* -DCD360_WHATIF builds only */
  int32_t gemm_small;       /* 0: no small-batch tilings (64 x 128 tiles, 128-wide tiles for wide outputs): the A/B partner */
  int32_t qattn_keys16;     /* 0: 65 .. 80 keys padded to 96 in cd360_qproj_attn_bf16 (default: five groups of 16) */
  int32_t qattn_split;      /* 1: second launch for the last 128 columns of a 256 k + 128 wide projection (A/B only: measured slower) */
  int32_t store_wt;         /* 0 | 1: GEMM-family output tiles by plain / write-through (sc1) stores; -1 = the measured default */
  int32_t conv_halo;        /* 0 | 1: halo form of the 3x3 convolution (input pixels of a tile fetched once per 64-channel chunk) never / wherever it fits */
  int32_t gemm_asm4;        /* 0 | 1: 256 x 256 tiles as four waves of 128 x 128 on a generated instruction stream never / always; -1 = where measured */
  int32_t reserved[2];
} cd360_tuning;
int cd360_set_tuning(const cd360_tuning* t);   /* the process-wide DEFAULT (NULL restores the built-in defaults) */
int cd360_get_tuning(cd360_tuning* t);
/* Per-stream override -- the re-entrant form ("no global mutable state, re-entrant across streams"): every launch
 * issued ON `stream` reads *t instead of the default, so two samplers / two captures in one process hold different tilings; t == NULL
 * removes the override; at most 16 streams; thread-safe.  cd360_get_stream_tuning returns what launches on `stream` read (override or
 * default).  The shape queries below that size a buffer for a following launch (cd360_gemm_tile_n, cd360_gemm_cstats_rows,
 * cd360_conv_stats_slabs, cd360_conv_stats_rows, cd360_conv_dma_slab_rows, cd360_conv_k_order) take no stream: on the CALLING THREAD they
 * answer for the stream named by the last cd360_query_stream (NULL or a stream without override: the default) -- thread-local state only. */
int cd360_set_stream_tuning(void* stream, const cd360_tuning* t);
int cd360_get_stream_tuning(void* stream, cd360_tuning* t);
int cd360_query_stream(void* stream);
int cd360_whatif_build(void);

/* ---- weight prefetcher of a captured step --------------------------------------------------------------------
 * One denoise step streams 5.3 GB of weights through a 256 MB Infinity Cache: every GEMM / convolution launch finds its weights in HBM.
 * The launches of a captured step are a fixed sequence, so they can be fetched ahead.  cd360_prefetch_arm -- called INSIDE a stream
 * capture, after `side_stream` has been forked into it -- makes every following launch of the GEMM family (cd360_gemm_bf16,
 * cd360_qproj_attn*_bf16, cd360_conv3x3_dma_bf16, cd360_conv_up2x_bf16, cd360_gemm_cstats_bf16) enqueue on the side stream a small kernel
 * (`wgs` workgroups) that touches one dword per 128-byte line of that launch's weights (weights below `min_bytes` are skipped) and waits
 * only for the completion of the launch `lag` (1 .. 7) positions earlier: in the replayed graph the weights of launch i arrive in the
 * Infinity Cache while launches i - lag + 1 ... i - 1 compute.  The main stream never waits for the side stream; the caller joins it once,
 * after the disarm, before the capture ends.  sink: 4 bytes of device scratch. */
int cd360_prefetch_arm_on(void* main_stream, void* side_stream, int lag, int wgs, int64_t min_bytes, void* sink);
int cd360_prefetch_disarm_on(void* main_stream);
/* cd360_prefetch_arm_on arms ONE capturing stream: only launches issued on `main_stream` enqueue touches (on `side_stream`, forked into the
 * same capture), so two captures on two streams prefetch independently (up to 8 armed streams; thread-safe).  The two entry points below
 * are the round-4 forms: a wildcard arm serving launches on any stream that has no arm of its own (one capture at a time). */
int cd360_prefetch_arm(void* side_stream, int lag, int wgs, int64_t min_bytes, void* sink);
int cd360_prefetch_disarm(void);

/* ---- attention ---------------------------------------------------------------------------------------------
 * replaces xformers.ops.memory_efficient_attention(q, k, v, attn_bias=None, op=None)
 *          sgm/modules/attention.py:393-408 (MemoryEfficientCrossAttention.forward), head dim 64, no mask.
 * Strided form: q[b][h][n][d] at q + b*qs[0] + h*qs[1] + n*qs[2] + d (element strides, multiples of 8); k and v likewise
 * (v ROW-MAJOR, v[b][h][key][d]: transposed on the fly by the kernel's LDS reads); o like q (strides multiples of 4).  With
 * these strides the kernel consumes the three slices of ONE merged q|k|v projection output [b, N, 3*H*64] in place and writes
 * [b, N, H*64] (the permute / contiguous copies of attention.py:394-418 are gone). */
int cd360_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                        const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides,
                        float scale, void* stream);
/* fp8-MFMA variant of cd360_attn_fwd_bf16 for Nk <= 96 -- the text / pose-token cross-attention of attention.py:578-588,620-625
 * (BASELINE.json configs[4]): same bf16 tensors and strides; Q, K, V and the softmax probabilities are rounded to OCP e4m3 in
 * registers with per-tensor scales amax[i] / 448 (amax = {max|q|, max|k|, max|v|}, host floats) and both contractions run on
 * v_mfma_f32_32x32x16_fp8_fp8.  Returns CD360_ERR_SHAPE for Nk > 96 or output rows that are not 16-byte aligned. */
int cd360_attn_fwd_fp8mfma_bf16(const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                                const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                const int64_t* o_strides, float scale, const float* amax, void* stream);

/* cd360_attn_fwd_bf16 for a q that already carries the softmax scale and the base change: q' = q * (scale * log2 e),
 * o = softmax_2(q' k^T) v (base-2 softmax; identical to softmax(scale q k^T) v).  The transformer blocks fold the factor into the q rows
 * of the merged q|k|v projection when they pack it (attention.py:393-408 computes q with nn.Linear, then xformers scales inside), so q'
 * is rounded to bf16 once, like the reference's q, and self-attention runs on the whole-tile kernel (lazy running maximum, LDS-DMA
 * K / V ring) whenever Nq % 128 == 0 and Nk % 64 == 0; other shapes take the kernels of cd360_attn_fwd_bf16. */
int cd360_attn_fwd_prescaled_bf16(const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                                  const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                  const int64_t* o_strides, void* stream);

/* Training pair (the fine-tuning loop differentiates attention.py:406 through xformers' autograd; BASELINE.json configs[3]).
 * cd360_attn_fwd_lse_bf16 = cd360_attn_fwd_bf16 that also writes lse [B*H, Nq] fp32, the natural-log log-sum-exp of every
 * query row's scaled scores.  cd360_attn_bwd_bf16 takes q, k, v, o, dout (bf16, forward layouts, strides multiples of 8) and
 * that lse, and writes dq (may be NULL) and dk / dv (both or neither) in bf16 with their own strides (multiples of 4) -- e.g.
 * the three column slices of one d(q|k|v) buffer.  delta_ws: B*H*Nq floats of caller-allocated workspace.  No atomics: the
 * result is deterministic. */
int cd360_attn_fwd_lse_bf16(const void* q, const void* k, const void* v, void* o, void* lse, int B, int H, int Nq, int Nk,
                            const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides,
                            float scale, void* stream);
int cd360_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* dout, const void* lse, void* dq,
                        void* dk, void* dv, void* delta_ws, int B, int H, int Nq, int Nk, const int64_t* q_strides,
                        const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, const int64_t* do_strides,
                        const int64_t* dq_strides, const int64_t* dk_strides, const int64_t* dv_strides, float scale, void* stream);

/* xformers layout: q, k, v, o contiguous [B*H, N, 64], exactly the call of attention.py:406. */
int cd360_attn_fwd_xformers_bf16(const void* q, const void* k, const void* v, void* o, int BH, int Nq, int Nk, float scale, void* stream);

/* ---- rays, projection, integer bilinear indices ----------------------------------------------------------------
 * replaces get_patch_rays / get_patch_raybundle / get_directional_raybundle  (sgm/modules/utils_cameraray.py:61-196)
 *          and the pytorch3d calls inside them (unproject_points, get_camera_center).
 * cams [b, n+1, 16]; xs, ys [r] NDC patch positions (host computes them exactly as utils_cameraray.py:106-147);
 * rays out [b, n+1, r*r, 6] fp32 = (origin, unit direction), ray k = row*r + col. */
int cd360_patch_rays(const void* cams, const void* xs, const void* ys, void* rays, int b, int n, int r, void* stream);
/* replaces ray_bundle_to_ray_points (nerfsd_pytorch3d.py:381-387), cam.transform_points_ndc (:73-77), the
 * negate / nan_to_num / clip (:89-95) and the index arithmetic inside F.grid_sample(align_corners=True) (:79-98).
 * t: sample depths [hw, S] (t_ray_stride = S) or [S] (t_ray_stride = 0).
 * outputs, any may be NULL: points [b, hw, S, 3] fp32; grid [b, n, hw, S, 2] fp32 (x first);
 * x0, y0 [b, n, hw, S] int32 north-west texel; mask [b, n, hw, S] int32 (bit0 nw, bit1 ne, bit2 sw, bit3 se in-bounds).
 * Bit-exact against oracle/pose_path.py (sample_grid, bilinear_corners). */
int cd360_ray_project_index(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, int b, int n, int r, int S,
                            void* points, void* grid, void* x0, void* y0, void* mask, void* stream);
/* replaces pytorch3d._C.sample_pdf(bins, weights, outputs, eps) as called by Raymarcher.importance_sampling
 * (sgm/modules/nerfsd_pytorch3d.py:300-306; SURVEY.md section 8 row f4 -- dead code upstream: NerfSDModule.forward never forwards
 * imp_sample_next_step, :442 vs :333).  pytorch3d is not part of the reference tree; the algorithm is its published sample_pdf_python
 * (inverse-CDF sampling, the NeRF hierarchical sampler): w = weights + eps, pdf = w / sum w, cdf = (0, cumsum pdf), k = searchsorted(cdf,
 * u, right), sample = bins[k-1] + (u - cdf[k-1]) / (cdf[k] - cdf[k-1]) (bins[k] - bins[k-1]) (denominator 1 when < eps, indices clamped).
 * bins [rows, n_bins + 1], weights [rows, n_bins], u [rows, n_samples] in [0, 1) -> samples [rows, n_samples] (may alias u: pytorch3d's
 * in-place form); dists (may be NULL; needs samples != u) [rows, n_samples] = gap to the next sample, the last to bins[n_bins] (:306).
 * fp32 throughout.  Errors: CD360_ERR_ARG for NULL / non-positive sizes / eps <= 0 / dists with samples == u. */
int cd360_sample_pdf(const void* bins, const void* weights, const void* u, void* samples, void* dists, float eps, int64_t rows,
                     int n_bins, int n_samples, void* stream);

/* ---- feature gather --------------------------------------------------------------------------------------------
 * replaces F.grid_sample(input[b*n, C, r, r], grid[b*n, hw, S, 2], bilinear, align_corners=True, zeros)
 *          nerfsd_pytorch3d.py:79-98, on channels-last data: xref [n_img, r*r, C], grid [n_img, P, 2] fp32,
 * out [n_img, P, C].  dtype: 0 = fp32, 1 = bf16 (xref and out). */
int cd360_feature_gather(const void* xref, const void* grid, void* out, int n_img, int pts_per_img, int r, int C, int dtype, void* stream);

/* ---- FeatureNeRF per-sample MLP + view aggregation ----------------------------------------------------------------
 * replaces FeatureNeRFEncoding.forward lines nerfsd_pytorch3d.py:102-158 (frame changes, positional encodings,
 * torch.cat, plane_coefs, nviews softmax, weighted sum over views) together with cd360_plucker_features and three
 * host GEMMs (see custom-diffusion360_amd/cd360/nerf.py and DESIGN.md §3 for the algebra).
 *   Y     [b*n, hw, C] bf16 = xref @ W1[:, :C]^T            zP [b*n, hw, C] bf16 = plucker_feats @ W1[:, C+99:]^T + b1
 *   lv    [b*n, hw] fp32    = xref @ w_v[:C]                cview [b, n] fp32 per-view logit constant
 *   Wk    [C, cd360_nerf_k_padded()] bf16 = W1[:, C:C+99] with columns permuted to the kernel's input order
 *   g out [b, hw*S, C] bf16 = sum_i softmax_i(logit_i) * SiLU(z_i)   (plane_coefs.2 is applied to g by the caller)
 *   img_map [b*n] int32 (optional): Y / lv may then hold only the DISTINCT reference images [n_tab, hw, C] and (batch, view)
 *   reads table image img_map[batch*n + view] -- the CFG batch of sample.py:89-96 repeats the same views three times;
 *   logits [b, n, hw*S] fp32 and lse [b, hw*S, 2] = (max, sum) are optional (NULL to skip).  C % 64 == 0. */
int cd360_plucker_features(const void* cams, const void* xs, const void* ys, void* out /* [b, n, hw, 104] fp32 */, int b, int n, int r,
                           void* stream);
/* the same features written as bf16 rows of 128 (99 values + zero pad), 16-byte aligned: the A operand of the table GEMM
 * zP = [enc8(plucker), dir] Wp^T + b1 on cd360_gemm_bf16 (no fp32 intermediate, no cast pass) */
int cd360_plucker_features_bf16(const void* cams, const void* xs, const void* ys, void* out, int b, int n, int r, void* stream);
int cd360_nerf_k_padded(void);
int cd360_nerf_mlp_aggregate(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, const void* Y,
                             const void* zP, const void* lv, const void* cview, const void* Wk, const void* img_map, void* g, void* logits,
                             void* lse, int b, int n, int r, int S, int C, void* stream);

/* ---- volume rendering --------------------------------------------------------------------------------------------
 * replaces _TruncExp.forward (sgm/modules/attention.py:192-199) + VolRender.forward/get_weights
 *          (nerfsd_pytorch3d.py:170-231) + the sigmoid on rgb (attention.py:594).
 * feats [b, hw, S, C] (dtype 0 fp32 / 1 bf16), sigma_raw [b, hw, S] fp32, rgb_raw [b, hw, S, 3] fp32 or NULL,
 * dists [hw, S] (d_ray_stride = S) or [S] (0).  Outputs: rendered [b, hw, C] (dtype of feats); fg [b, hw];
 * alphas, weights [b, hw, S]; rgb [b, hw, 3] (fp32; any of fg/alphas/weights/rgb may be NULL).  S <= 64.
 * flags: bit0 = sigma_raw already exponentiated, bit1 = rgb_raw already sigmoid'ed (the VolRender module API, :196-231). */
int cd360_volrender(const void* feats, const void* sigma_raw, const void* rgb_raw, const void* dists, int d_ray_stride, void* rendered,
                    void* fg, void* alphas, void* weights, void* rgb, int b, int hw, int S, int C, int dtype, int flags, void* stream);

/* cd360_nerf_mlp_aggregate in TWO passes (the default of the Python binding): pass 1 computes, once per (batch, view, sample), what does
 * not depend on the channel slice -- the projection into the view, the bilinear corner pixel / clamps / mask / fractions (the same ordered
 * fp32 chains: indices bit-exact), the view logit and the softmax statistics over the views -- into `ws`; pass 2 is the gather + 99-input
 * MFMA slice + SiLU + weighted accumulate per 64-channel slice, reading a 32-byte record instead of redoing that work C / 64 times.
 * ntab = table images in Y / lv (b * n without img_map); ws: cd360_nerf_ws_bytes(b, n, r, S) bytes, 16-byte aligned scratch. */
int64_t cd360_nerf_ws_bytes(int b, int n, int r, int S);
int cd360_nerf_mlp_aggregate_ws(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, const void* Y, const void* zP,
                                const void* lv, const void* cview, const void* Wk, const void* img_map, void* g, void* logits, void* lse, int b,
                                int n, int r, int S, int C, int ntab, void* ws, int64_t ws_bytes, void* stream);

/* Backward of cd360_nerf_mlp_aggregate for the fine-tuning loop (torch autograd through FeatureNeRFEncoding.forward in the
 * reference; the trainable parameters here are plane_coefs, nviews and decoder, diffusion.py:139-144).  Inputs as the forward,
 * plus its outputs g and lse, and dg [b, hw*S, C] bf16.  Writes dz [b, n, hw*S, C] bf16 (= softmax_i dg SiLU'(z_i)) and the
 * generated per-sample inputs F [b, n, hw*S, cd360_nerf_k_padded()] bf16 (the host forms dWk = dz^T F and dzP = sum_s dz with
 * library calls), and ACCUMULATES with fp32 atomics into caller-zeroed dY [tables, hw, C], dlogit [b, n, hw*S],
 * dlv [tables, hw], dcview [b, n].  dY may be NULL, and dlv / dcview both NULL: the training path skips the table scatters and
 * forms the weight gradients it needs as GEMMs against the gathered reference features (cd360_feature_gather). */
int cd360_nerf_mlp_aggregate_bwd(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, const void* Y,
                                 const void* zP, const void* lv, const void* cview, const void* Wk, const void* img_map, const void* g,
                                 const void* lse, const void* dg, void* dz, void* F, void* dY, void* dlogit, void* dlv, void* dcview, int b,
                                 int n, int r, int S, int C, void* stream);
/* The deterministic form (what the package calls): dlogit is WRITTEN -- no zero-fill -- as the sum, in channel-chunk order, of the terms the
 * kernel stores into the scratch dlogit_parts [C / 64, b, n, hw*S] fp32 (fully written); no atomic touches it.  dY / dlv / dcview as above.
 * With fp32 atomics the ~C / 64 additions per element came in launch order, the view-logit parameters' gradients moved in their last bits
 * from run to run, and now and then that flipped the bf16 rounding of a trained weight (1e-4 of the loss a few steps later). */
int cd360_nerf_mlp_aggregate_bwd_det(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, const void* Y,
                                     const void* zP, const void* lv, const void* cview, const void* Wk, const void* img_map, const void* g,
                                     const void* lse, const void* dg, void* dz, void* F, void* dY, void* dlogit, void* dlogit_parts, void* dlv,
                                     void* dcview, int b, int n, int r, int S, int C, void* stream);

/* Backward of cd360_volrender for the fine-tuning loop (the reference differentiates VolRender.forward (nerfsd_pytorch3d.py:170-231)
 * and _TruncExp, whose backward is g * exp(clamp(x, -15, 15)) (attention.py:203-207), through torch autograd).  Same feats /
 * sigma_raw / rgb_raw / dists / flags as the forward call; incoming gradients d_rendered [b, hw, C] (dtype of feats; required),
 * d_fg [b, hw], d_alphas, d_weights [b, hw, S], d_rgb [b, hw, 3] (fp32; NULL = zero).  Outputs d_feats [b, hw, S, C] (dtype of
 * feats), d_sigma_raw [b, hw, S] fp32, d_rgb_raw [b, hw, S, 3] fp32 (may be NULL). */
int cd360_volrender_bwd(const void* feats, const void* sigma_raw, const void* rgb_raw, const void* dists, int d_ray_stride,
                        const void* d_rendered, const void* d_fg, const void* d_alphas, const void* d_weights, const void* d_rgb,
                        void* d_feats, void* d_sigma_raw, void* d_rgb_raw, int b, int hw, int S, int C, int dtype, int flags, void* stream);

/* replaces FeatureNeRFEncoding.decoder, Linear(C -> 1+3, bias=False) (nerfsd_pytorch3d.py:49-51,160) and the channel split in
 * NerfSDModule.forward (:443-449): h [rows, C] bf16, w [4, C] fp32 -> out [rows, 4] fp32 = (rgb_raw 0..2, sigma_raw 3). */
int cd360_rowdot4_bf16(const void* h, const void* w, void* out, int64_t rows, int C, void* stream);
/* The view-logit column of the FeatureNeRF reference tables: lv = xref . vf, where vf folds the aggregation head with the feature rows of
 * plane_coefs.0 (nerfsd_pytorch3d.py:130-146; cd360/nerf.py reference_tables): h [rows, C] bf16, w [C] fp32 -> out [rows] fp32, fp32
 * accumulation, one read of h (this was a torch.mv -> library gemv over an fp32 copy of the features). */
int cd360_rowdot1_bf16(const void* h, const void* w, void* out, int64_t rows, int C, void* stream);
/* Fine-tuning (trainkeys = pose, sgm/models/diffusion.py:139-144): the operands the fused render reads, derived from the LIVE FeatureNeRF
 * parameters in one launch -- the slices of plane_coefs.0.weight W1 [C, C + 198] behind FeatureNeRFEncoding.forward's concatenated input
 * (nerfsd_pytorch3d.py:130-134: features | xyz encoding | Plucker / direction), the two parts of nviews.weight (:146-147), biases and the
 * decoder in fp32 -- and the way back for their gradients.  All parameters bf16, contiguous; kcol [NK] / kpos [C + 198] int32 device tables
 * (cd360/nerf.py xyz_k_columns and its inverse).
 * pack:   out_bf16 = Wf [C, C] | Wk [C, NK] | Wp [C, 128];  out_f32 = b1 [C] | b2 [C] | vf [C] | v_cam [99], 1 pad | bv, 3 pad | Wd [4, C].
 * unpack: grads[9] = dWf, dWk, dWp, db1, db2, dvf, dvc [99], dbv, dWd (NULL = zero), dtypes[9] (0 fp32, 1 bf16) -> dW1 [C, C + 198] bf16,
 *         small = db1 [C] | db2 [C] | dwv [C + 198] | dbv, 1 pad | dWd [4, C] (bf16). */
/* The three rendering terms of the fine-tuning loss for one pose block (StandardDiffusionLossImgRef.get_loss, sgm/modules/diffusionmodules/
 * loss.py:188-207): out[b] = (mean_k (clamp(fg, 0, 1) - op)^2, mean_{k,s} |alpha - op| bgw, sum_{c,k} (want - rgb)^2 mask / den), with op the
 * resized opacity, bgw = (1 - op) [op < 0.1], mask / want the resized mask and rgb target (constants of the step, prepared by the caller),
 * den = mask.sum + 1e-6 of the full-size mask.  fg [b, hw], alpha [b, hw, S], rgb [b, hw, 3] (NULL: no rgb term), op / bgw / mask [b, hw],
 * want [b, 3, hw], den [b], out [b, 3]; fp32; one workgroup per batch element, fixed summation order.  _bwd: g [b, 3] arriving on out ->
 * d_fg, d_alpha, d_rgb (clamp passes the gradient on the closed interval, |x| has sign(0) = 0: torch's conventions). */
int cd360_render_loss_f32(const void* fg, const void* alpha, const void* rgb, const void* op, const void* bgw, const void* mask,
                          const void* want, const void* den, void* out, int b, int hw, int S, void* stream);
int cd360_render_loss_bwd_f32(const void* fg, const void* alpha, const void* rgb, const void* op, const void* bgw, const void* mask,
                              const void* want, const void* den, const void* g, void* d_fg, void* d_alpha, void* d_rgb, int b, int hw, int S,
                              void* stream);
int cd360_nerf_pack_weights_bf16(const void* W1, const void* b1, const void* b2, const void* wv, const void* bv, const void* Wd,
                                 const void* kcol, void* out_bf16, void* out_f32, int C, int NK, void* stream);
int cd360_nerf_unpack_grads_bf16(const void* const* grads, const int* dtypes, const void* kpos, void* dW1, void* small, int C, int NK,
                                 void* stream);
/* Backward of cd360_rowdot4_bf16 (the decoder is in the reference's trainkeys `pose`, sgm/models/diffusion.py:139-144; the reference
 * differentiates nn.Linear through autograd): d [rows, 4] fp32 = gradient of the output; dh [rows, C] bf16 = d w (NULL to skip);
 * dw_part [cd360_rowdot4_bwd_slabs(rows), 4, C] fp32 = per slab of rows the partial sums of dw = d^T h (NULL to skip; the caller
 * sums the slabs in order: deterministic). */
int cd360_rowdot4_bwd_slabs(int64_t rows);
int cd360_rowdot4_bwd_bf16(const void* d, const void* h, const void* w, void* dh, void* dw_part, int64_t rows, int C, void* stream);

/* ---- GroupNorm (+SiLU) ---------------------------------------------------------------------------------------------
 * replaces GroupNorm32 -> SiLU (sgm/modules/diffusionmodules/util.py:309-311, openaimodel.py:280-283,315-318) and
 *          SpatialTransformer.norm (attention.py:118-121,833) on channels-last bf16 [N, P, C]; gamma/beta fp32 [C];
 * ws = cd360_gn_workspace_bytes(N, P, C) bytes; y may alias x.  C % 8 == 0, C <= 4096, G <= 64. */
int64_t cd360_gn_workspace_bytes(int N, int P, int C);
int cd360_gn_silu_bf16(const void* x, const void* gamma, const void* beta, void* y, void* ws, int N, int P, int C, int G, float eps,
                       int silu, const void* tile_stats, int stats_slabs, void* stream);
/* tile_stats / stats_slabs (optional, NULL / 0): per-slab channel sums [N, stats_slabs, C, 2] already produced by the conv that
 * wrote x (cd360_conv_igemm_bf16); the statistics read pass over x is then skipped. */

/* Backward of cd360_gn_silu_bf16 with respect to x (GroupNorm32 / SiLU under torch autograd in the reference's training loop):
 * x, dy, dx [N, P, C] bf16 (dx may alias dy); ws = cd360_gn_bwd_workspace_bytes(N, P, C) bytes.  No gamma / beta gradients. */
int64_t cd360_gn_bwd_workspace_bytes(int N, int P, int C);
int cd360_gn_silu_bwd_bf16(const void* x, const void* dy, const void* gamma, const void* beta, void* dx, void* ws, int N, int P, int C, int G,
                           float eps, int silu, void* stream);

/* ---- epilogues around the transformer blocks -------------------------------------------------------------------------
 * replaces GEGLU.forward's chunk / F.gelu / multiply (sgm/modules/attention.py:94-96): in [rows, 2*inner] bf16 = [x | gate]
 * -> out [rows, inner] = x * gelu(gate) (erf form).  inner % 8 == 0. */
int cd360_geglu_bf16(const void* in, void* out, int64_t rows, int inner, void* stream);
/* replaces th.cat([h, hs.pop()], dim=1) (sgm/modules/diffusionmodules/openaimodel.py:1074-1076) on channels-last activations:
 * a [pixels, ca], b [pixels, cb] -> out [pixels, ca + cb].  ca, cb % 8 == 0. */
int cd360_concat_channels_bf16(const void* a, const void* b, void* out, int64_t pixels, int ca, int cb, void* stream);

/* replaces the residual add followed by the next nn.LayerNorm in BasicTransformerBlock._forward (sgm/modules/attention.py:609-636):
 * sum_out = a + b (NULL to skip; b NULL = plain LayerNorm), ln_out = LayerNorm(a + b) * gamma + beta.  All bf16, C % 8 == 0, C <= 2048. */
int cd360_add_layernorm_bf16(const void* a, const void* b, const void* gamma, const void* beta, void* sum_out, void* ln_out, int64_t rows,
                             int C, float eps, void* stream);

/* Backward of the two epilogues above (torch autograd of GEGLU.forward / nn.LayerNorm + residual add in the reference):
 * geglu: in [rows, 2*inner] (the forward input), dy [rows, inner] -> din [rows, 2*inner] = [dx | dgate];
 * add_layernorm: x = a + b (the forward's sum_out, or a when b was NULL), d_ln, d_sum (gradient arriving on sum_out; NULL = none)
 * -> dx = da = db, all [rows, C] bf16 (dx may alias d_ln / d_sum).  No gamma / beta gradients. */
int cd360_geglu_bwd_bf16(const void* in, const void* dy, void* din, int64_t rows, int inner, void* stream);
int cd360_add_layernorm_bwd_bf16(const void* x, const void* gamma, const void* d_ln, const void* d_sum, void* dx, int64_t rows, int C, float eps,
                                 void* stream);

/* replaces the elementwise tail of one sampling step: DiscreteDenoiser's c_out/c_skip (denoiser.py:41-44), ScheduledCFGImgTextRef.__call__
 * (guiders.py:111-114), to_d and the Euler update (sampling.py:101-106).  x [n] fp32, eps [3n] fp32 (u | ic | c), sigma / sigma_next
 * device scalars; out [n] = x + (x - d0)/sigma * (sigma_next - sigma), d0 = den_u + scale (den_c - den_ic) + scale_im (den_ic - den_u). */
int cd360_cfg_euler_step_f32(const void* x, const void* eps, const void* sigma, const void* sigma_next, float scale, float scale_im,
                             void* out, int64_t n, void* stream);

/* The two ends of one CAPTURED sampling step, with every per-step scalar read from tables through a device-side step index (replaces, inside
 * the hipGraph of cd360/job.py::Sampler, the ~45 torch elementwise launches of DiscreteDenoiser.network_inputs -- denoiser.py:47-79: sigma ->
 * table index, EpsScaling, c_in x --, timestep_embedding + time_embed + label_emb + SiLU -- openaimodel.py:1006-1030, util.py:206-231 --, the
 * UNet's 4 -> 320 input convolution -- openaimodel.py:663-670 -- and the casts around the fused CFG + Euler tail).
 * cd360_unet_stage_in: x [bs, 4, H, W] fp32; step_tab [nsteps, 4] fp32 = (sigma, sigma_next, c_in, -); step: device int32 (row of the tables);
 *   w_k36 [36, Cout] fp32 = the convolution weight, k = (ky * 3 + kx) * 4 + ci; bias [Cout] fp32; h [rep * bs, H * W, Cout] bf16 out
 *   (channels-last; the `rep` CFG branches of a sample get identical rows); temb_tab [nsteps, E] bf16 = time_embed(timestep_embedding(c_noise))
 *   per step; lab [rep * bs, E] bf16 = label_emb(y); emb_act [rep * bs, E] bf16 out = silu(temb_tab[step] + lab).
 * cd360_cfg_euler_step_cl: cd360_cfg_euler_step_f32's arithmetic on x [bs, 4, HW] fp32 IN PLACE with eps [3 bs, HW, ld] bf16 channels-last
 *   (channels 0..3 of every ld-wide row: the 320 -> 4 output convolution's padded rows), sigma / sigma_next = step_tab[step][0 / 1]. */
int cd360_unet_stage_in(const void* x, const void* step_tab, const void* step, const void* w_k36, const void* bias, void* h, const void* temb_tab,
                        const void* lab, void* emb_act, int bs, int rep, int H, int W, int Cout, int E, void* stream);
int cd360_cfg_euler_step_cl(void* x, const void* eps, const void* step_tab, const void* step, float scale, float scale_im, int bs, int64_t HW,
                            int ld, void* stream);

/* The UNet's output convolution (openaimodel.py:967-973: Conv2d(model_channels -> 4, 3 x 3, padding 1) behind GroupNorm + SiLU) as a small
 * MFMA kernel of its own: x [images, H W, Cin] bf16 channels-last, w36 [64, Cin] bf16 with row tap * 4 + co = weight[co, :, ky, kx] (tap = 3 ky +
 * kx; rows 36 .. 63 zero), bias fp32 [4] -> out [images, H W, 4] bf16 (the rows cd360_cfg_euler_step_cl reads).  W in {32, 64, 128}, H even,
 * Cin % 64 == 0; other shapes: cd360_conv_igemm_bf16 with Cout padded to 16. */
int cd360_out_conv4_bf16(const void* x, const void* w36, const void* bias, void* out, int images, int H, int W, int Cin, void* stream);

/* ---- 3x3 convolution / GEMM with fused epilogue -----------------------------------------------------------------------
 * replaces nn.Conv2d(3x3, stride 1, padding 1) + the adds around it in ResBlock._forward (openaimodel.py:350-376:
 * `h + emb_out`, `skip_connection(x) + h`) and Upsample.conv (:161-164) on channels-last bf16; taps = 1 gives out = x @ w^T
 * (the 1x1 skip_connection conv, :337).  x [N*H*W, Cin]; w_packed [Cout, taps*Cin] in the kernel's K order:
 * with G = cd360_conv_k_order(Cin, taps) 64-channel chunks per group, k = ((cg*taps + ky*3+kx)*G + j)*64 + ci%64 where
 * ci/64 = cg*G + j (group outer, tap middle, chunk inner; G = Cin/64 is plain tap-major k = tap*Cin + ci); taps = 1: the plain [Cout, Cin] matrix;
 * bias fp32 [Cout] | NULL (8-byte aligned: the LDS-DMA kernels fetch its slices 16 bytes per lane); emb bf16 [N, Cout] | NULL (per-image addend);
 * res bf16 [N*H*W, Cout] | NULL; out bf16 [N*H*W, Cout].
 * Cin % 64 == 0, Cout % 16 == 0, 16-byte aligned pointers. */
int cd360_conv_k_order(int Cin, int taps);
int cd360_conv_igemm_bf16(const void* x, const void* w_packed, const void* bias, const void* emb, int64_t emb_stride, const void* res,
                          void* out, int N, int H, int W, int Cin, int Cout, int taps, int stride, void* tile_stats, void* stream);
/* stride: 1, or 2 for Downsample.op (openaimodel.py:190-213: conv3x3, stride 2, pad 1; H, W even; out is [N*(H/2)*(W/2), Cout]). */
/* emb_stride: elements between the rows of `emb` (>= Cout, multiple of 8): the time-embedding projections of all ResBlocks are
 * computed as ONE GEMM and each conv reads its column slice in place. */
/* tile_stats (optional, NULL to skip): fp32 [N*H*W/128 * cd360_conv_stats_slabs(Cout), Cout, 2] = per pixel slab and channel the
 * sum and the sum of squares of the bf16 outputs: the statistics pass of the GroupNorm that follows the conv (openaimodel.py:
 * 352-376 h = out_layers(GN -> SiLU -> conv)), handed to cd360_gn_silu_bf16.  Requires H*W % 128 == 0. */
/* replaces Upsample.forward (openaimodel.py:114-181): F.interpolate(x, scale_factor=2, mode="nearest") followed by conv3x3 / pad 1, in one
 * launch that never builds the upsampled image: output pixel (2i + a, 2j + b) reads only the 2 x 2 source pixels {i+a-1, i+a} x {j+b-1, j+b},
 * so each of the four phases (a, b) is a 2 x 2-tap convolution of the SOURCE image whose weights are the sums of the 3 x 3 taps that
 * coincide (4 / 9 of the multiply-adds).  x [N*H*W, Cin] bf16; w_phases [4, Cout, 4*Cin] bf16 (phase 2a + b; K order of
 * cd360_conv_k_order(Cin, 9) with tap slot 2 ty + tx: cd360.ops.pack_upsample_conv_weight); bias fp32 [Cout] | NULL;
 * out [N*2H*2W, Cout] bf16.  Cin % 64 == 0, Cout % 16 == 0. */
int cd360_conv_up2x_bf16(const void* x, const void* w_phases, const void* bias, void* out, int N, int H, int W, int Cin, int Cout, void* stream);
int cd360_conv_stats_slabs(int Cout);
/* Pixels per slab of `tile_stats` for this very call (the kernel that serves it decides: 64 or 128 on the LDS-DMA core that runs the
 * 3 x 3 / stride 1 convolutions, 128 / cd360_conv_stats_slabs(Cout) otherwise): tile_stats is fp32 [N*Ho*Wo / rows, Cout, 2]. */
int cd360_conv_stats_rows(int N, int H, int W, int Cin, int Cout, int taps, int stride);
/* The 3 x 3 / stride 1 / pad 1 case of cd360_conv_igemm_bf16 on the GEMM core of cd360_gemm_bf16 (operands L2 -> LDS by LDS-DMA, the
 * im2col matrix implicit: a K tile is the 64-channel chunk of the pixel shifted by the tile's tap, padding read as zeros through the
 * buffer descriptor's range check); cd360_conv_igemm_bf16 forwards to it whenever cd360_conv_dma_slab_rows(...) > 0.
 * tile_stats fp32 [N*H*W / cd360_conv_dma_slab_rows(...), Cout, 2]. */
int cd360_conv_dma_slab_rows(int N, int H, int W, int Cin, int Cout, int taps, int stride);
int cd360_conv3x3_dma_bf16(const void* x, const void* w_packed, const void* bias, const void* emb, int64_t emb_stride, const void* res,
                           void* out, int N, int H, int W, int Cin, int Cout, void* tile_stats, void* stream);

/* replaces pose_emb_layers(torch.cat([x, xref], -1)), Linear(2C -> C, bias=False) (sgm/modules/attention.py:515-516,634) without the
 * concat: out = x wa^T + xref wb^T, wa = W[:, :C], wb = W[:, C:] as contiguous [C, C] bf16; x, xref, out [rows, C] bf16, C % 64 == 0.
 * (Two GEMM-mode launches of the implicit-GEMM kernel; the shipped modules use the library GEMM for the same two products.) */
int cd360_pose_embed_bf16(const void* x, const void* xref, const void* wa, const void* wb, void* out, int64_t rows, int C, void* stream);

/* ---- Linear layers of the transformer blocks with fused epilogues -----------------------------------------------------------
 * replaces nn.Linear.forward / F.linear at sgm/modules/attention.py:323-329,368-372,422 (to_q / to_k / to_v / to_out of
 *          MemoryEfficientCrossAttention), :89-96,107-115 (GEGLU.proj, FeedForward.net[2]), :515-516,634 (pose_emb_layers),
 *          :748,786,824-826,882-884 (SpatialTransformer.proj_in / proj_out) -- and the elementwise work between them:
 *          nn.LayerNorm (:516-518 norm1-3, applied :609-636), `x * F.gelu(gate)` (:94-96) and the residual adds (:609-636,826,884).
 * out[M, N] (bf16, row stride ldo) = epilogue(A[M, K] @ W[N, K]^T): A, W bf16 with row strides lda / ldw (elements, multiples of 8),
 * K % 64 == 0, N % 16 == 0, base pointers 16-byte aligned.  Hand-scheduled MFMA kernel (LDS-DMA staging, two wave groups one
 * barrier apart), 256 x 256 or 128 x 128 tiles chosen from (M, N).
 *   bias      fp32 [N] | NULL
 *   res       bf16 [M, N], row stride ldr | NULL -- added last (the residual stream)
 *   ln_stats  fp32 [M, ln_parts, 2] | NULL -- LayerNorm over the ln_dim (= K) channels of every A row folded in FRONT of the GEMM:
 *             LN(x) W^T + b = rstd (x (gamma o W)^T - mu rowsum(gamma o W)) + (beta W^T + b); the caller passes W := gamma o W (bf16),
 *             wsum := its fp32 row sums, bias := beta W^T + b; the row statistics are the per-row partial (sum, sum of squares) another
 *             cd360_gemm_bf16 call wrote through `stats_out` (or cd360_row_stats_bf16)
 *   stats_out fp32 [M, ceil(N / cd360_gemm_tile_n(M, N)), 2] | NULL -- those partials for THIS call's bf16 output rows
 *   flags     bit 0: GEGLU -- W rows (and bias / wsum) packed per 64 rows as [32 value rows | 32 gate rows] of the same 32 output
 *             columns; out is [M, N / 2] = value * gelu(gate) (exact erf form) */
int cd360_gemm_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo, const void* bias,
                    const void* res, int64_t ldr, const void* ln_stats, int ln_parts, int ln_dim, float ln_eps, const void* wsum,
                    void* stats_out, int flags, void* stream);
/* Backward of nn.Linear on the fine-tuning path (BASELINE configs[3]; the reference trains through torch autograd of F.linear:
 * sgm/modules/attention.py:515-516,634 pose_emb_layers, sgm/modules/nerfsd_pytorch3d.py:40-51 plane_coefs / decoder).
 *   data gradient   dX[M, K] = dY[M, N] W[N, K]      = cd360_gemm_bf16(a = dY, w = W^T [K, N]) -- the same kernel on the transposed weight;
 *   weight gradient dW[N, K] = dY[M, N]^T X[M, K]    = cd360_gemm_tn_bf16 below: both operands row-major over the CONTRACTION index
 *     (LDS-DMA staging, transposing LDS reads ds_read_b64_tr_b16 for both MFMA operands, fp32 accumulation, M split over slabs of
 *     workgroups whose fp32 partial tiles are summed in a fixed order: deterministic, no atomics).
 * out[N, K] = A[M, N]^T B[M, K]; A, B bf16, row strides lda >= N, ldb >= K (elements, multiples of 8), N % 8 == 0, K % 8 == 0, 16-byte
 * aligned pointers; out_dtype 0: fp32, 1: bf16; ws = cd360_gemm_tn_workspace_bytes(M, N, K) bytes of scratch. */
int64_t cd360_gemm_tn_workspace_bytes(int64_t M, int N, int K);
int cd360_gemm_tn_bf16(const void* a, const void* b, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldb, int out_dtype, void* ws,
                       void* stream);
/* replaces to_q (nn.Linear, attention.py:323,368) + xformers.ops.memory_efficient_attention (attention.py:406) of a cross-attention whose
 * context has Nk <= 96 tokens -- attn2 over the 77 text tokens, in every transformer block (attention.py:620-625) and on the FeatureNeRF
 * pose tokens (attention.py:578-588, the north-star kernel: 98 304 queries per batch element at 1024^2): the query projection and
 * softmax(q k^T * scale) v run in ONE kernel, Q never exists in memory.  a / w / bias / ln_stats / wsum as in cd360_gemm_bf16 (optional
 * LayerNorm fold of norm2 in front of to_q); N = heads * 64; k, v bf16 [B, >= Nk, N] with element strides (batch, key), head h at
 * columns 64 h; out bf16 [M, N] (row stride ldo), M = B * Nq, Nq % 128 == 0 (tiles of 256 x 256 for tall problems, 128 x 128 when those would not fill the 256 CUs). */
int cd360_qproj_attn_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                          const void* bias, const void* ln_stats, int ln_parts, int ln_dim, float ln_eps, const void* wsum, const void* k,
                          const void* v, int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn, int Nq, int Nk, float scale, void* stream);
/* cd360_qproj_attn_bf16 for a CFG batch whose last `dup` query batch elements are each needed against TWO key / value sets (sample.py's
 * 3-way CFG: the image-conditional and the image+text-conditional thirds have identical pose tokens and differ in the text context only):
 * k, v hold B + dup batch elements (B = M / Nq), out (B + dup) * Nq rows; query element i >= B - dup writes batch i (keys of batch i) and
 * batch i + dup (keys of batch i + dup).  The projection of those queries runs once. */
int cd360_qproj_attn_dedup_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                                const void* bias, const void* ln_stats, int ln_parts, int ln_dim, float ln_eps, const void* wsum,
                                const void* k, const void* v, int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn, int Nq, int Nk,
                                float scale, int dup, void* stream);
/* BASELINE.json configs[4] -- the same fused cross-attention with q K^T and P V on fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, the only
 * fp8 form above the bf16 rate on gfx950; unit block scales, real scales applied to the fp32 scores / outputs).  The keys and values of
 * attention.py:578-588,620-625 are the text context's: constant over a trajectory, so they are quantised once per image:
 *   cd360_kv_pack_fp8: k, v bf16 [B, >= Nk, H * 64] (element strides batch / key), Nk <= 96 -> kv8 (cd360_kv_fp8_bytes(B, H) bytes, 16-byte
 *   aligned: per (batch, head) OCP e4m3 K rows and V^T rows in the layout the kernel's LDS reads want) + scales fp32 [B, H, 2] (amax / 448
 *   of that head's K and V);
 *   cd360_qproj_attn_fp8_bf16: arguments of cd360_qproj_attn_dedup_bf16 with (kv8, scales) -- B + dup batch elements -- in place of k, v
 *   and their strides; 65 <= Nk <= 96 (CD360_ERR_SHAPE otherwise: the caller keeps the bf16 kernel).  The projected query is scaled per
 *   token in registers; accumulation fp32, output bf16. */
int64_t cd360_kv_fp8_bytes(int B, int H);
int cd360_kv_pack_fp8(const void* k, const void* v, void* kv8, void* scales, int B, int H, int Nk, int64_t k_sb, int64_t k_sn, int64_t v_sb,
                      int64_t v_sn, void* stream);
int cd360_qproj_attn_fp8_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                              const void* bias, const void* ln_stats, int ln_parts, int ln_dim, float ln_eps, const void* wsum,
                              const void* kv8, const void* scales, int Nq, int Nk, float scale, int dup, void* stream);
/* N-tile width (256 | 192 | 128) cd360_gemm_bf16 uses for an [M, N] output: stats_out holds ceil(N / that) partials per row. */
int cd360_gemm_tile_n(int64_t M, int N);
/* cd360_gemm_bf16(a, w, out, ..., bias, res) for an output that a GroupNorm reads next -- SpatialTransformer.proj_out plus its residual
 * (attention.py:880-886) feeding the next ResBlock's in_layers: also writes cstats fp32 [M / S, N, 2] = per slab of S rows and channel
 * the (sum, sum of squares) of the stored bf16 outputs, i.e. the `tile_stats` of cd360_gn_silu_bf16 (as the convolution epilogue does).
 * S = cd360_gemm_cstats_rows(M, N): 64 on the 128 x 128 tilings, 32 on the 64 x 128 tiling of small batches, 0 = this shape's tiling
 * writes none (CD360_ERR_SHAPE); needs M % 64 == 0. */
int cd360_gemm_cstats_rows(int64_t M, int N);
int cd360_gemm_cstats_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                           const void* bias, const void* res, int64_t ldr, void* cstats, void* stream);
/* Fine-tuning optimiser (configs/train_co3d_concept.yaml optimizer_config: torch.optim.AdamW; the step the reference's Lightning loop
 * takes after main.py's backward): AdamW on fp32 master weights of bf16 parameters in ONE pass -- for each of n <= CD360_ADAMW_MAX_TENSORS
 * tensors reads the bf16 gradient grads[t] and the fp32 master / exp_avg / exp_avg_sq at offset begin[t] (multiple of 8) of the flat state
 * buffers, applies torch's update (decoupled weight decay wd[t], learning rate lr[t], bias correction with the device-side step count
 * *step, no amsgrad), writes the states back and OVERWRITES the bf16 parameter params[t] with the rounded master.  grads / params / begin /
 * numel / lr / wd are host arrays of n entries.  cd360_adamw_tick(step) adds 1 to *step: once per optimisation step, before the update
 * launches (the count lives on the device so that hipGraph replays advance it). */
#define CD360_ADAMW_MAX_TENSORS 64
int cd360_adamw_tick(void* step, void* stream);
int cd360_adamw_bf16(int n, const void* const* grads, void* const* params, const int64_t* begin, const int64_t* numel, const float* lr,
                     const float* wd, void* master, void* exp_avg, void* exp_avg_sq, const void* step, float beta1, float beta2, float eps,
                     void* stream);
/* (sum, sum of squares) of every row of a bf16 [rows, C] matrix (row stride ld) as one fp32 partial per row: the `ln_stats` input for a
 * tensor that did not come out of cd360_gemm_bf16 (C % 8 == 0). */
int cd360_row_stats_bf16(const void* x, void* stats, int64_t rows, int C, int64_t ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CD360_HIP_H */
