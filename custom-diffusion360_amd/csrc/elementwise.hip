// HBM-bound epilogue kernels bracketing the transformer blocks of the UNet (16-byte accesses, grid-stride).
//   geglu          : GEGLU gate of FeedForward, x * gelu(gate) on the [rows, 2*inner] projection output
//                    (sgm/modules/attention.py:89-96) -- one pass instead of chunk + gelu + mul (3 passes, 2 temporaries)
//   concat_channels: skip-connection concat th.cat([h, hs.pop()], dim=1) (sgm/modules/diffusionmodules/openaimodel.py:1074)
//                    on channels-last activations, where it is a per-pixel interleave of two channel runs.
#include "cd360_common.h"

namespace {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

__global__ __launch_bounds__(256) void geglu_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, long rows, int inner) {
  const int cpv = inner >> 3;
  const long total = rows * cpv;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    const long row = gid / cpv;
    const int cv = (int)(gid - row * cpv);
    const uint16_t* src = in + row * 2L * inner + cv * 8;
    const u32x4 xv = *reinterpret_cast<const u32x4*>(src);
    const u32x4 gv = *reinterpret_cast<const u32x4*>(src + inner);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack_bf16x2(bf16lo_to_f32(xv[e]) * gelu_erf(bf16lo_to_f32(gv[e])), bf16hi_to_f32(xv[e]) * gelu_erf(bf16hi_to_f32(gv[e])));
    *reinterpret_cast<u32x4*>(out + row * (long)inner + cv * 8) = o;
  }
}

__global__ __launch_bounds__(256) void concat_channels_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                              uint16_t* __restrict__ out, long pixels, int ca, int cb) {
  const int va = ca >> 3, vt = (ca + cb) >> 3;
  const long total = pixels * vt;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    const long pix = gid / vt;
    const int v = (int)(gid - pix * vt);
    const u32x4 val = v < va ? *reinterpret_cast<const u32x4*>(a + pix * ca + v * 8) : *reinterpret_cast<const u32x4*>(b + pix * cb + (v - va) * 8);
    *reinterpret_cast<u32x4*>(out + pix * (long)(ca + cb) + v * 8) = val;
  }
}

inline unsigned grid_for(long total) {
  const long blocks = (total + 255) / 256;
  return (unsigned)(blocks > 256L * 16 ? 256L * 16 : (blocks < 1 ? 1 : blocks));
}

}  // namespace

// in [rows, 2*inner] bf16 = [x | gate] -> out [rows, inner] bf16 = x * gelu(gate)   (exact erf GELU, as F.gelu's default)
extern "C" int cd360_geglu_bf16(const void* in, void* out, int64_t rows, int inner, void* stream) {
  if (!in || !out || rows <= 0 || inner <= 0) return CD360_ERR_ARG;
  if (inner % 8) return CD360_ERR_SHAPE;
  hipLaunchKernelGGL(geglu_kernel, dim3(grid_for(rows * (inner / 8))), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, (uint16_t*)out,
                     (long)rows, inner);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// a [pixels, ca], b [pixels, cb] bf16 -> out [pixels, ca + cb]
extern "C" int cd360_concat_channels_bf16(const void* a, const void* b, void* out, int64_t pixels, int ca, int cb, void* stream) {
  if (!a || !b || !out || pixels <= 0 || ca <= 0 || cb <= 0) return CD360_ERR_ARG;
  if (ca % 8 || cb % 8) return CD360_ERR_SHAPE;
  hipLaunchKernelGGL(concat_channels_kernel, dim3(grid_for(pixels * ((ca + cb) / 8))), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a,
                     (const uint16_t*)b, (uint16_t*)out, (long)pixels, ca, cb);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// ---- residual add + LayerNorm in one pass (sgm/modules/attention.py:609-636: `x = attn(norm(x)) + x` followed by the next norm) ----
// sum = a + b (written once, bf16) and ln = LayerNorm(sum) * gamma + beta from the fp32 sum: 2 reads + 2 writes instead of the
// 5 passes of add-kernel + LayerNorm-kernel.  One wave per row, row held in registers (C <= 2048), two-pass statistics.
namespace {
constexpr int LN_MAX_IT = 4;  // 64 lanes x 8 channels x 4 = 2048 channels

__global__ __launch_bounds__(256) void add_layernorm_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                            const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta,
                                                            uint16_t* __restrict__ sum_out, uint16_t* __restrict__ ln_out, long rows, int C,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int nchunk = C >> 3;
  for (long row = wave0; row < rows; row += nwaves) {
    float v[LN_MAX_IT][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
      const int ch = lane + 64 * it;
      if (ch < nchunk) {
        const u32x4 av = *reinterpret_cast<const u32x4*>(a + row * C + ch * 8);
        u32x4 bv = {0u, 0u, 0u, 0u};
        if (b) bv = *reinterpret_cast<const u32x4*>(b + row * C + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[it][2 * e] = bf16lo_to_f32(av[e]) + bf16lo_to_f32(bv[e]);
          v[it][2 * e + 1] = bf16hi_to_f32(av[e]) + bf16hi_to_f32(bv[e]);
        }
        if (sum_out) {
          u32x4 o = {pack_bf16x2(v[it][0], v[it][1]), pack_bf16x2(v[it][2], v[it][3]), pack_bf16x2(v[it][4], v[it][5]),
                     pack_bf16x2(v[it][6], v[it][7])};
          *reinterpret_cast<u32x4*>(sum_out + row * C + ch * 8) = o;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[it][e];
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
      if (lane + 64 * it < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[it][e] - mean; ss = fmaf(d, d, ss); }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rstd = 1.f / sqrtf(ss / (float)C + eps);
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
      const int ch = lane + 64 * it;
      if (ch < nchunk) {
        const u32x4 gv = *reinterpret_cast<const u32x4*>(gamma + ch * 8);
        const u32x4 bv = *reinterpret_cast<const u32x4*>(beta + ch * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_bf16x2(fmaf((v[it][2 * e] - mean) * rstd, bf16lo_to_f32(gv[e]), bf16lo_to_f32(bv[e])),
                             fmaf((v[it][2 * e + 1] - mean) * rstd, bf16hi_to_f32(gv[e]), bf16hi_to_f32(bv[e])));
        *reinterpret_cast<u32x4*>(ln_out + row * C + ch * 8) = o;
      }
    }
  }
}
}  // namespace

// a, b [rows, C] bf16 (b may be NULL: plain LayerNorm of a); gamma, beta [C] bf16; sum_out [rows, C] bf16 = a + b (may be NULL);
// ln_out [rows, C] bf16 = LayerNorm(a + b) * gamma + beta.  C % 8 == 0, C <= 2048.
extern "C" int cd360_add_layernorm_bf16(const void* a, const void* b, const void* gamma, const void* beta, void* sum_out, void* ln_out,
                                        int64_t rows, int C, float eps, void* stream) {
  if (!a || !gamma || !beta || !ln_out || rows <= 0 || C <= 0) return CD360_ERR_ARG;
  if (C % 8 || C > 64 * 8 * LN_MAX_IT) return CD360_ERR_SHAPE;
  const long blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(add_layernorm_kernel, dim3((unsigned)(blocks > 256L * 32 ? 256L * 32 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)a, (const uint16_t*)b, (const uint16_t*)gamma, (const uint16_t*)beta, (uint16_t*)sum_out,
                     (uint16_t*)ln_out, (long)rows, C, eps);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// ---- tail of one 3-way-CFG Euler step: EpsScaling c_out (denoiser.py:41-44, denoiser_scaling.py:26-32),
// ScheduledCFGImgTextRef combine (guiders.py:111-114), to_d + Euler update (sampling.py:101-106, sampling_utils.py:39-40) ----
namespace {
__global__ void cfg_euler_step_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ sigma,
                                      const float* __restrict__ sigma_next, float scale, float scale_im, float* __restrict__ out, long n) {
  const float s = *sigma, sn = *sigma_next;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float xv = x[i];
    const float du = xv - s * eps[i], dic = xv - s * eps[n + i], dc = xv - s * eps[2 * n + i];
    const float d0 = du + scale * (dc - dic) + scale_im * (dic - du);
    out[i] = xv + (xv - d0) / s * (sn - s);
  }
}
}  // namespace

// x [n] fp32 latent, eps [3n] fp32 network output (uncond | image-cond | image+text-cond), sigma / sigma_next: device scalars
extern "C" int cd360_cfg_euler_step_f32(const void* x, const void* eps, const void* sigma, const void* sigma_next, float scale,
                                        float scale_im, void* out, int64_t n, void* stream) {
  if (!x || !eps || !sigma || !sigma_next || !out || n <= 0) return CD360_ERR_ARG;
  hipLaunchKernelGGL(cfg_euler_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)eps,
                     (const float*)sigma, (const float*)sigma_next, scale, scale_im, (float*)out, (long)n);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// ---- backward of the elementwise epilogues (BASELINE config 4: the reference differentiates them through torch autograd) ----------
namespace {

// d(x * gelu(g)): dx = dy gelu(g), dg = dy x gelu'(g), gelu'(g) = Phi(g) + g phi(g)
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ dy, uint16_t* __restrict__ din,
                                                        long rows, int inner) {
  const int cpv = inner >> 3;
  const long total = rows * cpv;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    const long row = gid / cpv;
    const int cv = (int)(gid - row * cpv);
    const long ioff = row * 2L * inner + cv * 8;
    const u32x4 xv = *reinterpret_cast<const u32x4*>(in + ioff);
    const u32x4 gv = *reinterpret_cast<const u32x4*>(in + ioff + inner);
    const u32x4 dv = *reinterpret_cast<const u32x4*>(dy + row * (long)inner + cv * 8);
    float dx[8], dg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (e & 1) ? bf16hi_to_f32(xv[e >> 1]) : bf16lo_to_f32(xv[e >> 1]);
      const float g = (e & 1) ? bf16hi_to_f32(gv[e >> 1]) : bf16lo_to_f32(gv[e >> 1]);
      const float d = (e & 1) ? bf16hi_to_f32(dv[e >> 1]) : bf16lo_to_f32(dv[e >> 1]);
      const float cdf = 0.5f * (1.f + erff(g * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * expf(-0.5f * g * g);
      dx[e] = d * g * cdf;
      dg[e] = d * x * (cdf + g * pdf);
    }
    const u32x4 ox = {pack_bf16x2(dx[0], dx[1]), pack_bf16x2(dx[2], dx[3]), pack_bf16x2(dx[4], dx[5]), pack_bf16x2(dx[6], dx[7])};
    const u32x4 og = {pack_bf16x2(dg[0], dg[1]), pack_bf16x2(dg[2], dg[3]), pack_bf16x2(dg[4], dg[5]), pack_bf16x2(dg[6], dg[7])};
    *reinterpret_cast<u32x4*>(din + ioff) = ox;
    *reinterpret_cast<u32x4*>(din + ioff + inner) = og;
  }
}

// LayerNorm backward with the residual stream's own gradient added: dx = rstd (gh - mean(gh) - xhat mean(gh xhat)) + d_sum,
// gh = d_ln * gamma.  One wave per row, the row in registers (C <= 2048), statistics recomputed from x (two-pass, as the forward).
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gamma,
                                                                const uint16_t* __restrict__ d_ln, const uint16_t* __restrict__ d_sum,
                                                                uint16_t* __restrict__ dx, long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int nchunk = C >> 3;
  for (long row = wave0; row < rows; row += nwaves) {
    float v[LN_MAX_IT][8], gh[LN_MAX_IT][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
      const int ch = lane + 64 * it;
      if (ch < nchunk) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + row * C + ch * 8);
        const u32x4 dv = *reinterpret_cast<const u32x4*>(d_ln + row * C + ch * 8);
        const u32x4 gv = *reinterpret_cast<const u32x4*>(gamma + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[it][2 * e] = bf16lo_to_f32(xv[e]);
          v[it][2 * e + 1] = bf16hi_to_f32(xv[e]);
          gh[it][2 * e] = bf16lo_to_f32(dv[e]) * bf16lo_to_f32(gv[e]);
          gh[it][2 * e + 1] = bf16hi_to_f32(dv[e]) * bf16hi_to_f32(gv[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[it][e];
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
      if (lane + 64 * it < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[it][e] - mean; ss = fmaf(d, d, ss); }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rstd = 1.f / sqrtf(ss / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
      if (lane + 64 * it < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[it][e] = (v[it][e] - mean) * rstd;  // xhat
          m1 += gh[it][e];
          m2 = fmaf(gh[it][e], v[it][e], m2);
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { m1 += __shfl_xor(m1, off); m2 += __shfl_xor(m2, off); }
    m1 /= (float)C;
    m2 /= (float)C;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
      const int ch = lane + 64 * it;
      if (ch < nchunk) {
        u32x4 rv = {0u, 0u, 0u, 0u};
        if (d_sum) rv = *reinterpret_cast<const u32x4*>(d_sum + row * C + ch * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_bf16x2(rstd * (gh[it][2 * e] - m1 - v[it][2 * e] * m2) + bf16lo_to_f32(rv[e]),
                             rstd * (gh[it][2 * e + 1] - m1 - v[it][2 * e + 1] * m2) + bf16hi_to_f32(rv[e]));
        *reinterpret_cast<u32x4*>(dx + row * C + ch * 8) = o;
      }
    }
  }
}
}  // namespace

// Backward of cd360_geglu_bf16: in [rows, 2*inner] (the forward input), dy [rows, inner] -> din [rows, 2*inner] = [dx | dgate], bf16.
extern "C" int cd360_geglu_bwd_bf16(const void* in, const void* dy, void* din, int64_t rows, int inner, void* stream) {
  if (!in || !dy || !din || rows <= 0 || inner <= 0) return CD360_ERR_ARG;
  if (inner % 8) return CD360_ERR_SHAPE;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(rows * (inner / 8))), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in,
                     (const uint16_t*)dy, (uint16_t*)din, (long)rows, inner);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// Backward of cd360_add_layernorm_bf16 with respect to the summed input x = a + b (da = db = dx): x, d_ln [rows, C] bf16, gamma [C]
// bf16, d_sum [rows, C] bf16 = the gradient arriving on sum_out (may be NULL) -> dx [rows, C] bf16 (may alias d_ln or d_sum).
extern "C" int cd360_add_layernorm_bwd_bf16(const void* x, const void* gamma, const void* d_ln, const void* d_sum, void* dx, int64_t rows,
                                            int C, float eps, void* stream) {
  if (!x || !gamma || !d_ln || !dx || rows <= 0 || C <= 0) return CD360_ERR_ARG;
  if (C % 8 || C > 64 * 8 * LN_MAX_IT) return CD360_ERR_SHAPE;
  const long blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(add_layernorm_bwd_kernel, dim3((unsigned)(blocks > 256L * 32 ? 256L * 32 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x, (const uint16_t*)gamma, (const uint16_t*)d_ln, (const uint16_t*)d_sum, (uint16_t*)dx, (long)rows, C,
                     eps);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The operands the fused FeatureNeRF render reads, derived from TRAINABLE parameters (fine-tuning: trainkeys = pose, diffusion.py:139-144)
// in one launch, and the way back for their gradients in another.  plane_coefs.0.weight W1 [C, C + 198] is read by three GEMM-side
// operands: Wf = W1[:, :C] (features), Wk = the 99 xyz-encoding columns permuted into the kernel's 112-wide k order (cd360/nerf.py
// xyz_k_columns), Wp = the 99 Plucker / direction columns padded to 128; nviews.weight splits into the feature part vf and the camera part
// v_cam; biases and the decoder go to fp32.  Recorded op by op this was ~13 torch kernels forward and ~14 backward per pose block.
namespace {

struct NerfPackArgs {
  const uint16_t *W1, *b1, *b2, *wv, *bv, *Wd;  // bf16 parameters
  const int* kcol;                               // [NK]: column of W1 behind k-input j of the kernel, -1 = zero pad
  uint16_t* wb;                                  // bf16 out: Wf [C, C] | Wk [C, NK] | Wp [C, 128]
  float* wf;                                     // fp32 out: b1 [C] | b2 [C] | vf [C] | v_cam [99] + 1 pad | bv [1] + 3 pad | Wd [4, C]
  int C, NK;
};

__global__ __launch_bounds__(256) void nerf_pack_weights_kernel(NerfPackArgs a) {
  const int C = a.C, NK = a.NK, ld = C + 198;
  const long nWf = (long)C * C, nWk = (long)C * NK, nWp = (long)C * 128, nB = nWf + nWk + nWp, nF = 7L * C + 104;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nB + nF; i += (long)gridDim.x * blockDim.x) {
    if (i < nWf) {
      const long r = i / C, c = i - r * C;
      a.wb[i] = a.W1[r * ld + c];
    } else if (i < nWf + nWk) {
      const long k = i - nWf, r = k / NK;
      const int src = a.kcol[k - r * NK];
      a.wb[i] = src < 0 ? (uint16_t)0 : a.W1[r * ld + src];
    } else if (i < nB) {
      const long k = i - nWf - nWk, r = k >> 7;
      const int j = (int)(k & 127);
      a.wb[i] = j < 99 ? a.W1[r * ld + C + 99 + j] : (uint16_t)0;
    } else {
      const long k = i - nB;
      float v = 0.f;
      if (k < C) v = bf16_to_f32(a.b1[k]);
      else if (k < 2L * C) v = bf16_to_f32(a.b2[k - C]);
      else if (k < 3L * C) v = bf16_to_f32(a.wv[k - 2L * C]);
      else if (k < 3L * C + 99) v = bf16_to_f32(a.wv[C + 99 + (k - 3L * C)]);
      else if (k == 3L * C + 100) v = bf16_to_f32(a.bv[0]);
      else if (k >= 3L * C + 104) v = bf16_to_f32(a.Wd[k - 3L * C - 104]);
      a.wf[k] = v;
    }
  }
}

struct NerfUnpackArgs {
  const void *dWf, *dWk, *dWp, *db1, *db2, *dvf, *dvc, *dbv, *dWd;  // gradients of the packed operands (NULL = none arrived)
  int dt[9];                                                         // 0 fp32, 1 bf16
  const int* kpos;                                                   // [C + 198]: k-input fed by W1 column c, -1 = none
  uint16_t* dW1;                                                     // bf16 out [C, C + 198]
  uint16_t* small;                                                   // bf16 out: db1 [C] | db2 [C] | dwv [C + 198] | dbv [1] + 1 pad | dWd [4, C]
  int C, NK;
};

__device__ __forceinline__ float ld_grad(const void* p, int dt, long i) {
  if (!p) return 0.f;
  return dt ? bf16_to_f32(reinterpret_cast<const uint16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}

__global__ __launch_bounds__(256) void nerf_unpack_grads_kernel(NerfUnpackArgs a) {
  const int C = a.C, NK = a.NK, ld = C + 198;
  const long nW = (long)C * ld, nS = 7L * C + 200;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nW + nS; i += (long)gridDim.x * blockDim.x) {
    if (i < nW) {
      const long r = i / ld;
      const int c = (int)(i - r * ld);
      float v = 0.f;
      if (c < C) v = ld_grad(a.dWf, a.dt[0], r * C + c);
      else if (c >= C + 99) v = ld_grad(a.dWp, a.dt[2], r * 128 + (c - C - 99));
      else {
        const int k = a.kpos[c];
        if (k >= 0) v = ld_grad(a.dWk, a.dt[1], r * NK + k);
      }
      a.dW1[i] = f32_to_bf16(v);
    } else {
      const long k = i - nW;
      float v = 0.f;
      if (k < C) v = ld_grad(a.db1, a.dt[3], k);
      else if (k < 2L * C) v = ld_grad(a.db2, a.dt[4], k - C);
      else if (k < 3L * C) v = ld_grad(a.dvf, a.dt[5], k - 2L * C);
      else if (k < 3L * C + 99) v = 0.f;  // the xyz-encoding columns of nviews.weight: the view logits never read them
      else if (k < 3L * C + 198) v = ld_grad(a.dvc, a.dt[6], k - 3L * C - 99);
      else if (k == 3L * C + 198) v = ld_grad(a.dbv, a.dt[7], 0);
      else if (k >= 3L * C + 200) v = ld_grad(a.dWd, a.dt[8], k - 3L * C - 200);
      a.small[k] = f32_to_bf16(v);
    }
  }
}

}  // namespace

// W1 [C, C + 198], b1 [C], b2 [C], wv [C + 198], bv [1], Wd [4, C]: bf16, contiguous.  kcol [NK] int32 (device).  Outputs as in NerfPackArgs.
extern "C" int cd360_nerf_pack_weights_bf16(const void* W1, const void* b1, const void* b2, const void* wv, const void* bv, const void* Wd,
                                            const void* kcol, void* out_bf16, void* out_f32, int C, int NK, void* stream) {
  if (!W1 || !b1 || !b2 || !wv || !bv || !Wd || !kcol || !out_bf16 || !out_f32 || C <= 0 || NK <= 0) return CD360_ERR_ARG;
  if (C % 4) return CD360_ERR_SHAPE;
  NerfPackArgs a{(const uint16_t*)W1, (const uint16_t*)b1, (const uint16_t*)b2, (const uint16_t*)wv, (const uint16_t*)bv, (const uint16_t*)Wd,
                 (const int*)kcol, (uint16_t*)out_bf16, (float*)out_f32, C, NK};
  const long total = (long)C * (C + NK + 128) + 7L * C + 104;
  const long nblk = (total + 255) / 256;
  hipLaunchKernelGGL(nerf_pack_weights_kernel, dim3((unsigned)(nblk > 4096 ? 4096 : nblk)), dim3(256), 0, (hipStream_t)stream, a);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// grads[9] = dWf [C, C], dWk [C, NK], dWp [C, 128], db1 [C], db2 [C], dvf [C], dvc [99], dbv [1], dWd [4, C] (NULL = zero), dtypes[9]
// (0 fp32, 1 bf16); kpos [C + 198] int32 (device).  Outputs: dW1 [C, C + 198] bf16 and the small arena of NerfUnpackArgs.
extern "C" int cd360_nerf_unpack_grads_bf16(const void* const* grads, const int* dtypes, const void* kpos, void* dW1, void* small, int C, int NK,
                                            void* stream) {
  if (!grads || !dtypes || !kpos || !dW1 || !small || C <= 0 || NK <= 0) return CD360_ERR_ARG;
  NerfUnpackArgs a{};
  a.dWf = grads[0]; a.dWk = grads[1]; a.dWp = grads[2]; a.db1 = grads[3]; a.db2 = grads[4]; a.dvf = grads[5]; a.dvc = grads[6];
  a.dbv = grads[7]; a.dWd = grads[8];
  for (int i = 0; i < 9; ++i) {
    if (dtypes[i] != 0 && dtypes[i] != 1) return CD360_ERR_ARG;
    a.dt[i] = dtypes[i];
  }
  a.kpos = (const int*)kpos; a.dW1 = (uint16_t*)dW1; a.small = (uint16_t*)small; a.C = C; a.NK = NK;
  const long total = (long)C * (C + 198) + 7L * C + 200;
  const long nblk = (total + 255) / 256;
  hipLaunchKernelGGL(nerf_unpack_grads_kernel, dim3((unsigned)(nblk > 4096 ? 4096 : nblk)), dim3(256), 0, (hipStream_t)stream, a);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The three rendering terms of the fine-tuning loss for ONE pose block (sgm/modules/diffusionmodules/loss.py:188-207 of the reference:
// foreground, background and rgb terms of StandardDiffusionLossImgRef.get_loss), forward and backward.  In torch each block costs ~14
// elementwise / reduce kernels forward and ~20 backward on tensors of a few thousand values; twelve blocks per step.
//   l_fg[b]  = mean_k (clamp(fg[b,k], 0, 1) - op[b,k])^2
//   l_bg[b]  = mean_{k,s} |alpha[b,k,s] - op[b,k]| * bgw[b,k]            bgw = (1 - op) [op < 0.1], precomputed by the caller
//   l_rgb[b] = sum_{c,k} (want[b,c,k] - rgb[b,k,c])^2 mask[b,k] / den[b]
// One workgroup per batch element (fixed summation order: deterministic).
namespace {

__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void render_loss_kernel(const float* __restrict__ fg, const float* __restrict__ alpha, const float* __restrict__ rgb,
                                                          const float* __restrict__ op, const float* __restrict__ bgw,
                                                          const float* __restrict__ mask, const float* __restrict__ want,
                                                          const float* __restrict__ den, float* __restrict__ out, int hw, int S) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float sfg = 0.f, sbg = 0.f, srgb = 0.f;
  for (int k = threadIdx.x; k < hw; k += 256) {
    const float o = op[(long)b * hw + k];
    const float f = fminf(fmaxf(fg[(long)b * hw + k], 0.f), 1.f) - o;
    sfg += f * f;
    if (rgb) {
      const float m = mask[(long)b * hw + k];
      float e = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = want[((long)b * 3 + c) * hw + k] - rgb[((long)b * hw + k) * 3 + c];
        e += d * d;
      }
      srgb += e * m;
    }
  }
  for (long i = threadIdx.x; i < (long)hw * S; i += 256) {
    const long k = i / S;
    sbg += fabsf(alpha[(long)b * hw * S + i] - op[(long)b * hw + k]) * bgw[(long)b * hw + k];
  }
  sfg = block_sum256(sfg, red);
  sbg = block_sum256(sbg, red);
  srgb = block_sum256(srgb, red);
  if (threadIdx.x == 0) {
    out[b * 3] = sfg / (float)hw;
    out[b * 3 + 1] = sbg / ((float)hw * (float)S);
    out[b * 3 + 2] = rgb ? srgb / den[b] : 0.f;
  }
}

__global__ __launch_bounds__(256) void render_loss_bwd_kernel(const float* __restrict__ fg, const float* __restrict__ alpha, const float* __restrict__ rgb,
                                                              const float* __restrict__ op, const float* __restrict__ bgw,
                                                              const float* __restrict__ mask, const float* __restrict__ want,
                                                              const float* __restrict__ den, const float* __restrict__ g, float* __restrict__ d_fg,
                                                              float* __restrict__ d_alpha, float* __restrict__ d_rgb, int nb, int hw, int S) {
  const long per = (long)hw * (S + 1 + (rgb ? 3 : 0));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nb * per; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    long j = i - b * per;
    if (j < hw) {  // torch's clamp passes the gradient where min <= x <= max
      const float x = fg[(long)b * hw + j];
      const float f = fminf(fmaxf(x, 0.f), 1.f) - op[(long)b * hw + j];
      d_fg[(long)b * hw + j] = (x >= 0.f && x <= 1.f) ? g[b * 3] * 2.f * f / (float)hw : 0.f;
    } else if (j < (long)hw * (S + 1)) {
      j -= hw;
      const long k = j / S;
      const float d = alpha[(long)b * hw * S + j] - op[(long)b * hw + k];
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      d_alpha[(long)b * hw * S + j] = g[b * 3 + 1] * sg * bgw[(long)b * hw + k] / ((float)hw * (float)S);
    } else {
      j -= (long)hw * (S + 1);
      const long k = j / 3;
      const int c = (int)(j - k * 3);
      const float d = want[((long)b * 3 + c) * hw + k] - rgb[(long)b * hw * 3 + j];
      // a zero upstream gradient means the term was dropped from the total (combine_losses' device-side `loss_rgb.mean() > 0` select, the
      // reference's host-side `if`, diffusion.py:236-240): write an exact zero, not 0 * d -- d is NaN where the prediction is
      const float gg = g[b * 3 + 2];
      d_rgb[(long)b * hw * 3 + j] = gg == 0.f ? 0.f : gg * (-2.f) * d * mask[(long)b * hw + k] / den[b];
    }
  }
}

}  // namespace

// fg [b, hw], alpha [b, hw, S], rgb [b, hw, 3] | NULL, op / bgw / mask [b, hw], want [b, 3, hw], den [b]: fp32 -> out [b, 3] = (l_fg, l_bg, l_rgb)
extern "C" int cd360_render_loss_f32(const void* fg, const void* alpha, const void* rgb, const void* op, const void* bgw, const void* mask,
                                     const void* want, const void* den, void* out, int b, int hw, int S, void* stream) {
  if (!fg || !alpha || !op || !bgw || !out || b <= 0 || hw <= 0 || S <= 0) return CD360_ERR_ARG;
  if (rgb && (!mask || !want || !den)) return CD360_ERR_ARG;
  hipLaunchKernelGGL(render_loss_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, (const float*)fg, (const float*)alpha, (const float*)rgb,
                     (const float*)op, (const float*)bgw, (const float*)mask, (const float*)want, (const float*)den, (float*)out, hw, S);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// g [b, 3] = the gradient arriving on out -> d_fg [b, hw], d_alpha [b, hw, S], d_rgb [b, hw, 3] (with rgb)
extern "C" int cd360_render_loss_bwd_f32(const void* fg, const void* alpha, const void* rgb, const void* op, const void* bgw, const void* mask,
                                         const void* want, const void* den, const void* g, void* d_fg, void* d_alpha, void* d_rgb, int b, int hw,
                                         int S, void* stream) {
  if (!fg || !alpha || !op || !bgw || !g || !d_fg || !d_alpha || b <= 0 || hw <= 0 || S <= 0) return CD360_ERR_ARG;
  if (rgb && (!mask || !want || !den || !d_rgb)) return CD360_ERR_ARG;
  const long total = (long)b * hw * (S + 1 + (rgb ? 3 : 0));
  const long nblk = (total + 255) / 256;
  hipLaunchKernelGGL(render_loss_bwd_kernel, dim3((unsigned)(nblk > 2048 ? 2048 : nblk)), dim3(256), 0, (hipStream_t)stream, (const float*)fg,
                     (const float*)alpha, (const float*)rgb, (const float*)op, (const float*)bgw, (const float*)mask, (const float*)want,
                     (const float*)den, (const float*)g, (float*)d_fg, (float*)d_alpha, (float*)d_rgb, b, hw, S);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}
