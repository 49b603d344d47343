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
