// K7: GroupNorm (fp32 statistics, as the reference's GroupNorm32: sgm/modules/diffusionmodules/util.py:309-311)
// with an optional fused SiLU, on channels-last bf16 activations [N, P=H*W, C].
// Brackets the injected transformer blocks: ResBlock in/out layers (openaimodel.py:280-328, GN -> SiLU -> conv)
// and SpatialTransformer.norm (attention.py:118-121,748, GN only, eps 1e-6).  HBM-bound: 2 reads + 1 write of the
// activation instead of the 5-6 passes of the eager float()/group_norm/type()/silu chain, and the channels-last
// result is already the "b (h w) c" token layout the transformer wants (no rearrange copy).
//   pass 1: per-(n, slab) per-channel partial sums (coalesced 16-byte reads), deterministic (no atomics)
//   pass 2: each workgroup rebuilds scale/shift per channel in LDS from the partials, then streams its slab.
#include "cd360_common.h"

namespace {

constexpr int MAX_C = 4096;

__global__ __launch_bounds__(256) void gn_partial_kernel(const uint16_t* __restrict__ x, float* __restrict__ partial, int P, int C,
                                                         int nslab) {
  // grid: (nslab, N); partial [N, nslab, C, 2].  Thread (row, cv) sums 8 channels over pixels row, row+rstep, ...
  // then the rows are added in a fixed order: bit-reproducible.
  __shared__ float red[2 * MAX_C];  // [rstep][C][2], rstep*C <= 2048 whenever rstep > 1
  const int n = blockIdx.y, slab = blockIdx.x, CV = C >> 3, tid = threadIdx.x;
  const int p0 = (int)((long)P * slab / nslab), p1 = (int)((long)P * (slab + 1) / nslab);
  const bool wide = CV > 256;
  const int rstep = wide ? 1 : 256 / CV;
  const int row = wide ? 0 : tid / CV;
  if (row < rstep) {
    for (int cv = wide ? tid : tid % CV; cv < CV; cv += wide ? 256 : CV) {
      float s[8], ss[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
      for (int pix = p0 + row; pix < p1; pix += rstep) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + ((long)n * P + pix) * C + cv * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = bf16lo_to_f32(v[e]), c = bf16hi_to_f32(v[e]);
          s[2 * e] += a; ss[2 * e] = fmaf(a, a, ss[2 * e]);
          s[2 * e + 1] += c; ss[2 * e + 1] = fmaf(c, c, ss[2 * e + 1]);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[2 * ((long)row * C + cv * 8 + e)] = s[e];
        red[2 * ((long)row * C + cv * 8 + e) + 1] = ss[e];
      }
    }
  }
  __syncthreads();
  float* dst = partial + ((long)n * nslab + slab) * 2 * C;
  for (int i = tid; i < 2 * C; i += 256) {
    float acc = 0.f;
    for (int rr = 0; rr < rstep; ++rr) acc += red[(long)rr * 2 * C + i];
    dst[i] = acc;
  }
}

// grid (G, N): one workgroup reduces the nslab x cpg x (sum, sumsq) partials of one group -- every thread a fixed stride of them with
// four independent loads in flight, an xor tree inside each wave, the four wave sums through the LDS in a fixed order: deterministic, ONE
// workgroup barrier (the eight-barrier LDS tree of rounds 1-5 was most of this kernel's 6 us: 46 launches per denoise step).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stat, int P, int C, int G,
                                                          float eps, int nslab) {
  __shared__ float red[8];
  const int g = blockIdx.x, n = blockIdx.y, cpg = C / G, tid = threadIdx.x, lane = tid & 63, items = nslab * cpg;
  const f32x2* src = reinterpret_cast<const f32x2*>(partial) + (long)n * nslab * C + g * cpg;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i0 = tid; i0 < items; i0 += 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
      if (i < items) {
        const int sl = i / cpg;
        const f32x2 v = src[(long)sl * C + (i - sl * cpg)];
        s[u] += v[0];
        ss[u] += v[1];
      }
    }
  }
  float a = (s[0] + s[1]) + (s[2] + s[3]), b = (ss[0] + ss[1]) + (ss[2] + ss[3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  if (lane == 0) {
    red[2 * (tid >> 6)] = a;
    red[2 * (tid >> 6) + 1] = b;
  }
  __syncthreads();
  if (tid == 0) {
    const float sum = (red[0] + red[2]) + (red[4] + red[6]), sq = (red[1] + red[3]) + (red[5] + red[7]);
    const float cnt = (float)cpg * (float)P;
    const float mean = sum / cnt;
    const float var = fmaxf(sq / cnt - mean * mean, 0.f);
    stat[((long)n * G + g) * 2] = mean;
    stat[((long)n * G + g) * 2 + 1] = 1.f / sqrtf(var + eps);
  }
}

template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ x, const float* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       uint16_t* __restrict__ y, int P, int C, int G, float eps, int nslab,
                                                       int nslab_apply) {
  __shared__ float ab[2 * MAX_C];  // per channel: scale, shift
  const int n = blockIdx.y, slab = blockIdx.x, cpg = C / G, CV = C >> 3;
  const float* gstat = partial + (long)n * 2 * G;  // here `partial` is the finalised [N, G, 2] = (mean, rstd)
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int g = c / cpg;
    const float a = gstat[2 * g + 1] * gamma[c];
    ab[2 * c] = a;
    ab[2 * c + 1] = beta[c] - gstat[2 * g] * a;
  }
  __syncthreads();
  const long i0 = (long)P * slab / nslab_apply * CV, i1 = (long)P * (slab + 1) / nslab_apply * CV;
  for (long it = i0 + threadIdx.x; it < i1; it += blockDim.x) {
    const long pix = it / CV;
    const int cv = (int)(it - pix * CV);
    const long off = ((long)n * P + pix) * C + cv * 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + off);
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = fmaf(bf16lo_to_f32(v[e]), ab[2 * (cv * 8 + 2 * e)], ab[2 * (cv * 8 + 2 * e) + 1]);
      o[2 * e + 1] = fmaf(bf16hi_to_f32(v[e]), ab[2 * (cv * 8 + 2 * e + 1)], ab[2 * (cv * 8 + 2 * e + 1) + 1]);
    }
    if (SILU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = o[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * o[e]));
    }
    u32x4 w = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
    *reinterpret_cast<u32x4*>(y + off) = w;
  }
}

}  // namespace

static inline int gn_num_slabs(int P) {
  const int n = (P + 63) / 64;  // ~64 pixels per workgroup: >= 256 workgroups per image at 128^2
  return n < 1 ? 1 : (n > 256 ? 256 : n);
}

extern "C" int64_t cd360_gn_workspace_bytes(int N, int P, int C) {
  return ((int64_t)N * gn_num_slabs(P) * 2 * C + (int64_t)N * 2 * 64) * 4;
}

// x, y: [N, P, C] bf16 channels-last (y may alias x); gamma, beta: [C] fp32; ws: cd360_gn_workspace_bytes(N, P, C) bytes.
// tile_stats (optional) = the per-slab channel sums a producer already computed while writing x -- cd360_conv_igemm_bf16's
// `tile_stats`, fp32 [N, stats_slabs, C, 2] with stats_slabs equal slabs per image: the statistics read pass over x is skipped.
extern "C" int cd360_gn_silu_bf16(const void* x, const void* gamma, const void* beta, void* y, void* ws, int N, int P, int C, int G,
                                  float eps, int silu, const void* tile_stats, int stats_slabs, void* stream) {
  if (!x || !gamma || !beta || !y || !ws || N <= 0 || P <= 0 || C <= 0 || G <= 0) return CD360_ERR_ARG;
  if (C % 8 || C % G || C > MAX_C || G > 64) return CD360_ERR_SHAPE;
  if (tile_stats && stats_slabs <= 0) return CD360_ERR_ARG;
  const int nslab = tile_stats ? stats_slabs : gn_num_slabs(P);
  float* partial = tile_stats ? (float*)tile_stats : (float*)ws;
  float* stat = (float*)ws + (long)N * gn_num_slabs(P) * 2 * C;
  if (!tile_stats) {
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nslab, N), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, partial, P, C, nslab);
    CD360_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, N), dim3(256), 0, (hipStream_t)stream, (const float*)partial, stat, P, C, G, eps, nslab);
  CD360_LAUNCH_CHECK();
  const int nslab_apply = (int)(((long)P * (C / 8) + 256 * 8 - 1) / (256 * 8));
  const int na = nslab_apply < 1 ? 1 : (nslab_apply > 1024 ? 1024 : nslab_apply);
  if (silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(na, N), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const float*)stat,
                       (const float*)gamma, (const float*)beta, (uint16_t*)y, P, C, G, eps, nslab, na);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(na, N), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const float*)stat,
                       (const float*)gamma, (const float*)beta, (uint16_t*)y, P, C, G, eps, nslab, na);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// ---- backward (BASELINE config 4; the reference differentiates GroupNorm32 / SiLU through torch autograd) -------------------------
// y = act(z), z = xhat * gamma + beta, xhat = (x - mean_g) * rstd_g.  With gh = dy * act'(z) * gamma and the group means
// m1 = mean_g(gh), m2 = mean_g(gh * xhat):      dx = rstd_g * (gh - m1 - xhat * m2).
// Same deterministic slab structure as the forward: statistics of x (forward kernels), per-slab channel sums of (gh, gh xhat),
// group reduction, one streaming apply pass.  No gamma / beta gradients: the shipped configs never train them.
namespace {

__device__ __forceinline__ float silu_grad(float z) {
  const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
  return sg * (1.f + z * (1.f - sg));
}

template <bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                             const float* __restrict__ stat, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ partial, int P, int C, int G,
                                                             int nslab) {
  __shared__ float red[2 * MAX_C];
  const int n = blockIdx.y, slab = blockIdx.x, CV = C >> 3, tid = threadIdx.x, cpg = C / G;
  const int p0 = (int)((long)P * slab / nslab), p1 = (int)((long)P * (slab + 1) / nslab);
  const bool wide = CV > 256;
  const int rstep = wide ? 1 : 256 / CV;
  const int row = wide ? 0 : tid / CV;
  if (row < rstep) {
    for (int cv = wide ? tid : tid % CV; cv < CV; cv += wide ? 256 : CV) {
      float mu[8], rs[8], ga[8], be[8], s[8], ss[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = cv * 8 + e, g = c / cpg;
        mu[e] = stat[((long)n * G + g) * 2];
        rs[e] = stat[((long)n * G + g) * 2 + 1];
        ga[e] = gamma[c];
        be[e] = beta[c];
        s[e] = 0.f;
        ss[e] = 0.f;
      }
      for (int pix = p0 + row; pix < p1; pix += rstep) {
        const long off = ((long)n * P + pix) * C + cv * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + off);
        const u32x4 d = *reinterpret_cast<const u32x4*>(dy + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = (e & 1) ? bf16hi_to_f32(v[e >> 1]) : bf16lo_to_f32(v[e >> 1]);
          const float dv = (e & 1) ? bf16hi_to_f32(d[e >> 1]) : bf16lo_to_f32(d[e >> 1]);
          const float xh = (xv - mu[e]) * rs[e];
          float gh = dv * ga[e];
          if (SILU) gh *= silu_grad(fmaf(xh, ga[e], be[e]));
          s[e] += gh;
          ss[e] = fmaf(gh, xh, ss[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[2 * ((long)row * C + cv * 8 + e)] = s[e];
        red[2 * ((long)row * C + cv * 8 + e) + 1] = ss[e];
      }
    }
  }
  __syncthreads();
  float* dst = partial + ((long)n * nslab + slab) * 2 * C;
  for (int i = tid; i < 2 * C; i += 256) {
    float acc = 0.f;
    for (int rr = 0; rr < rstep; ++rr) acc += red[(long)rr * 2 * C + i];
    dst[i] = acc;
  }
}

// grid (G, N): group means (m1, m2) from the per-slab channel sums
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stat2, int P, int C, int G,
                                                              int nslab) {
  __shared__ float red[2 * 256];
  const int g = blockIdx.x, n = blockIdx.y, cpg = C / G, tid = threadIdx.x;
  float s = 0.f, ss = 0.f;
  for (int i = tid; i < nslab * cpg; i += 256) {
    const int sl = i / cpg, c = g * cpg + (i - sl * cpg);
    const float* src = partial + (((long)n * nslab + sl) * C + c) * 2;
    s += src[0];
    ss += src[1];
  }
  red[2 * tid] = s;
  red[2 * tid + 1] = ss;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { red[2 * tid] += red[2 * (tid + off)]; red[2 * tid + 1] += red[2 * (tid + off) + 1]; }
    __syncthreads();
  }
  if (tid == 0) {
    const float cnt = (float)cpg * (float)P;
    stat2[((long)n * G + g) * 2] = red[0] / cnt;
    stat2[((long)n * G + g) * 2 + 1] = red[1] / cnt;
  }
}

template <bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                           const float* __restrict__ stat, const float* __restrict__ stat2,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           uint16_t* __restrict__ dx, int P, int C, int G, int nslab_apply) {
  __shared__ float gs[64 * 4];  // per group: mean, rstd, m1, m2
  const int n = blockIdx.y, slab = blockIdx.x, cpg = C / G, CV = C >> 3;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    gs[4 * g] = stat[((long)n * G + g) * 2];
    gs[4 * g + 1] = stat[((long)n * G + g) * 2 + 1];
    gs[4 * g + 2] = stat2[((long)n * G + g) * 2];
    gs[4 * g + 3] = stat2[((long)n * G + g) * 2 + 1];
  }
  __syncthreads();
  const long i0 = (long)P * slab / nslab_apply * CV, i1 = (long)P * (slab + 1) / nslab_apply * CV;
  for (long it = i0 + threadIdx.x; it < i1; it += blockDim.x) {
    const long pix = it / CV;
    const int cv = (int)(it - pix * CV);
    const long off = ((long)n * P + pix) * C + cv * 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + off);
    const u32x4 d = *reinterpret_cast<const u32x4*>(dy + off);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cv * 8 + e, g = c / cpg;
      const float xv = (e & 1) ? bf16hi_to_f32(v[e >> 1]) : bf16lo_to_f32(v[e >> 1]);
      const float dv = (e & 1) ? bf16hi_to_f32(d[e >> 1]) : bf16lo_to_f32(d[e >> 1]);
      const float ga = gamma[c];
      const float xh = (xv - gs[4 * g]) * gs[4 * g + 1];
      float gh = dv * ga;
      if (SILU) gh *= silu_grad(fmaf(xh, ga, beta[c]));
      o[e] = gs[4 * g + 1] * (gh - gs[4 * g + 2] - xh * gs[4 * g + 3]);
    }
    u32x4 w = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
    *reinterpret_cast<u32x4*>(dx + off) = w;
  }
}

}  // namespace

extern "C" int64_t cd360_gn_bwd_workspace_bytes(int N, int P, int C) {
  return ((int64_t)N * gn_num_slabs(P) * 2 * C + (int64_t)N * 4 * 64) * 4;
}

// Backward of cd360_gn_silu_bf16 with respect to x: x, dy, dx [N, P, C] bf16 channels-last (dx may alias dy); gamma, beta [C] fp32;
// ws: cd360_gn_bwd_workspace_bytes(N, P, C) bytes.
extern "C" int cd360_gn_silu_bwd_bf16(const void* x, const void* dy, const void* gamma, const void* beta, void* dx, void* ws, int N, int P,
                                      int C, int G, float eps, int silu, void* stream) {
  if (!x || !dy || !gamma || !beta || !dx || !ws || N <= 0 || P <= 0 || C <= 0 || G <= 0) return CD360_ERR_ARG;
  if (C % 8 || C % G || C > MAX_C || G > 64) return CD360_ERR_SHAPE;
  const int nslab = gn_num_slabs(P);
  float* partial = (float*)ws;
  float* stat = (float*)ws + (long)N * nslab * 2 * C;
  float* stat2 = stat + (long)N * 2 * 64;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nslab, N), dim3(256), 0, st, (const uint16_t*)x, partial, P, C, nslab);
  CD360_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, N), dim3(256), 0, st, (const float*)partial, stat, P, C, G, eps, nslab);
  CD360_LAUNCH_CHECK();
  if (silu)
    hipLaunchKernelGGL(gn_bwd_partial_kernel<true>, dim3(nslab, N), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)dy, (const float*)stat,
                       (const float*)gamma, (const float*)beta, partial, P, C, G, nslab);
  else
    hipLaunchKernelGGL(gn_bwd_partial_kernel<false>, dim3(nslab, N), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)dy, (const float*)stat,
                       (const float*)gamma, (const float*)beta, partial, P, C, G, nslab);
  CD360_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(G, N), dim3(256), 0, st, (const float*)partial, stat2, P, C, G, nslab);
  CD360_LAUNCH_CHECK();
  const int nslab_apply = (int)(((long)P * (C / 8) + 256 * 8 - 1) / (256 * 8));
  const int na = nslab_apply < 1 ? 1 : (nslab_apply > 1024 ? 1024 : nslab_apply);
  if (silu)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3(na, N), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)dy, (const float*)stat,
                       (const float*)stat2, (const float*)gamma, (const float*)beta, (uint16_t*)dx, P, C, G, na);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3(na, N), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)dy, (const float*)stat,
                       (const float*)stat2, (const float*)gamma, (const float*)beta, (uint16_t*)dx, P, C, G, na);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}
