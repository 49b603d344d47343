// A10: _TruncExp forward + VolRender (sgm/modules/attention.py:192-199,590-594; sgm/modules/nerfsd_pytorch3d.py:170-231).
//   sigma = exp(sigma_raw); dd = dists*sigma; alpha = 1-exp(-dd); Tr = exp(-excl_cumsum(dd)); w = nan_to_num(alpha*Tr)
//   rendered = sum_s w*feat ; fg = sum_s w ; rgb = sum_s w*sigmoid(rgb_raw)
// HBM-bound scan over S samples per ray: one work item = (ray, 8 channels); the S weights are recomputed per item
// (S=24 scalar exps, negligible next to S x 16-byte feature reads) so the features are streamed exactly once.
#include "cd360_common.h"

namespace {

constexpr int MAX_S = 64;

template <bool BF16>
__global__ void volrender_kernel(const void* __restrict__ feats_, const float* __restrict__ sigma_raw, const float* __restrict__ rgb_raw,
                                 const float* __restrict__ dists, int d_ray_stride, void* __restrict__ rendered_, float* __restrict__ fg,
                                 float* __restrict__ alphas, float* __restrict__ weights, float* __restrict__ rgb, int b, int hw, int S,
                                 int C, int flags) {
  constexpr int VEC = BF16 ? 8 : 4;
  const int cpv = C / VEC;
  const long nrays = (long)b * hw;
  const long total = nrays * cpv;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    const long ray = gid / cpv;
    const int cv = (int)(gid - ray * cpv);
    const int k = (int)(ray % hw);
    const float* sr = sigma_raw + ray * S;
    const float* dr = dists + (long)k * d_ray_stride;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    float cum = 0.f, fgsum = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int s = 0; s < S; ++s) {
      const float dd = dr[s] * ((flags & 1) ? sr[s] : expf(sr[s]));
      const float alpha = 1.f - expf(-dd);
      float w = alpha * expf(-cum);
      cum = cum + dd;
      if (w != w) w = 0.f;
      w = fminf(fmaxf(w, -3.402823466e38f), 3.402823466e38f);
      const long foff = (ray * S + s) * (long)C + (long)cv * VEC;
      if (BF16) {
        const u32x4 v = *reinterpret_cast<const u32x4*>((const uint16_t*)feats_ + foff);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] = acc[2 * e] + w * bf16lo_to_f32(v[e]);
          acc[2 * e + 1] = acc[2 * e + 1] + w * bf16hi_to_f32(v[e]);
        }
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>((const float*)feats_ + foff);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] + w * v[e];
      }
      if (cv == 0) {
        fgsum = fgsum + w;
        if (alphas) alphas[ray * S + s] = alpha;
        if (weights) weights[ray * S + s] = w;
        if (rgb_raw) {
          const float* rr = rgb_raw + (ray * S + s) * 3;
          if (flags & 2) {
            c0 = c0 + w * rr[0]; c1 = c1 + w * rr[1]; c2 = c2 + w * rr[2];
          } else {
            c0 = c0 + w * (1.f / (1.f + expf(-rr[0])));
            c1 = c1 + w * (1.f / (1.f + expf(-rr[1])));
            c2 = c2 + w * (1.f / (1.f + expf(-rr[2])));
          }
        }
      }
    }
    const long ooff = ray * (long)C + (long)cv * VEC;
    if (BF16) {
      u32x4 o = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
      *reinterpret_cast<u32x4*>((uint16_t*)rendered_ + ooff) = o;
    } else {
      f32x4 o = {acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4*>((float*)rendered_ + ooff) = o;
    }
    if (cv == 0) {
      if (fg) fg[ray] = fgsum;
      if (rgb && rgb_raw) { rgb[ray * 3] = c0; rgb[ray * 3 + 1] = c1; rgb[ray * 3 + 2] = c2; }
    }
  }
}

}  // namespace

// feats [b, hw, S, C] (dtype 0 fp32 / 1 bf16), sigma_raw [b, hw, S] fp32 (pre-exp), rgb_raw [b, hw, S, 3] fp32 (pre-sigmoid) or NULL,
// dists [hw, S] (d_ray_stride = S) or [S] (d_ray_stride = 0).  flags: bit0 = sigma_raw is already exp'ed, bit1 = rgb_raw is already sigmoid'ed.
// out: rendered [b, hw, C] (same dtype as feats), fg [b, hw], alphas [b, hw, S], weights [b, hw, S], rgb [b, hw, 3] (fp32; any may be NULL)
extern "C" int cd360_volrender(const void* feats, const void* sigma_raw, const void* rgb_raw, const void* dists, int d_ray_stride,
                               void* rendered, void* fg, void* alphas, void* weights, void* rgb, int b, int hw, int S, int C, int dtype,
                               int flags, void* stream) {
  if (!feats || !sigma_raw || !dists || !rendered || b <= 0 || hw <= 0 || S <= 0 || C <= 0) return CD360_ERR_ARG;
  if (S > MAX_S || (dtype != 0 && dtype != 1) || C % (dtype ? 8 : 4)) return CD360_ERR_SHAPE;
  if (d_ray_stride != 0 && d_ray_stride != S) return CD360_ERR_SHAPE;
  const long total = (long)b * hw * (C / (dtype ? 8 : 4));
  const unsigned blocks = (unsigned)((total + 255) / 256 > 256 * 32 ? 256 * 32 : (total + 255) / 256);
  if (dtype)
    hipLaunchKernelGGL(volrender_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feats, (const float*)sigma_raw,
                       (const float*)rgb_raw, (const float*)dists, d_ray_stride, rendered, (float*)fg, (float*)alphas, (float*)weights,
                       (float*)rgb, b, hw, S, C, flags);
  else
    hipLaunchKernelGGL(volrender_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feats, (const float*)sigma_raw,
                       (const float*)rgb_raw, (const float*)dists, d_ray_stride, rendered, (float*)fg, (float*)alphas, (float*)weights,
                       (float*)rgb, b, hw, S, C, flags);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// ---- backward of cd360_volrender (the reference differentiates VolRender.forward and _TruncExp through torch autograd;
// _TruncExp.backward = g * exp(clamp(x, -15, 15)), attention.py:203-207).  One wave per ray: lane s owns sample s of the scan
// (S <= 64), channels are spread over the lanes for the feature pass, which reads feats once and writes d_feats once.
//   w_s = alpha_s T_s,  alpha_s = 1 - e^{-dd_s},  T_s = e^{-sum_{j<s} dd_j},  dd = dists * sigma
//   dw_s = <d_rendered, feat_s> + d_fg + <d_rgb, col_s> + d_weights_s          (0 where nan_to_num replaced w)
//   d(dd)_j = dw_j T_{j+1} - sum_{s>j} dw_s w_s + d_alphas_j e^{-dd_j}
namespace {
template <bool BF16>
__global__ __launch_bounds__(256) void volrender_bwd_kernel(const void* __restrict__ feats_, const float* __restrict__ sigma_raw,
                                                            const float* __restrict__ rgb_raw, const float* __restrict__ dists, int d_ray_stride,
                                                            const void* __restrict__ d_rendered_, const float* __restrict__ d_fg,
                                                            const float* __restrict__ d_alphas, const float* __restrict__ d_weights,
                                                            const float* __restrict__ d_rgb, void* __restrict__ d_feats_,
                                                            float* __restrict__ d_sigma_raw, float* __restrict__ d_rgb_raw, long nrays, int hw,
                                                            int S, int C, int flags) {
  constexpr int VEC = BF16 ? 8 : 4;
  const int lane = threadIdx.x & 63;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long ray = wave0; ray < nrays; ray += nwaves) {
    const int k = (int)(ray % hw);
    // ---- the scan, lane s = sample s ----
    float dd = 0.f, sig = 0.f, raw = 0.f, dist = 0.f;
    if (lane < S) {
      raw = sigma_raw[ray * S + lane];
      dist = dists[(long)k * d_ray_stride + lane];
      sig = (flags & 1) ? raw : expf(raw);
      dd = dist * sig;
    }
    float incl = dd;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    float excl = __shfl_up(incl, 1);  // (not incl - dd: one huge dd would cancel the small prefix away)
    if (lane == 0) excl = 0.f;
    const float e_own = expf(-dd), alpha = 1.f - e_own, t_next = expf(-incl), t_cur = expf(-excl);
    float w = alpha * t_cur;
    const bool dead = (w != w) || lane >= S;  // nan_to_num replaced a NaN by the constant 0 (no gradient); +-inf clamps likewise
    if (dead) w = 0.f;
    const bool clamped = fabsf(w) > 3.402823466e38f;
    w = fminf(fmaxf(w, -3.402823466e38f), 3.402823466e38f);

    // ---- feature pass: d_feats = w_s d_rendered, dot_s = <d_rendered, feat_s> ----
    float dot_own = 0.f;
    for (int s = 0; s < S; ++s) {
      const float ws = __shfl(w, s);
      float part = 0.f;
      for (int c = lane * VEC; c < C; c += 64 * VEC) {
        const long foff = (ray * S + s) * (long)C + c, roff = ray * (long)C + c;
        if (BF16) {
          const u32x4 f = *reinterpret_cast<const u32x4*>((const uint16_t*)feats_ + foff);
          const u32x4 g = *reinterpret_cast<const u32x4*>((const uint16_t*)d_rendered_ + roff);
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g0 = bf16lo_to_f32(g[e]), g1 = bf16hi_to_f32(g[e]);
            part += g0 * bf16lo_to_f32(f[e]) + g1 * bf16hi_to_f32(f[e]);
            o[e] = pack_bf16x2(ws * g0, ws * g1);
          }
          *reinterpret_cast<u32x4*>((uint16_t*)d_feats_ + foff) = o;
        } else {
          const f32x4 f = *reinterpret_cast<const f32x4*>((const float*)feats_ + foff);
          const f32x4 g = *reinterpret_cast<const f32x4*>((const float*)d_rendered_ + roff);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            part += g[e] * f[e];
            o[e] = ws * g[e];
          }
          *reinterpret_cast<f32x4*>((float*)d_feats_ + foff) = o;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
      if (lane == s) dot_own = part;
    }

    // ---- per-sample gradients, lane s = sample s ----
    float dw = dot_own;
    if (d_fg) dw += d_fg[ray];
    if (d_weights && lane < S) dw += d_weights[ray * S + lane];
    if (rgb_raw && d_rgb && lane < S) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float rr = rgb_raw[(ray * S + lane) * 3 + j], gj = d_rgb[ray * 3 + j];
        const float col = (flags & 2) ? rr : 1.f / (1.f + expf(-rr));
        dw += gj * col;
        if (d_rgb_raw) d_rgb_raw[(ray * S + lane) * 3 + j] = w * gj * ((flags & 2) ? 1.f : col * (1.f - col));
      }
    } else if (d_rgb_raw && lane < S) {
#pragma unroll
      for (int j = 0; j < 3; ++j) d_rgb_raw[(ray * S + lane) * 3 + j] = 0.f;
    }
    if (dead || clamped) dw = 0.f;
    const float dww = dw * w;
    float suf = dww;  // inclusive suffix sum of dw_s w_s (a direct reverse scan: total - prefix would cancel)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float t = __shfl_down(suf, off);
      if (lane + off < 64) suf += t;
    }
    float after = __shfl_down(suf, 1);  // sum over s > lane
    if (lane == 63) after = 0.f;
    float ddd = dw * t_next - after;
    if (d_alphas && lane < S) ddd += d_alphas[ray * S + lane] * e_own;
    if (lane < S) {
      float gs = ddd * dist;
      if (!(flags & 1)) gs *= expf(fminf(fmaxf(raw, -15.f), 15.f));
      d_sigma_raw[ray * S + lane] = gs;
    }
  }
}
}  // namespace

// Backward of cd360_volrender: same feats / sigma_raw / rgb_raw / dists / flags as the forward call; incoming gradients d_rendered
// [b, hw, C] (feats' dtype; required), d_fg [b, hw], d_alphas / d_weights [b, hw, S], d_rgb [b, hw, 3] (fp32, any may be NULL = zero).
// out: d_feats [b, hw, S, C] (feats' dtype), d_sigma_raw [b, hw, S] fp32, d_rgb_raw [b, hw, S, 3] fp32 (may be NULL).
extern "C" int cd360_volrender_bwd(const void* feats, const void* sigma_raw, const void* rgb_raw, const void* dists, int d_ray_stride,
                                   const void* d_rendered, const void* d_fg, const void* d_alphas, const void* d_weights, const void* d_rgb,
                                   void* d_feats, void* d_sigma_raw, void* d_rgb_raw, int b, int hw, int S, int C, int dtype, int flags,
                                   void* stream) {
  if (!feats || !sigma_raw || !dists || !d_rendered || !d_feats || !d_sigma_raw || b <= 0 || hw <= 0 || S <= 0 || C <= 0) return CD360_ERR_ARG;
  if (S > MAX_S || (dtype != 0 && dtype != 1) || C % (dtype ? 8 : 4)) return CD360_ERR_SHAPE;
  if (d_ray_stride != 0 && d_ray_stride != S) return CD360_ERR_SHAPE;
  const long nrays = (long)b * hw;
  const unsigned blocks = (unsigned)((nrays + 3) / 4 > 256 * 16 ? 256 * 16 : (nrays + 3) / 4);
  if (dtype)
    hipLaunchKernelGGL(volrender_bwd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feats, (const float*)sigma_raw,
                       (const float*)rgb_raw, (const float*)dists, d_ray_stride, d_rendered, (const float*)d_fg, (const float*)d_alphas,
                       (const float*)d_weights, (const float*)d_rgb, d_feats, (float*)d_sigma_raw, (float*)d_rgb_raw, nrays, hw, S, C, flags);
  else
    hipLaunchKernelGGL(volrender_bwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feats, (const float*)sigma_raw,
                       (const float*)rgb_raw, (const float*)dists, d_ray_stride, d_rendered, (const float*)d_fg, (const float*)d_alphas,
                       (const float*)d_weights, (const float*)d_rgb, d_feats, (float*)d_sigma_raw, (float*)d_rgb_raw, nrays, hw, S, C, flags);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// ---- A9: decoder (zero-init Linear(C -> 4, no bias), sgm/modules/nerfsd_pytorch3d.py:49-51,160) on bf16 tokens with
// fp32 weights, fp32 accumulation and fp32 output (sigma_raw feeds exp(): it must not be rounded to bf16).
// One wave per row; HBM-bound (reads h once).
namespace {
// A lane owns the SAME 8 channels of every 512-channel chunk for all the rows its wave visits, so the NW x 8 weights of a chunk live in
// registers for the whole launch (the first version re-read them from the cache for every row: 32 dword loads beside each 16-byte load of
// h, 1.6 TB/s).  CH = chunks of 512 channels (C <= 512 CH); rows are dealt to the waves round-robin.
template <int NW, int CH>
__global__ __launch_bounds__(256) void rowdot_kernel(const uint16_t* __restrict__ h, const float* __restrict__ w, float* __restrict__ out,
                                                     long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  float wr[CH][NW][8];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    const int c = ch * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) wr[ch][j][e] = c < C ? w[j * C + c + e] : 0.f;
  }
  auto load_row = [&](long row, u32x4 (&v)[CH]) {
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const int c = ch * 512 + lane * 8;
      v[ch] = (c < C && row < rows) ? *reinterpret_cast<const u32x4*>(h + row * C + c) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  u32x4 v[CH], vn[CH];
  load_row(wave0, v);
  for (long row = wave0; row < rows; row += nwaves) {
    load_row(row + nwaves, vn);  // the next row travels while this one is reduced
    float a[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) a[j] = 0.f;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf16lo_to_f32(v[ch][e]), hi = bf16hi_to_f32(v[ch][e]);
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          a[j] = fmaf(lo, wr[ch][j][2 * e], a[j]);
          a[j] = fmaf(hi, wr[ch][j][2 * e + 1], a[j]);
        }
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int j = 0; j < NW; ++j) a[j] += __shfl_xor(a[j], off);
    }
    if (lane == 0) {
      if constexpr (NW == 4) {
        f32x4 o = {a[0], a[1], a[2], a[3]};
        *reinterpret_cast<f32x4*>(out + row * 4) = o;
      } else {
#pragma unroll
        for (int j = 0; j < NW; ++j) out[row * NW + j] = a[j];
      }
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) v[ch] = vn[ch];
  }
}
}  // namespace

template <int NW>
static int rowdot_pick(const void* h, const void* w, void* out, int64_t rows, int C, void* stream) {
  const long waves = rows < 256L * 3 * 4 ? rows : 256L * 3 * 4;  // three workgroups of four waves per CU (the 134 registers of three chunks allow that): ~16 rows per wave at the render shapes
  const unsigned nblk = (unsigned)((waves + 3) / 4);
  const int ch = (C + 511) / 512;
#define CD360_ROWDOT(CHN)                                                                                                             \
  hipLaunchKernelGGL((rowdot_kernel<NW, CHN>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)h, (const float*)w, \
                     (float*)out, (long)rows, C)
  switch (ch) {
    case 1: CD360_ROWDOT(1); break;
    case 2: CD360_ROWDOT(2); break;
    case 3: CD360_ROWDOT(3); break;
    case 4: CD360_ROWDOT(4); break;
    default: return CD360_ERR_SHAPE;
  }
#undef CD360_ROWDOT
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

static int rowdot_launch(int nw, const void* h, const void* w, void* out, int64_t rows, int C, void* stream) {
  if (!h || !w || !out || rows <= 0 || C <= 0) return CD360_ERR_ARG;
  if (C % 8 || C > 2048) return CD360_ERR_SHAPE;
  return nw == 4 ? rowdot_pick<4>(h, w, out, rows, C, stream) : rowdot_pick<1>(h, w, out, rows, C, stream);
}

// h [rows, C] bf16, w [4, C] fp32 -> out [rows, 4] fp32
extern "C" int cd360_rowdot4_bf16(const void* h, const void* w, void* out, int64_t rows, int C, void* stream) {
  return rowdot_launch(4, h, w, out, rows, C, stream);
}

// h [rows, C] bf16, w [C] fp32 -> out [rows] fp32: the view-logit column lv = xref . vf of the reference tables (plane_coefs.0's rows of the
// feature block folded with the aggregation head, cd360/nerf.py reference_tables; nerfsd_pytorch3d.py:130-146).  One read of h.
extern "C" int cd360_rowdot1_bf16(const void* h, const void* w, void* out, int64_t rows, int C, void* stream) {
  return rowdot_launch(1, h, w, out, rows, C, stream);
}

// Backward of cd360_rowdot4_bf16 (the FeatureNeRF decoder is trained: trainkeys = pose, diffusion.py:139-144):
//   dh[row, c] = sum_j d[row, j] w[j, c]   (bf16, the layout of h)          dw[j, c] = sum_row d[row, j] h[row, c]   (fp32)
// dh is one pass (reads 16 B of d per row, writes the row); dw is a column reduction over all rows: each workgroup reduces a slab of
// rows for 512 channels into a partial [slab, 4, C], summed by the caller in slab order (deterministic, no atomics).
namespace {
__global__ __launch_bounds__(256) void rowdot4_bwd_dh_kernel(const float* __restrict__ d, const float* __restrict__ w, uint16_t* __restrict__ dh,
                                                             long rows, int C) {
  const int cpv = C >> 3;
  const long total = rows * cpv;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    const long row = gid / cpv;
    const int c = (int)(gid - row * cpv) * 8;
    const f32x4 dv = *reinterpret_cast<const f32x4*>(d + row * 4);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v0 = fmaf(dv[j], w[(long)j * C + c + 2 * e], v0);
        v1 = fmaf(dv[j], w[(long)j * C + c + 2 * e + 1], v1);
      }
      o[e] = pack_bf16x2(v0, v1);
    }
    *reinterpret_cast<u32x4*>(dh + row * (long)C + c) = o;
  }
}

constexpr int DW_SLAB_ROWS = 256;
// One workgroup: 256 rows x 512 channels.  A wave owns every fourth row and reads it as one 1-KB access (16 bytes = 8 channels per
// lane), 32 fp32 accumulators per lane; the four waves meet once in the LDS.  (The first version gave a thread ONE channel and 512 rows:
// 2-byte loads, 0.4-0.6 TB/s.)
__global__ __launch_bounds__(256) void rowdot4_bwd_dw_kernel(const float* __restrict__ d, const uint16_t* __restrict__ h, float* __restrict__ part,
                                                             long rows, int C) {
  __shared__ float red[3][64][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 8;
  const bool live = c < C;
  const long r0 = (long)blockIdx.y * DW_SLAB_ROWS, r1 = r0 + DW_SLAB_ROWS < rows ? r0 + DW_SLAB_ROWS : rows;
  float a[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[j][e] = 0.f;
#pragma unroll 4
  for (long r = r0 + wv; r < r1; r += 4) {
    u32x4 hv = {0u, 0u, 0u, 0u};
    if (live) hv = *reinterpret_cast<const u32x4*>(h + r * C + c);
    const f32x4 dv = *reinterpret_cast<const f32x4*>(d + r * 4);  // wave-uniform address
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = bf16lo_to_f32(hv[e]), hi = bf16hi_to_f32(hv[e]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j][2 * e] = fmaf(dv[j], lo, a[j][2 * e]);
        a[j][2 * e + 1] = fmaf(dv[j], hi, a[j][2 * e + 1]);
      }
    }
  }
  if (wv) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wv - 1][lane][j * 8 + e] = a[j][e];
  }
  __syncthreads();
  if (wv == 0 && live) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a[j][e] = ((a[j][e] + red[0][lane][j * 8 + e]) + red[1][lane][j * 8 + e]) + red[2][lane][j * 8 + e];
      float* dst = part + ((long)blockIdx.y * 4 + j) * C + c;
      *reinterpret_cast<f32x4*>(dst) = f32x4{a[j][0], a[j][1], a[j][2], a[j][3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{a[j][4], a[j][5], a[j][6], a[j][7]};
    }
  }
}
}  // namespace

extern "C" int cd360_rowdot4_bwd_slabs(int64_t rows) { return (int)((rows + DW_SLAB_ROWS - 1) / DW_SLAB_ROWS); }

// d [rows, 4] fp32 (gradient of the output), h [rows, C] bf16, w [4, C] fp32 -> dh [rows, C] bf16 (NULL to skip),
// dw_part [cd360_rowdot4_bwd_slabs(rows), 4, C] fp32 partial sums of dw (NULL to skip; the caller sums the slabs)
extern "C" int cd360_rowdot4_bwd_bf16(const void* d, const void* h, const void* w, void* dh, void* dw_part, int64_t rows, int C, void* stream) {
  if (!d || !h || !w || rows <= 0 || C <= 0) return CD360_ERR_ARG;
  if (C % 8 || ((uintptr_t)d | (uintptr_t)h | (uintptr_t)dh) % 16) return CD360_ERR_SHAPE;
  if (dh) {
    const long total = rows * (C / 8), blocks = (total + 255) / 256;
    hipLaunchKernelGGL(rowdot4_bwd_dh_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, (const float*)d,
                       (const float*)w, (uint16_t*)dh, (long)rows, C);
    CD360_LAUNCH_CHECK();
  }
  if (dw_part) {
    const long slabs = (rows + DW_SLAB_ROWS - 1) / DW_SLAB_ROWS;
    if (slabs > 65535) return CD360_ERR_SHAPE;
    if ((uintptr_t)dw_part % 16) return CD360_ERR_ARG;
    hipLaunchKernelGGL(rowdot4_bwd_dw_kernel, dim3((unsigned)((C + 511) / 512), (unsigned)slabs), dim3(256), 0, (hipStream_t)stream,
                       (const float*)d, (const uint16_t*)h, (float*)dw_part, (long)rows, C);
    CD360_LAUNCH_CHECK();
  }
  return CD360_OK;
}
