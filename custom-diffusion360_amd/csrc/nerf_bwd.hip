// Backward of cd360_nerf_mlp_aggregate (nerf_fused.hip) for the fine-tuning loop: the reference differentiates
// FeatureNeRFEncoding.forward (sgm/modules/nerfsd_pytorch3d.py:53-158 -- grid_sample, plane_coefs, the view softmax) through
// torch autograd; the trainable parameters on this path are plane_coefs.{0,2}, nviews and decoder ("pose" in the name,
// diffusion.py:139-144).
//
// Forward (per batch b, sample = (ray k, depth s), view i, channel c):
//   z_i = bilinear(Y_i) + zP_i(k) + Wk . F(q_i)        a_i = softmax_i(bilinear(lv_i) + cview_i)        g = sum_i a_i silu(z_i)
// Backward, given dg:
//   dz_i      = a_i dg silu'(z_i)                                     -> written out [b, n, hw*S, C] bf16
//   dlogit_i  = a_i <dg, silu(z_i) - g>   (over ALL channels)         -> fp32 atomics into dlogit [b, n, hw*S] (one add per 64-channel chunk);
//               the deterministic form (dlogit_parts) takes g = sum_j a_j silu(z_j) from its own fp32 terms instead of the saved bf16 g
//   dY        += w_corner dz_i   at the four gathered texels         -> fp32 atomics into dY [tables, hw, C]
// and the generated inputs F(q_i) (the 112 sin / cos / xyz columns, bf16, the forward's k order) are written once [b, n, hw*S, 112].
// The remaining reductions are plain library work on those two tensors (host side, grad.NerfAggregateFn):
//   dWk = dz^T F (one GEMM),  dzP = sum_s dz.
// nerf_logit_bwd_kernel then scatters dlogit through the same bilinear weights into dlv and sums it into dcview.
// Same wave decomposition as the forward (32 samples x 64 channels per wave, z recomputed on MFMA with the identical instruction
// sequence), so the recomputed z_i and a_i are the forward's.
#include "cd360_geom.h"

namespace {

constexpr int CN = 64;
constexpr int KP = 112;
constexpr int W_PITCH = 240;
constexpr int TILES_PER_WAVE = 4;
constexpr int PTS_PER_WG = 4 * 32 * TILES_PER_WAVE;

struct NerfBwdParams {
  const float* cams;
  const float* xs;
  const float* ys;
  const float* t;
  const uint16_t* Y;
  const uint16_t* zP;
  const float* lv;
  const float* cview;
  const uint16_t* Wk;
  const int* img_map;
  const uint16_t* g;    // [b, hw*S, C] forward output
  const float* lse;     // [b, hw*S, 2] forward (max * ln2, sum)
  const uint16_t* dg;   // [b, hw*S, C]
  uint16_t* dz;         // [b, n, hw*S, C] out
  uint16_t* F;          // [b, n, hw*S, KP] out
  float* dY;            // [tables, hw, C] fp32, zero-initialised by the caller, atomically accumulated
  float* dlogit;        // [b, n, hw*S] fp32, zero-initialised by the caller, atomically accumulated (unless dlogit_parts)
  float* dlogit_parts;  // NULL, or [C / 64, b, n, hw*S] fp32: every channel chunk STORES its term (one writer per element), summed afterwards
                        // in chunk order by nerf_dlogit_sum_kernel: the deterministic form (cd360_nerf_mlp_aggregate_bwd_det)
  int b, n, r, S, C, t_ray_stride, ncc, ngroups;
};

__device__ __forceinline__ int chan_pos(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

__global__ __launch_bounds__(256) void nerf_bwd_kernel(NerfBwdParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char Ws[CN * W_PITCH];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int hw = p.r * p.r;
  const long npts = (long)hw * p.S;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int cc = wg / (p.b * p.ngroups);
  const int rem = wg - cc * (p.b * p.ngroups);
  const int bi = rem / p.ngroups, grp = rem - bi * p.ngroups;
  const int ch0 = cc * CN;

  for (int idx = tid; idx < CN * (KP / 8); idx += 256) {
    const int row = idx / (KP / 8), c8 = idx - row * (KP / 8);
    *reinterpret_cast<u32x4*>(Ws + row * W_PITCH + c8 * 16) = *reinterpret_cast<const u32x4*>(p.Wk + (long)(ch0 + row) * KP + c8 * 8);
  }
  __syncthreads();

  const Cam c0 = load_cam(p.cams + (long)bi * (p.n + 1) * 16);
  const float hs = hh ? 2.f : 1.f;
  const int arow0 = chan_pos(l31);

  for (int tw = 0; tw < TILES_PER_WAVE; ++tw) {
    const long pt0 = (long)grp * PTS_PER_WG + (wave * TILES_PER_WAVE + tw) * 32;
    if (pt0 >= npts) break;  // wave-uniform
    const long pt = pt0 + l31;
    const bool valid = pt < npts;
    const long ptc = valid ? pt : npts - 1;
    const int k = (int)(ptc / p.S), s = (int)(ptc - (long)k * p.S);
    float o[3], d[3], P[3];
    patch_ray(c0, p.xs[k % p.r], p.ys[k / p.r], o, d);
    const float ts = p.t[(long)k * p.t_ray_stride + s];
#pragma unroll
    for (int j = 0; j < 3; ++j) P[j] = o[j] + ts * d[j];

    // this lane's 2 x 16 channels of dg and g (fp32), and the softmax statistics of its sample
    float dgv[2][16], gv[2][16];
    {
      const uint16_t* dsrc = p.dg + ((long)bi * npts + ptc) * p.C + ch0 + 16 * hh;
      const uint16_t* gsrc = p.g + ((long)bi * npts + ptc) * p.C + ch0 + 16 * hh;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(dsrc + mb * 32 + half * 8);
          const u32x4 c = *reinterpret_cast<const u32x4*>(gsrc + mb * 32 + half * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dgv[mb][half * 8 + 2 * e] = bf16lo_to_f32(a[e]);
            dgv[mb][half * 8 + 2 * e + 1] = bf16hi_to_f32(a[e]);
            gv[mb][half * 8 + 2 * e] = bf16lo_to_f32(c[e]);
            gv[mb][half * 8 + 2 * e + 1] = bf16hi_to_f32(c[e]);
          }
        }
    }
    const float m2 = p.lse[((long)bi * npts + ptc) * 2] * 1.4426950408889634f;  // running max in log2 units
    const float inv_l = 1.f / p.lse[((long)bi * npts + ptc) * 2 + 1];

    for (int iv = 0; iv < p.n; ++iv) {
      const Cam ci = load_cam(p.cams + ((long)bi * (p.n + 1) + 1 + iv) * 16);
      float q[3];
      world_to_view(ci, P, q);
      const Corner cr = bilinear_corner(grid_coord(ci.f[0], ci.c[0], q[0], q[2]), grid_coord(ci.f[1], ci.c[1], q[1], q[2]), p.r);
      const int x0 = min(max(cr.x0, 0), p.r - 1), x1 = min(max(cr.x0 + 1, 0), p.r - 1);
      const int y0 = min(max(cr.y0, 0), p.r - 1), y1 = min(max(cr.y0 + 1, 0), p.r - 1);
      const long img = (long)bi * p.n + iv;
      const long yimg = p.img_map ? (long)p.img_map[img] : img;
      const long pix[4] = {yimg * hw + (long)y0 * p.r + x0, yimg * hw + (long)y0 * p.r + x1, yimg * hw + (long)y1 * p.r + x0,
                           yimg * hw + (long)y1 * p.r + x1};
      float w[4] = {(1.f - cr.tx) * (1.f - cr.ty), cr.tx * (1.f - cr.ty), (1.f - cr.tx) * cr.ty, cr.tx * cr.ty};
#pragma unroll
      for (int c = 0; c < 4; ++c) if (!((cr.mask >> c) & 1)) w[c] = 0.f;

      u32x4 yv[4][2][2];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const uint16_t* src = p.Y + pix[c] * p.C + ch0 + mb * 32 + 16 * hh;
          yv[c][mb][0] = *reinterpret_cast<const u32x4*>(src);
          yv[c][mb][1] = *reinterpret_cast<const u32x4*>(src + 8);
        }
      u32x4 zp[2][2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const uint16_t* src = p.zP + (img * hw + k) * p.C + ch0 + mb * 32 + 16 * hh;
        zp[mb][0] = *reinterpret_cast<const u32x4*>(src);
        zp[mb][1] = *reinterpret_cast<const u32x4*>(src + 8);
      }
      float logit = p.cview[bi * p.n + iv];
#pragma unroll
      for (int c = 0; c < 4; ++c) logit = fmaf(w[c], p.lv[pix[c]], logit);

      // ---- recompute z (identical to the forward), and write the generated inputs once (chunk 0) ----
      const float qh[3] = {q[0] * hs, q[1] * hs, q[2] * hs};
      f32x16 z[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) { z[0][i] = 0.f; z[1][i] = 0.f; }
      uint16_t* frow = p.F + (img * npts + ptc) * KP + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
        uint32_t fw[4];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const int wi = ks * 4 + pr;
          if (wi < 24) {
            const int comp = wi % 3, kfp = wi / 3;
            const float rev = __builtin_amdgcn_fractf(qh[comp] * __builtin_bit_cast(float, (uint32_t)((127 + 2 * kfp - 9) << 23)));
            fw[pr] = pack_bf16x2(__builtin_amdgcn_sinf(rev), __builtin_amdgcn_cosf(rev));
          } else if (wi == 24) {
            fw[pr] = pack_bf16x2(hh ? q[2] : q[0], hh ? 0.f : q[1]);
          } else {
            fw[pr] = 0u;
          }
        }
        u32x4 fv = {fw[0], fw[1], fw[2], fw[3]};
        if (cc == 0 && valid) *reinterpret_cast<u32x4*>(frow + ks * 16) = fv;
        const bf16x8 fb = __builtin_bit_cast(bf16x8, fv);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ws + (mb * 32 + arow0) * W_PITCH + ks * 32 + hh * 16);
          z[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fb, z[mb], 0, 0, 0);
        }
      }

      const float a = __builtin_amdgcn_exp2f(logit * 1.4426950408889634f - m2) * inv_l;  // this view's softmax weight

      // ---- z += zP + bilinear(Y); dz = a dg silu'(z); dot = sum_c dg (silu(z) - g) ----
      float dot = 0.f;
      uint16_t* dzrow = p.dz + (img * npts + ptc) * p.C + ch0 + 16 * hh;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        float dzv[16];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r0 = half * 8 + 2 * e;
            float z0 = z[mb][r0] + bf16lo_to_f32(zp[mb][half][e]);
            float z1 = z[mb][r0 + 1] + bf16hi_to_f32(zp[mb][half][e]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              z0 = fmaf(w[c], bf16lo_to_f32(yv[c][mb][half][e]), z0);
              z1 = fmaf(w[c], bf16hi_to_f32(yv[c][mb][half][e]), z1);
            }
            const float sg0 = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z0));
            const float sg1 = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z1));
            // (deterministic form: <dg, silu(z)> alone -- the mean over the views is taken from these very terms afterwards, see
            // nerf_dlogit_sum_kernel; the atomic form subtracts the SAVED, bf16-rounded g)
            dot = fmaf(dgv[mb][r0], p.dlogit_parts ? z0 * sg0 : z0 * sg0 - gv[mb][r0], dot);
            dot = fmaf(dgv[mb][r0 + 1], p.dlogit_parts ? z1 * sg1 : z1 * sg1 - gv[mb][r0 + 1], dot);
            dzv[r0] = a * dgv[mb][r0] * (sg0 * (1.f + z0 * (1.f - sg0)));
            dzv[r0 + 1] = a * dgv[mb][r0 + 1] * (sg1 * (1.f + z1 * (1.f - sg1)));
          }
        }
        if (valid) {
          u32x4 o0, o1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o0[e] = pack_bf16x2(dzv[2 * e], dzv[2 * e + 1]);
            o1[e] = pack_bf16x2(dzv[8 + 2 * e], dzv[8 + 2 * e + 1]);
          }
          *reinterpret_cast<u32x4*>(dzrow + mb * 32) = o0;
          *reinterpret_cast<u32x4*>(dzrow + mb * 32 + 8) = o1;
          // ---- dY += w_corner dz at the four texels (corners with zero weight -- out of range -- are skipped).  Optional: the
          // training path forms the weight gradient it is after as ONE GEMM, dWf = gather(xref)^T dz, and passes dY = NULL ----
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (p.dY && w[c] != 0.f) {
              float* dst = p.dY + pix[c] * p.C + ch0 + mb * 32 + 16 * hh;
#pragma unroll
              for (int e = 0; e < 16; ++e) atomicAdd(dst + e, w[c] * dzv[e]);
            }
          }
        }
      }
      dot += __shfl_xor(dot, 32);
      if (valid && hh == 0) {
        if (p.dlogit_parts) {
          p.dlogit_parts[((long)cc * p.b * p.n + img) * npts + pt] = a * dot;
          if (cc == 0) p.dlogit[img * npts + pt] = a;  // the softmax weight itself, for nerf_dlogit_sum_kernel (which overwrites it)
        } else {
          atomicAdd(p.dlogit + img * npts + pt, a * dot);
        }
      }
    }
  }
}

// dlogit_i = a_i (e_i - sum_j a_j e_j), e_i = <dg, silu(z_i)> over ALL channels.  parts[cc][b, i, pt] = a_i e_i restricted to channel chunk cc,
// dlogit holds a_i on entry.  One thread per (batch, sample): E_i = the chunks' terms added in chunk order, their sum over the views is
// the mean term -- so the view-logit gradients of a sample sum to zero to fp32 rounding, as the softmax demands.  (Rounds 1-5 used
// a_i <dg, silu(z_i) - g> with the SAVED g: g is bf16, and its rounding, the same for every view of a sample, left each sample's gradients
// with a common-mode error that no cancellation removes: 3e-2 ... 5e-2 of the nviews.weight gradient against the reference's autograd.)
__global__ __launch_bounds__(256) void nerf_dlogit_sum_kernel(float* __restrict__ parts, float* __restrict__ dlogit, int b, int n, long npts, int ncc) {
  const long total = (long)b * n * npts;
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < (long)b * npts; j += (long)gridDim.x * blockDim.x) {
    const long bi = j / npts, pt = j - bi * npts;
    float ebar = 0.f;
    for (int iv = 0; iv < n; ++iv) {
      const long i = (bi * n + iv) * npts + pt;
      float acc = parts[i];
      for (int cc = 1; cc < ncc; ++cc) acc += parts[(long)cc * total + i];
      parts[i] = acc;
      ebar += acc;
    }
    for (int iv = 0; iv < n; ++iv) {
      const long i = (bi * n + iv) * npts + pt;
      dlogit[i] = parts[i] - dlogit[i] * ebar;
    }
  }
}

// dlogit [b, n, hw*S] -> dlv (scatter through the forward's bilinear weights) and dcview (sum per (batch, view)).
// grid (blocks over samples, b*n): one (batch, view) per block row.
__global__ __launch_bounds__(256) void nerf_logit_bwd_kernel(const float* __restrict__ cams, const float* __restrict__ xs, const float* __restrict__ ys,
                                                             const float* __restrict__ t, int t_ray_stride, const int* __restrict__ img_map,
                                                             const float* __restrict__ dlogit, float* __restrict__ dlv, float* __restrict__ dcview,
                                                             int n, int r, int S) {
  __shared__ float red[4];
  const int img = blockIdx.y, bi = img / n, iv = img - bi * n, hw = r * r;
  const long npts = (long)hw * S;
  const Cam c0 = load_cam(cams + (long)bi * (n + 1) * 16);
  const Cam ci = load_cam(cams + ((long)bi * (n + 1) + 1 + iv) * 16);
  const long yimg = img_map ? (long)img_map[img] : (long)img;
  float acc = 0.f;
  for (long pt = (long)blockIdx.x * blockDim.x + threadIdx.x; pt < npts; pt += (long)gridDim.x * blockDim.x) {
    const int k = (int)(pt / S), s = (int)(pt - (long)k * S);
    float o[3], d[3], P[3], q[3];
    patch_ray(c0, xs[k % r], ys[k / r], o, d);
    const float ts = t[(long)k * t_ray_stride + s];
#pragma unroll
    for (int j = 0; j < 3; ++j) P[j] = o[j] + ts * d[j];
    world_to_view(ci, P, q);
    const Corner cr = bilinear_corner(grid_coord(ci.f[0], ci.c[0], q[0], q[2]), grid_coord(ci.f[1], ci.c[1], q[1], q[2]), r);
    const int x0 = min(max(cr.x0, 0), r - 1), x1 = min(max(cr.x0 + 1, 0), r - 1);
    const int y0 = min(max(cr.y0, 0), r - 1), y1 = min(max(cr.y0 + 1, 0), r - 1);
    const long pix[4] = {yimg * hw + (long)y0 * r + x0, yimg * hw + (long)y0 * r + x1, yimg * hw + (long)y1 * r + x0, yimg * hw + (long)y1 * r + x1};
    const float w[4] = {(1.f - cr.tx) * (1.f - cr.ty), cr.tx * (1.f - cr.ty), (1.f - cr.tx) * cr.ty, cr.tx * cr.ty};
    const float dl = dlogit[(long)img * npts + pt];
    acc += dl;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if ((cr.mask >> c) & 1) atomicAdd(dlv + pix[c], w[c] * dl);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dcview + img, (red[0] + red[1]) + (red[2] + red[3]));
}

}  // namespace

// Inputs as cd360_nerf_mlp_aggregate, plus its outputs g and lse and the incoming gradient dg [b, hw*S, C] bf16.
// Outputs: dz [b, n, hw*S, C] bf16 and F [b, n, hw*S, 112] bf16 (fully written); dY [tables, hw, C], dlogit [b, n, hw*S],
// dlv [tables, hw], dcview [b, n] fp32 -- these four are ACCUMULATED atomically: the caller zero-fills them.  dY may be NULL and
// dlv / dcview may both be NULL (no scatter: the caller reduces dz and dlogit itself, e.g. against gathered reference features).
static int nerf_bwd_launch(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, const void* Y,
                           const void* zP, const void* lv, const void* cview, const void* Wk, const void* img_map,
                           const void* g, const void* lse, const void* dg, void* dz, void* F, void* dY, void* dlogit, void* dlogit_parts, void* dlv,
                           void* dcview, int b, int n, int r, int S, int C, void* stream) {
  if (!cams || !xs || !ys || !t || !Y || !zP || !lv || !cview || !Wk || !g || !lse || !dg || !dz || !F || !dlogit) return CD360_ERR_ARG;
  if ((dlv == nullptr) != (dcview == nullptr)) return CD360_ERR_ARG;
  if (b <= 0 || n <= 0 || r <= 0 || S <= 0 || C <= 0 || C % CN) return CD360_ERR_SHAPE;
  if (t_ray_stride != 0 && t_ray_stride != S) return CD360_ERR_SHAPE;
  if (((uintptr_t)Y | (uintptr_t)zP | (uintptr_t)Wk | (uintptr_t)g | (uintptr_t)dg | (uintptr_t)dz | (uintptr_t)F) % 16) return CD360_ERR_ARG;
  NerfBwdParams p;
  p.cams = (const float*)cams; p.xs = (const float*)xs; p.ys = (const float*)ys; p.t = (const float*)t;
  p.Y = (const uint16_t*)Y; p.zP = (const uint16_t*)zP; p.lv = (const float*)lv; p.cview = (const float*)cview;
  p.Wk = (const uint16_t*)Wk; p.img_map = (const int*)img_map; p.g = (const uint16_t*)g; p.lse = (const float*)lse;
  p.dg = (const uint16_t*)dg; p.dz = (uint16_t*)dz; p.F = (uint16_t*)F; p.dY = (float*)dY; p.dlogit = (float*)dlogit;
  p.dlogit_parts = (float*)dlogit_parts;
  p.b = b; p.n = n; p.r = r; p.S = S; p.C = C; p.t_ray_stride = t_ray_stride;
  p.ncc = C / CN;
  const long npts = (long)r * r * S;
  p.ngroups = (int)((npts + PTS_PER_WG - 1) / PTS_PER_WG);
  const long nwg = (long)p.ncc * b * p.ngroups;
  if (nwg > 0x7fffffffL) return CD360_ERR_SHAPE;
  hipLaunchKernelGGL(nerf_bwd_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
  CD360_LAUNCH_CHECK();
  if (dlogit_parts) {
    const long blocks = ((long)b * npts + 255) / 256;
    hipLaunchKernelGGL(nerf_dlogit_sum_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       (float*)dlogit_parts, (float*)dlogit, b, n, npts, p.ncc);
    CD360_LAUNCH_CHECK();
  }
  if (!dlv) return CD360_OK;
  const long blocks = (npts + 255) / 256;
  hipLaunchKernelGGL(nerf_logit_bwd_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks), (unsigned)(b * n)), dim3(256), 0, (hipStream_t)stream,
                     p.cams, p.xs, p.ys, p.t, t_ray_stride, p.img_map, (const float*)dlogit, (float*)dlv, (float*)dcview, n, r, S);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

extern "C" int cd360_nerf_mlp_aggregate_bwd(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, const void* Y,
                                            const void* zP, const void* lv, const void* cview, const void* Wk, const void* img_map,
                                            const void* g, const void* lse, const void* dg, void* dz, void* F, void* dY, void* dlogit, void* dlv,
                                            void* dcview, int b, int n, int r, int S, int C, void* stream) {
  return nerf_bwd_launch(cams, xs, ys, t, t_ray_stride, Y, zP, lv, cview, Wk, img_map, g, lse, dg, dz, F, dY, dlogit, nullptr, dlv, dcview, b, n, r, S, C,
                         stream);
}

// The deterministic form: dlogit is WRITTEN (no zero-fill needed), as the sum in chunk order of the per-channel-chunk terms the kernel stores
// into dlogit_parts [C / 64, b, n, hw*S] fp32 (scratch, fully written) -- no atomic touches it, so a fine-tuning step is bit-reproducible
// run to run (with fp32 atomics the order of ~C / 64 additions per element varied, the view-logit parameters' gradients moved in their last
// bits, and once in a while that flipped the bf16 rounding of a trained weight: a 1e-4 change of the loss a few steps later).
extern "C" int cd360_nerf_mlp_aggregate_bwd_det(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, const void* Y,
                                                const void* zP, const void* lv, const void* cview, const void* Wk, const void* img_map,
                                                const void* g, const void* lse, const void* dg, void* dz, void* F, void* dY, void* dlogit,
                                                void* dlogit_parts, void* dlv, void* dcview, int b, int n, int r, int S, int C, void* stream) {
  if (!dlogit_parts) return CD360_ERR_ARG;
  return nerf_bwd_launch(cams, xs, ys, t, t_ray_stride, Y, zP, lv, cview, Wk, img_map, g, lse, dg, dz, F, dY, dlogit, dlogit_parts, dlv, dcview, b, n, r,
                         S, C, stream);
}
