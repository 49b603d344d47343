// A5-A8 fused: FeatureNeRFEncoding.forward (sgm/modules/nerfsd_pytorch3d.py:53-158) without ever
// materialising the [b, n, hw, S, C+198] tensor the reference builds with torch.cat.
//
// Algebra (exact in real arithmetic):
//   plane_coefs.0 is linear, so its action on the bilinearly gathered features equals the bilinear
//   gather of  Y = xref @ W1[:, :C]^T  (one GEMM over n*hw pixels instead of n*hw*S samples);
//   its action on [enc8(plucker), dir] depends on (view, ray) only -> table zP (+ bias b1);
//   only [enc16(q_i), q_i] (99 inputs) is truly per (view, ray, sample): that part runs here on MFMA.
//   The view softmax weights sum to 1, so plane_coefs.2 commutes with the weighted sum over views
//   and is applied once per sample afterwards (host GEMM on the output of this kernel).
//   The view logit keeps only its view-dependent terms: the bilinear gather of lv = xref @ w_v[:C]
//   plus a per-view constant; the [enc16(q_0), q_0] terms are common to all views and cancel.
//
// Kernel: one wave = 32 samples x 64 channels.  z^T[channel, sample] = Wk[channel, 112] . F^T[112, sample]
// on v_mfma_f32_32x32x16_bf16 with the 112 = 96 sin/cos + 3 xyz (+ pad) inputs generated in registers in
// MFMA B-operand layout (v_sin/v_cos on exact power-of-two phase scalings; no tables, no LDS);
// the gathered Y / zP rows are added with fp32 bilinear weights; SiLU; flash-style online softmax over views.
// Each (lane, lane^32) pair owns one sample and reads one full 128-B line per texel corner.
#include "cd360_geom.h"
#include "cd360_tuning.h"
#include <stdlib.h>

namespace {

constexpr int CN = 64;          // channels per workgroup
constexpr int KP = 112;         // padded K of the per-sample GEMM (7 k-steps of 16)
constexpr int W_PITCH = 240;    // bytes per Wk row in LDS (224 + 16 pad: conflict-free ds_read_b128)
constexpr int TILES_PER_WAVE = 4;
constexpr bool NERF_LINE_DEFAULT = true;  // which render kernel cd360_tuning.nerf_kernel = -1 selects: the full-line kernel measures -3 % (1280
                                          // channels) / -8 ... -9 % (640 channels) against the register-gather kernel, bit-identical
constexpr int PTS_PER_WG = 4 * 32 * TILES_PER_WAVE;

struct NerfParams {
  const float* cams;      // [b, n+1, 16]
  const float* xs;        // [r]
  const float* ys;        // [r]
  const float* t;         // [hw, S] or [S]
  const uint16_t* Y;      // [b*n, hw, C] bf16
  const uint16_t* zP;     // [b*n, hw, C] bf16 (bias folded in)
  const float* lv;        // [b*n, hw]
  const float* cview;     // [b, n]
  const uint16_t* Wk;     // [C, KP] bf16, k-permuted (see nerf.py: xyz_k_columns)
  const int* img_map;     // optional [b*n]: index of the (Y, lv) table image used by (batch, view); NULL = identity
  uint16_t* g;            // [b, hw*S, C] bf16 out: sum_i softmax_i * silu(z_i)
  float* logits;          // optional [b, n, hw*S]
  float* lse;             // optional [b, hw*S, 2] = (max, sum)
  int b, n, r, S, C, t_ray_stride, ncc, ngroups;
  // two-pass form (nerf_geom_kernel -> nerf_fused_rec_kernel): per (batch, view, sample) a 32-byte record, per (batch, sample) the softmax
  // statistics over the views, both in a caller-provided workspace
  u32x4* rec;             // [b, n, hw*S, 2]: {q.x, q.y, q.z, logit * log2(e)} | {pix | dx << 26 | dy << 27 | mask << 28, tx, ty, 0}
  f32x2* ml;              // [b, hw*S]: (max over the views of logit * log2(e), sum of exp2(. - max))
};

// A-operand row i of a 32-channel block holds the weights of channel offset pos(i), so that MFMA output
// register r of lane half h is channel 16*h + r (16 consecutive channels per lane).
__device__ __forceinline__ int chan_pos(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

__global__ __launch_bounds__(256, 2) void nerf_fused_kernel(NerfParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char Ws[CN * W_PITCH];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int hw = p.r * p.r;
  const long npts = (long)hw * p.S;

  // logical work item: [cc][b][group]  (chunk-major so that one XCD's L2 holds few channel slices)
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int cc = wg / (p.b * p.ngroups);
  const int rem = wg - cc * (p.b * p.ngroups);
  const int bi = rem / p.ngroups, grp = rem - bi * p.ngroups;
  const int ch0 = cc * CN;

  // ---- stage this chunk's Wk rows into LDS ----
  for (int idx = tid; idx < CN * (KP / 8); idx += 256) {
    const int row = idx / (KP / 8), c8 = idx - row * (KP / 8);
    *reinterpret_cast<u32x4*>(Ws + row * W_PITCH + c8 * 16) =
        *reinterpret_cast<const u32x4*>(p.Wk + (long)(ch0 + row) * KP + c8 * 8);
  }
  __syncthreads();

  const Cam c0 = load_cam(p.cams + (long)bi * (p.n + 1) * 16);
  const float hs = hh ? 2.f : 1.f;
  const int arow0 = chan_pos(l31);  // my A-operand row within a 32-channel block

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

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 g[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { g[0][i] = 0.f; g[1][i] = 0.f; }

    for (int iv = 0; iv < p.n; ++iv) {
      const Cam ci = load_cam(p.cams + ((long)bi * (p.n + 1) + 1 + iv) * 16);
      float q[3];
      world_to_view(ci, P, q);
      const Corner cr = bilinear_corner(grid_coord(ci.f[0], ci.c[0], q[0], q[2]), grid_coord(ci.f[1], ci.c[1], q[1], q[2]), p.r);
      const int x0 = min(max(cr.x0, 0), p.r - 1), x1 = min(max(cr.x0 + 1, 0), p.r - 1);
      const int y0 = min(max(cr.y0, 0), p.r - 1), y1 = min(max(cr.y0 + 1, 0), p.r - 1);
      const long img = (long)bi * p.n + iv;
      const long yimg = p.img_map ? (long)p.img_map[img] : img;  // which table image (Y, lv) this (batch, view) reads
      const long pix[4] = {yimg * hw + (long)y0 * p.r + x0, yimg * hw + (long)y0 * p.r + x1, yimg * hw + (long)y1 * p.r + x0,
                           yimg * hw + (long)y1 * p.r + x1};
      float w[4] = {(1.f - cr.tx) * (1.f - cr.ty), cr.tx * (1.f - cr.ty), (1.f - cr.tx) * cr.ty, cr.tx * cr.ty};
#pragma unroll
      for (int c = 0; c < 4; ++c) if (!((cr.mask >> c) & 1)) w[c] = 0.f;

      // ---- issue the gathers early: 4 corners of Y (this lane's 2 x 16 channels) + zP row + lv ----
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

      // ---- per-sample inputs in B-operand layout: lane half h handles frequencies 2*kfp + h ----
      const float qh[3] = {q[0] * hs, q[1] * hs, q[2] * hs};
      f32x16 z[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) { z[0][i] = 0.f; z[1][i] = 0.f; }
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
        const bf16x8 fb = __builtin_bit_cast(bf16x8, fv);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ws + (mb * 32 + arow0) * W_PITCH + ks * 32 + hh * 16);
          z[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fb, z[mb], 0, 0, 0);
        }
      }

      // ---- online softmax over views ----
      const float lg = logit * 1.4426950408889634f;
      const float m_new = fmaxf(m_run, lg);
      const float sc = __builtin_amdgcn_exp2f(m_run - m_new);
      const float a = __builtin_amdgcn_exp2f(lg - m_new);
      l_run = fmaf(l_run, sc, a);
      m_run = m_new;
      if (p.logits) {
        if (valid && hh == 0 && cc == 0) p.logits[((long)bi * p.n + iv) * npts + pt] = logit;
      }

      // ---- z += zP + bilinear(Y);  g = g*sc + a*silu(z) ----
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
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
            const float s0 = z0 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z0));
            const float s1 = z1 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z1));
            g[mb][r0] = fmaf(a, s0, g[mb][r0] * sc);
            g[mb][r0 + 1] = fmaf(a, s1, g[mb][r0 + 1] * sc);
          }
        }
      }
    }

    if (valid) {
      const float inv = 1.f / l_run;
      uint16_t* dst = p.g + ((long)bi * npts + pt) * p.C + ch0 + 16 * hh;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        u32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = pack_bf16x2(g[mb][2 * e] * inv, g[mb][2 * e + 1] * inv);
          o1[e] = pack_bf16x2(g[mb][8 + 2 * e] * inv, g[mb][8 + 2 * e + 1] * inv);
        }
        *reinterpret_cast<u32x4*>(dst + mb * 32) = o0;
        *reinterpret_cast<u32x4*>(dst + mb * 32 + 8) = o1;
      }
      if (p.lse && hh == 0 && cc == 0) {
        p.lse[((long)bi * npts + pt) * 2] = m_run * 0.6931471805599453f;
        p.lse[((long)bi * npts + pt) * 2 + 1] = l_run;
      }
    }
  }
}

// ---- the same operator with FULL-LINE gathers staged through the wave's own LDS block ------------------------------------------------
// nerf_fused_kernel is bound by the vector-memory RETURN path, not by its arithmetic (TD_TD_BUSY = 100 % of the kernel's CU-cycles; builds
// without the sin / cos or without the SiLU transcendentals run no faster): with lane = sample every gather instruction returns 32
// different cache lines, 32 bytes of each, and every line is visited by four instructions.  Here the four corner rows of a view travel as
// FULL 128-byte lines: 8 lanes per row (lane i of piece j fetches the 16-byte chunk i % 8 of sample 8 j + i / 8, whose table pixel it
// gets from that sample's lane by ds_bpermute), 16 load instructions x 8 lines per wave and view instead of 32 x 32 line returns -- the
// "coalesced gather with LDS staging" of the north star.  The rows arrive in REGISTERS (ordinary buffer loads, so the wave's own vmcnt
// orders them -- unlike LDS-DMA pieces, which need a workgroup barrier before they can be read and cost this kernel its gain, see the
// probe kernel below), are written to a wave-private LDS block of four 4-KB corner slots (16-byte chunks XOR-swizzled by the sample:
// conflict-free ds_write_b128 / ds_read_b128) and read back in the compute layout (lane = sample, 2 x 16 channels).  Software pipeline,
// one view ahead: the loads of view iv+1 are issued before the encoding MFMAs of view iv and written to the slots after view iv's blend
// has consumed them, so neither the gather latency nor the LDS round trip is exposed.  The arithmetic and its order are those of
// nerf_fused_kernel: bit-identical outputs.  No workgroup barrier in the loop.
constexpr int SLOT_BYTES = 32 * 128;            // one corner: 32 samples x 128 B
constexpr int WAVE_LDS = 4 * SLOT_BYTES;        // four corners per wave

__global__ __launch_bounds__(256, 2) void nerf_fused_line_kernel(NerfParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_line[];
  unsigned char* const Ws = smem_line + 4 * WAVE_LDS;       // CN x W_PITCH
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const slots = smem_line + wave * WAVE_LDS;
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

  const float hs = hh ? 2.f : 1.f;
  const int arow0 = chan_pos(l31);
  typedef const __attribute__((address_space(4))) float cfloat;
  typedef const __attribute__((address_space(4))) int cint;
  cfloat* const cams_c = (cfloat*)(p.cams);
  cfloat* const cview_c = (cfloat*)(p.cview);
  cint* const imap_c = (cint*)(p.img_map);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Y, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.zP, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.lv, 0, 0xffffffff, 0x00020000);
  const uint32_t row_bytes = (uint32_t)p.C * 2u;
  const uint32_t lane_off = (uint32_t)(ch0 + 16 * hh) * 2u;  // this lane's 32 bytes of a 64-channel slice (block mb: + 64 B)
  // loader role: piece j moves samples 8 j .. 8 j + 7; this lane: sample 8 j + lane / 8, chunk lane % 8 of its 128-byte row
  const int dsub = lane >> 3, dchunk = lane & 7;
  const uint32_t ld_off = (uint32_t)(ch0 * 2 + dchunk * 16);
  // LDS: sample s, chunk q at s * 128 + ((q ^ ((s >> 1) & 7)) << 4); loader writes (8 j + dsub, dchunk), compute reads (l31, 4 mb + 2 hh + t)
  int wr_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int smp = 8 * j + dsub;
    wr_off[j] = smp * 128 + ((dchunk ^ ((smp >> 1) & 7)) << 4);
  }
  const int rswz = (l31 >> 1) & 7;

  for (int tw = 0; tw < TILES_PER_WAVE; ++tw) {
    const long pt0 = (long)grp * PTS_PER_WG + (wave * TILES_PER_WAVE + tw) * 32;
    if (pt0 >= npts) break;  // wave-uniform
    const long pt = pt0 + l31;
    const bool valid = pt < npts;
    const long ptc = valid ? pt : npts - 1;
    const int k = (int)(ptc / p.S), s = (int)(ptc - (long)k * p.S);
    float P[3];
    {
      const Cam c0 = load_cam(p.cams + (long)bi * (p.n + 1) * 16);
      float o[3], d[3];
      patch_ray(c0, p.xs[k % p.r], p.ys[k / p.r], o, d);
      const float ts = p.t[(long)k * p.t_ray_stride + s];
#pragma unroll
      for (int j = 0; j < 3; ++j) P[j] = o[j] + ts * d[j];
    }

    struct Geo {
      float q[3], w[4], lvv[4], cv;  // lvv: the four gathered logit texels, raw (combined one iteration after their loads were issued)
      int pix[4];                    // table pixel (image included) of the four corners
    };
    auto geometry = [&](int iv, Geo& G) {  // projection, corners, weights of view iv; issues its four logit-texel loads
      Cam ci;
      {
        const cfloat* cp = cams_c + ((long)bi * (p.n + 1) + 1 + iv) * 16;  // per-view uniforms: scalar loads
#pragma unroll
        for (int i = 0; i < 9; ++i) ci.R[i] = cp[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) ci.T[i] = cp[9 + i];
        ci.f[0] = cp[12]; ci.f[1] = cp[13]; ci.c[0] = cp[14]; ci.c[1] = cp[15];
      }
      world_to_view(ci, P, G.q);
      const Corner cr = bilinear_corner(grid_coord(ci.f[0], ci.c[0], G.q[0], G.q[2]), grid_coord(ci.f[1], ci.c[1], G.q[1], G.q[2]), p.r);
      const int x0 = min(max(cr.x0, 0), p.r - 1), x1 = min(max(cr.x0 + 1, 0), p.r - 1);
      const int y0 = min(max(cr.y0, 0), p.r - 1), y1 = min(max(cr.y0 + 1, 0), p.r - 1);
      const int img = bi * p.n + iv;
      const int yimg = p.img_map ? imap_c[img] : img;
      G.pix[0] = yimg * hw + y0 * p.r + x0;
      G.pix[1] = yimg * hw + y0 * p.r + x1;
      G.pix[2] = yimg * hw + y1 * p.r + x0;
      G.pix[3] = yimg * hw + y1 * p.r + x1;
      G.w[0] = (1.f - cr.tx) * (1.f - cr.ty);
      G.w[1] = cr.tx * (1.f - cr.ty);
      G.w[2] = (1.f - cr.tx) * cr.ty;
      G.w[3] = cr.tx * cr.ty;
#pragma unroll
      for (int c = 0; c < 4; ++c) if (!((cr.mask >> c) & 1)) G.w[c] = 0.f;
      G.cv = cview_c[img];
#pragma unroll
      for (int c = 0; c < 4; ++c) G.lvv[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrsrc, (uint32_t)G.pix[c] * 4u, 0, 0));
    };
    u32x4 zp[2][2];
    auto load_zp = [&](int iv) {
      const uint32_t off = (uint32_t)((bi * p.n + iv) * hw + k) * row_bytes + lane_off;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        zp[mb][0] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, off + mb * 64, 0, 0);
        zp[mb][1] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, off + mb * 64 + 16, 0, 0);
      }
    };
    // the 16 full-line loads of a view travel as two halves through the SAME 32 registers: corners 0, 1 then corners 2, 3;
    // L[c][j] = chunk dchunk of the row of sample 8 j + dsub under corner 2 half + c
    u32x4 L[2][4];
    auto load_rows = [&](const Geo& G, int half) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int px = __builtin_amdgcn_ds_bpermute(4 * (8 * j + dsub), half ? G.pix[2 + c] : G.pix[c]);
          L[c][j] = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, (uint32_t)px * row_bytes + ld_off, 0, 0);
        }
    };
    auto store_rows = [&](int half) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(slots + (2 * half + c) * SLOT_BYTES + wr_off[j]) = L[c][j];
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x2 g[2][8];  // channel pairs: the blend / SiLU / accumulate passes run on packed fp32 instructions (v_pk_fma_f32, v_pk_mul_f32,
                    // v_pk_add_f32: two channels per issue slot, each half rounded exactly like the scalar instruction)
#pragma unroll
    for (int i = 0; i < 8; ++i) { g[0][i] = f32x2{0.f, 0.f}; g[1][i] = f32x2{0.f, 0.f}; }

    // prologue: corners 0, 1 of view 0 in their slots, corners 2, 3 in flight (the previous tile's reads of the slots are complete: its
    // blend consumed them)
    Geo cur, nxt;
    geometry(0, cur);
    load_zp(0);
    load_rows(cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    store_rows(0);
    __builtin_amdgcn_sched_barrier(0);
    load_rows(cur, 1);
    __builtin_amdgcn_sched_barrier(0);

    for (int iv = 0; iv < p.n; ++iv) {
      const bool more = iv + 1 < p.n;
      // corners 2, 3 of THIS view (requested before the previous view's SiLU pass) -> slots 2, 3, which the previous blend has released
      store_rows(1);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        geometry(iv + 1, nxt);
        load_rows(nxt, 0);  // corners 0, 1 of the next view: in flight during this view's encoding and blend
      } else {
        nxt = cur;
      }
      __builtin_amdgcn_sched_barrier(0);

      // ---- per-sample inputs in B-operand layout: lane half h handles frequencies 2*kfp + h ----
      const float qh[3] = {cur.q[0] * hs, cur.q[1] * hs, cur.q[2] * hs};
      f32x16 z[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) { z[0][i] = 0.f; z[1][i] = 0.f; }
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
            fw[pr] = pack_bf16x2(hh ? cur.q[2] : cur.q[0], hh ? 0.f : cur.q[1]);
          } else {
            fw[pr] = 0u;
          }
        }
        u32x4 fv = {fw[0], fw[1], fw[2], fw[3]};
        const bf16x8 fb = __builtin_bit_cast(bf16x8, fv);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ws + (mb * 32 + arow0) * W_PITCH + ks * 32 + hh * 16);
          z[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fb, z[mb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // one k-step per scheduling region: bounds the live sin / cos temporaries
      }

      // ---- online softmax over views (texels and constant of this view were requested one iteration ago) ----
      float logit = cur.cv;
#pragma unroll
      for (int c = 0; c < 4; ++c) logit = fmaf(cur.w[c], cur.lvv[c], logit);
      const float lg = logit * 1.4426950408889634f;
      const float m_new = fmaxf(m_run, lg);
      const float sc = __builtin_amdgcn_exp2f(m_run - m_new);
      const float a = __builtin_amdgcn_exp2f(lg - m_new);
      l_run = fmaf(l_run, sc, a);
      m_run = m_new;
      if (p.logits) {
        if (valid && hh == 0 && cc == 0) p.logits[((long)bi * p.n + iv) * npts + pt] = logit;
      }

      // ---- z += zP, then the four corners in turn (the same order of additions as nerf_fused_kernel) ----
      f32x2 zv[2][8];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t d = zp[mb][half][e];
            zv[mb][half * 4 + e] = f32x2{z[mb][half * 8 + 2 * e], z[mb][half * 8 + 2 * e + 1]} + f32x2{bf16lo_to_f32(d), bf16hi_to_f32(d)};
          }
      __builtin_amdgcn_sched_barrier(0);
      if (more) load_zp(iv + 1);  // into the registers consumed above: one iteration of flight time
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_sched_barrier(0);  // one corner per scheduling region: bounds the live read-back registers
        u32x4 y[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            y[mb][t] = *reinterpret_cast<const u32x4*>(slots + c * SLOT_BYTES + l31 * 128 + (((4 * mb + 2 * hh + t) ^ rswz) << 4));
        const f32x2 wc = {cur.w[c], cur.w[c]};
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t d = y[mb][half][e];
              zv[mb][half * 4 + e] = __builtin_elementwise_fma(wc, f32x2{bf16lo_to_f32(d), bf16hi_to_f32(d)}, zv[mb][half * 4 + e]);
            }
      }
      __builtin_amdgcn_sched_barrier(0);
      // slots 0, 1 are free (the blend has consumed every read of them): the next view's corners 0, 1, requested before the encoding, go
      // in, and its corners 2, 3 are requested into the same registers -- they land during the SiLU pass and the next view's geometry
      if (more) {
        store_rows(0);
        __builtin_amdgcn_sched_barrier(0);
        load_rows(nxt, 1);
      }
      __builtin_amdgcn_sched_barrier(0);

      // ---- g = g*sc + a*silu(z) ----
      {
        const f32x2 av = {a, a}, scv = {sc, sc}, one = {1.f, 1.f}, nl2e = {-1.4426950408889634f, -1.4426950408889634f};
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f32x2 zz = zv[mb][i];
            const f32x2 t = nl2e * zz;
            const f32x2 den = one + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            const f32x2 sv = zz * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
            g[mb][i] = __builtin_elementwise_fma(av, sv, g[mb][i] * scv);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }

    if (valid) {
      const float inv = 1.f / l_run;
      uint16_t* dst = p.g + ((long)bi * npts + pt) * p.C + ch0 + 16 * hh;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        u32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = pack_bf16x2(g[mb][e][0] * inv, g[mb][e][1] * inv);
          o1[e] = pack_bf16x2(g[mb][4 + e][0] * inv, g[mb][4 + e][1] * inv);
        }
        *reinterpret_cast<u32x4*>(dst + mb * 32) = o0;
        *reinterpret_cast<u32x4*>(dst + mb * 32 + 8) = o1;
      }
      if (p.lse && hh == 0 && cc == 0) {
        p.lse[((long)bi * npts + pt) * 2] = m_run * 0.6931471805599453f;
        p.lse[((long)bi * npts + pt) * 2 + 1] = l_run;
      }
    }
  }
}

// ---- the same operator in TWO passes (round 6) ---------------------------------------------------------------------------------------
// Of the 684 VALU instructions nerf_fused_line_kernel spends per wave and view, the projection of the sample into the view, the bilinear
// corner arithmetic, the four logit texels and the online-softmax bookkeeping do not depend on the channel slice -- yet every one of the
// C / 64 slices (10 at the 640 level, 20 at the 1280 level) repeats them (what-if build without the projection: -23 %).
//   pass 1, nerf_geom_kernel: one thread per sample, all views: q_i = view-space point, corner pixel / clamps / in-bounds mask / tx / ty
//     (cd360_geom.h: the same ordered fp32 chains, so the indices stay bit-exact), view logit, running max and sum over the views.
//     Writes one 32-byte record per (view, sample) and (max, sum) per sample; `logits` / `lse` (the training outputs) fall out of it.
//   pass 2, nerf_fused_rec_kernel: nerf_fused_line_kernel with the geometry replaced by two 16-byte loads of the record (two views
//     ahead) and the online softmax replaced by p_i = exp2(lg_i - max): the maximum is final, so the accumulator is never rescaled.
// The softmax weights are exp2(lg - m_final) / l instead of the online form's running rescales: the same value to fp32 rounding.
constexpr uint32_t REC_PIX_MASK = (1u << 26) - 1u;
constexpr bool NERF_REC_HALF_DEFAULT = false;  // which pass-2 geometry cd360_tuning.nerf_kernel = -1 selects (4 = 32 channels per workgroup)

__global__ __launch_bounds__(256) void nerf_geom_kernel(NerfParams p) {
  const int hw = p.r * p.r;
  const long npts = (long)hw * p.S;
  const int bi = blockIdx.y;
  const long pt = (long)blockIdx.x * 256 + threadIdx.x;
  if (pt >= npts) return;
  const int k = (int)(pt / p.S), s = (int)(pt - (long)k * p.S);
  float P[3];
  {
    const Cam c0 = load_cam(p.cams + (long)bi * (p.n + 1) * 16);
    float o[3], d[3];
    patch_ray(c0, p.xs[k % p.r], p.ys[k / p.r], o, d);
    const float ts = p.t[(long)k * p.t_ray_stride + s];
#pragma unroll
    for (int j = 0; j < 3; ++j) P[j] = o[j] + ts * d[j];
  }
  typedef const __attribute__((address_space(4))) float cfloat;
  typedef const __attribute__((address_space(4))) int cint;
  cfloat* const cams_c = (cfloat*)(p.cams);
  cfloat* const cview_c = (cfloat*)(p.cview);
  cint* const imap_c = (cint*)(p.img_map);
  float m_run = -INFINITY, l_run = 0.f;
  for (int iv = 0; iv < p.n; ++iv) {
    Cam ci;
    {
      const cfloat* cp = cams_c + ((long)bi * (p.n + 1) + 1 + iv) * 16;  // per-view uniforms: scalar loads
#pragma unroll
      for (int i = 0; i < 9; ++i) ci.R[i] = cp[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) ci.T[i] = cp[9 + i];
      ci.f[0] = cp[12]; ci.f[1] = cp[13]; ci.c[0] = cp[14]; ci.c[1] = cp[15];
    }
    float q[3];
    world_to_view(ci, P, q);
    const Corner cr = bilinear_corner(grid_coord(ci.f[0], ci.c[0], q[0], q[2]), grid_coord(ci.f[1], ci.c[1], q[1], q[2]), p.r);
    const int x0 = min(max(cr.x0, 0), p.r - 1), x1 = min(max(cr.x0 + 1, 0), p.r - 1);
    const int y0 = min(max(cr.y0, 0), p.r - 1), y1 = min(max(cr.y0 + 1, 0), p.r - 1);
    const int img = bi * p.n + iv;
    const int yimg = p.img_map ? imap_c[img] : img;
    const int pix0 = yimg * hw + y0 * p.r + x0;
    const int dx = x1 - x0, dy = y1 - y0;  // 0 | 1 (the clamps)
    float w[4] = {(1.f - cr.tx) * (1.f - cr.ty), cr.tx * (1.f - cr.ty), (1.f - cr.tx) * cr.ty, cr.tx * cr.ty};
#pragma unroll
    for (int c = 0; c < 4; ++c) if (!((cr.mask >> c) & 1)) w[c] = 0.f;
    const float lv0 = p.lv[pix0], lv1 = p.lv[pix0 + dx], lv2 = p.lv[pix0 + dy * p.r], lv3 = p.lv[pix0 + dy * p.r + dx];
    float logit = cview_c[img];
    logit = fmaf(w[0], lv0, logit);
    logit = fmaf(w[1], lv1, logit);
    logit = fmaf(w[2], lv2, logit);
    logit = fmaf(w[3], lv3, logit);
    const float lg = logit * 1.4426950408889634f;
    const float m_new = fmaxf(m_run, lg);
    l_run = fmaf(l_run, __builtin_amdgcn_exp2f(m_run - m_new), __builtin_amdgcn_exp2f(lg - m_new));
    m_run = m_new;
    u32x4* dst = p.rec + (((long)bi * p.n + iv) * npts + pt) * 2;
    dst[0] = u32x4{__builtin_bit_cast(uint32_t, q[0]), __builtin_bit_cast(uint32_t, q[1]), __builtin_bit_cast(uint32_t, q[2]), __builtin_bit_cast(uint32_t, lg)};
    dst[1] = u32x4{(uint32_t)pix0 | ((uint32_t)dx << 26) | ((uint32_t)dy << 27) | ((uint32_t)cr.mask << 28), __builtin_bit_cast(uint32_t, cr.tx),
                   __builtin_bit_cast(uint32_t, cr.ty), 0u};
    if (p.logits) p.logits[((long)bi * p.n + iv) * npts + pt] = logit;
  }
  p.ml[(long)bi * npts + pt] = f32x2{m_run, l_run};
  if (p.lse) {
    p.lse[((long)bi * npts + pt) * 2] = m_run * 0.6931471805599453f;
    p.lse[((long)bi * npts + pt) * 2 + 1] = l_run;
  }
}

// NB = 32-channel blocks per wave: 2 = 64 channels per workgroup at two waves per SIMD (the line kernel's geometry); 1 = 32 channels per
// workgroup: half the accumulators, row pieces and read-backs per wave, so the register budget admits more waves per SIMD -- the
// per-view chain (loads -> LDS -> blend -> SiLU) is latency-bound at two -- against sin / cos generated twice per 64 channels.
template <int NB>
__global__ __launch_bounds__(256, NB == 1 ? 3 : 2) void nerf_fused_rec_kernel(NerfParams p) {
  constexpr int CNB = 32 * NB;            // channels per workgroup
  constexpr int ROWB = 64 * NB;           // bytes of one table row's slice
  constexpr int LPR = 4 * NB;             // loader lanes per row (16 bytes each)
  constexpr int RPI = 64 / LPR;           // rows per load instruction
  constexpr int PC = 32 / RPI;            // load instructions (pieces) per corner
  constexpr int SLOT = 32 * ROWB;         // one corner of 32 samples
  constexpr int WLDS = 4 * SLOT;          // four corners per wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_line[];
  unsigned char* const Ws = smem_line + 4 * WLDS;            // CNB x W_PITCH
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const slots = smem_line + wave * WLDS;
  const int hw = p.r * p.r;
  const long npts = (long)hw * p.S;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int cc = wg / (p.b * p.ngroups);
  const int rem = wg - cc * (p.b * p.ngroups);
  const int bi = rem / p.ngroups, grp = rem - bi * p.ngroups;
  const int ch0 = cc * CNB;

  for (int idx = tid; idx < CNB * (KP / 8); idx += 256) {
    const int row = idx / (KP / 8), c8 = idx - row * (KP / 8);
    *reinterpret_cast<u32x4*>(Ws + row * W_PITCH + c8 * 16) = *reinterpret_cast<const u32x4*>(p.Wk + (long)(ch0 + row) * KP + c8 * 8);
  }
  __syncthreads();

  const float hs = hh ? 2.f : 1.f;
  const int arow0 = chan_pos(l31);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Y, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.zP, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.rec, 0, 0xffffffff, 0x00020000);
  const uint32_t row_bytes = (uint32_t)p.C * 2u;
  const uint32_t lane_off = (uint32_t)(ch0 + 16 * hh) * 2u;
  // loader role: piece j moves samples RPI j .. RPI j + RPI - 1; this lane: sample RPI j + lane / LPR, chunk lane % LPR of its row slice.
  // LDS image: sample s, chunk q at s * ROWB + ((q ^ swz(s)) << 4), swz = (s >> 1) & 7 over 8 chunks | (s >> 2) & 3 over 4 (conflict-free
  // ds_read_b128 of one chunk of 32 consecutive samples either way)
  const int dsub = lane / LPR, dchunk = lane % LPR;
  const uint32_t ld_off = (uint32_t)(ch0 * 2 + dchunk * 16);
  int wr_off[PC];
#pragma unroll
  for (int j = 0; j < PC; ++j) {
    const int smp = RPI * j + dsub;
    wr_off[j] = smp * ROWB + ((dchunk ^ (NB == 2 ? (smp >> 1) & 7 : (smp >> 2) & 3)) << 4);
  }
  const int rswz = NB == 2 ? (l31 >> 1) & 7 : (l31 >> 2) & 3;

  for (int tw = 0; tw < TILES_PER_WAVE; ++tw) {
    const long pt0 = (long)grp * PTS_PER_WG + (wave * TILES_PER_WAVE + tw) * 32;
    if (pt0 >= npts) break;  // wave-uniform
    const long pt = pt0 + l31;
    const bool valid = pt < npts;
    const long ptc = valid ? pt : npts - 1;
    const int k = (int)(ptc / p.S);
    // this sample's 32-byte record of view iv: byte offset rec_off + iv * rec_view (buffer loads, like every other table of this kernel)
    const uint32_t rec_off = (uint32_t)(((long)bi * p.n * npts + ptc) * 32), rec_view = (uint32_t)(npts * 32);
    const f32x2 mlv = p.ml[(long)bi * npts + ptc];
    const float m_fin = mlv[0];

    struct Geo {
      float q[3], w[4], lg;
      int pix[4];
    };
    auto decode = [&](const u32x4& r0, const u32x4& r1, Geo& G) {
      // (every element goes through a uint32_t VALUE first: __builtin_bit_cast applied directly to an element of a vector reference
      // reads element 0 whatever the index -- hipcc 7.2)
      const uint32_t u0 = r0[0], u1 = r0[1], u2 = r0[2], u3 = r0[3], bits = r1[0], v1 = r1[1], v2 = r1[2];
      G.q[0] = __builtin_bit_cast(float, u0); G.q[1] = __builtin_bit_cast(float, u1); G.q[2] = __builtin_bit_cast(float, u2);
      G.lg = __builtin_bit_cast(float, u3);
      const int base = (int)(bits & REC_PIX_MASK), dx = (int)((bits >> 26) & 1u), dyr = ((bits >> 27) & 1u) ? p.r : 0;
      G.pix[0] = base; G.pix[1] = base + dx; G.pix[2] = base + dyr; G.pix[3] = base + dyr + dx;
      const float tx = __builtin_bit_cast(float, v1), ty = __builtin_bit_cast(float, v2);
      G.w[0] = (1.f - tx) * (1.f - ty);
      G.w[1] = tx * (1.f - ty);
      G.w[2] = (1.f - tx) * ty;
      G.w[3] = tx * ty;
#pragma unroll
      for (int c = 0; c < 4; ++c) if (!((bits >> (28 + c)) & 1u)) G.w[c] = 0.f;
    };
    u32x4 zp[NB][2];
    auto load_zp = [&](int iv) {
      const uint32_t off = (uint32_t)((bi * p.n + iv) * hw + k) * row_bytes + lane_off;
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        zp[mb][0] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, off + mb * 64, 0, 0);
        zp[mb][1] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, off + mb * 64 + 16, 0, 0);
      }
    };
    u32x4 L[2][PC];
    auto load_rows = [&](const Geo& G, int half) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < PC; ++j) {
          const int px = __builtin_amdgcn_ds_bpermute(4 * (RPI * j + dsub), half ? G.pix[2 + c] : G.pix[c]);
          L[c][j] = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, (uint32_t)px * row_bytes + ld_off, 0, 0);
        }
    };
    auto store_rows = [&](int half) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < PC; ++j) *reinterpret_cast<u32x4*>(slots + (2 * half + c) * SLOT + wr_off[j]) = L[c][j];
    };

    f32x2 g[NB][8];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
      for (int i = 0; i < 8; ++i) g[mb][i] = f32x2{0.f, 0.f};

    // prologue: records of views 0 and 1 decoded, the record of view 2 requested; corners 0, 1 of view 0 in their slots, 2, 3 in flight
    Geo cur, nxt;
    u32x4 ra0, ra1;  // raw record two views ahead
    {
      const u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rec_off, 0, 0), a1 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rec_off + 16, 0, 0);
      decode(a0, a1, cur);
      const uint32_t o1 = rec_off + (p.n > 1 ? rec_view : 0u);
      const u32x4 b0 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, o1, 0, 0), b1 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, o1 + 16, 0, 0);
      decode(b0, b1, nxt);
      const uint32_t o2 = rec_off + (p.n > 2 ? 2u * rec_view : 0u);
      ra0 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, o2, 0, 0);
      ra1 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, o2 + 16, 0, 0);
    }
    load_zp(0);
    load_rows(cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    store_rows(0);
    __builtin_amdgcn_sched_barrier(0);
    load_rows(cur, 1);
    __builtin_amdgcn_sched_barrier(0);

    for (int iv = 0; iv < p.n; ++iv) {
      const bool more = iv + 1 < p.n;
      store_rows(1);
      __builtin_amdgcn_sched_barrier(0);
      if (more) load_rows(nxt, 0);  // corners 0, 1 of the next view: in flight during this view's encoding and blend
      __builtin_amdgcn_sched_barrier(0);

      // ---- per-sample inputs in B-operand layout: lane half h handles frequencies 2*kfp + h ----
      const float qh[3] = {cur.q[0] * hs, cur.q[1] * hs, cur.q[2] * hs};
      f32x16 z[NB];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) z[mb][i] = 0.f;
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
            fw[pr] = pack_bf16x2(hh ? cur.q[2] : cur.q[0], hh ? 0.f : cur.q[1]);
          } else {
            fw[pr] = 0u;
          }
        }
        u32x4 fv = {fw[0], fw[1], fw[2], fw[3]};
        const bf16x8 fb = __builtin_bit_cast(bf16x8, fv);
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ws + (mb * 32 + arow0) * W_PITCH + ks * 32 + hh * 16);
          z[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fb, z[mb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }

      // ---- softmax weight of this view: the maximum over the views is already final ----
      const float a = __builtin_amdgcn_exp2f(cur.lg - m_fin);

      // ---- z += zP, then the four corners in turn (the order of additions of nerf_fused_kernel) ----
      f32x2 zv[NB][8];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t d = zp[mb][half][e];
            zv[mb][half * 4 + e] = f32x2{z[mb][half * 8 + 2 * e], z[mb][half * 8 + 2 * e + 1]} + f32x2{bf16lo_to_f32(d), bf16hi_to_f32(d)};
          }
      __builtin_amdgcn_sched_barrier(0);
      if (more) load_zp(iv + 1);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_sched_barrier(0);
        u32x4 y[NB][2];
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            y[mb][t] = *reinterpret_cast<const u32x4*>(slots + c * SLOT + l31 * ROWB + (((4 * mb + 2 * hh + t) ^ rswz) << 4));
        const f32x2 wc = {cur.w[c], cur.w[c]};
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
          for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t d = y[mb][half][e];
              zv[mb][half * 4 + e] = __builtin_elementwise_fma(wc, f32x2{bf16lo_to_f32(d), bf16hi_to_f32(d)}, zv[mb][half * 4 + e]);
            }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        store_rows(0);
        __builtin_amdgcn_sched_barrier(0);
        load_rows(nxt, 1);
      }
      __builtin_amdgcn_sched_barrier(0);

      // ---- g += a * silu(z) ----
      {
        const f32x2 av = {a, a}, one = {1.f, 1.f}, nl2e = {-1.4426950408889634f, -1.4426950408889634f};
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f32x2 zz = zv[mb][i];
            const f32x2 t = nl2e * zz;
            const f32x2 den = one + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            const f32x2 sv = zz * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
            g[mb][i] = __builtin_elementwise_fma(av, sv, g[mb][i]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the record requested two views ago has had a whole iteration to land: decode it, request the next one
      cur = nxt;
      if (iv + 2 < p.n) decode(ra0, ra1, nxt);
      if (iv + 3 < p.n) {
        ra0 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rec_off + (uint32_t)(iv + 3) * rec_view, 0, 0);
        ra1 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rec_off + (uint32_t)(iv + 3) * rec_view + 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    if (valid) {
      const float inv = 1.f / mlv[1];
      uint16_t* dst = p.g + ((long)bi * npts + pt) * p.C + ch0 + 16 * hh;
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        u32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = pack_bf16x2(g[mb][e][0] * inv, g[mb][e][1] * inv);
          o1[e] = pack_bf16x2(g[mb][4 + e][0] * inv, g[mb][4 + e][1] * inv);
        }
        *reinterpret_cast<u32x4*>(dst + mb * 32) = o0;
        *reinterpret_cast<u32x4*>(dst + mb * 32 + 8) = o1;
      }
    }
  }
}

#ifdef CD360_WHATIF  // probe builds only (tools/probe/README.md): kept as the record of what LDS-DMA row gathers cost and need
// ---- the same operator with the corner rows gathered by LDS-DMA -----------------------------------------------------------------------
// nerf_fused_kernel is bound by the vector-memory RETURN path, not by its arithmetic (TD_TD_BUSY = 100 % of the kernel's CU-cycles;
// builds without the sin / cos or without the SiLU transcendentals run no faster): with lane = sample every gather instruction returns
// 32 different cache lines, 32 bytes of each, and every line is visited by four instructions.  Here the four corner rows of a view
// travel as FULL 128-byte lines: buffer_load_dwordx4 ... lds, 8 lanes per row (lane i of piece j fetches 16-byte chunk i % 8 of sample
// 8 j + i / 8, whose table pixel it gets from that sample's lane by ds_bpermute), 16 instructions x 8 lines per wave and view instead of
// 32 x 32 line returns.  The rows land in a wave-private ring of four 4-KB slots (one per corner; 16-byte chunks XOR-swizzled by the sample
// on the SOURCE side: the destination of a DMA is lane-linear) and are read back in the compute layout with conflict-free ds_read_b128.
// No gather registers are left, so the loop is software-pipelined for free: corner c of view iv+1 is requested as soon as corner c of
// view iv has been blended (its slot), the geometry of view iv+1 having been computed before view iv's encoding; one vmcnt(0) + workgroup
// barrier per view stands between the pieces' arrival and their first ds_read (see the comment there).  zP (one or two distinct rows
// per wave: consecutive samples share their ray) and the four logit texels stay ordinary loads, consumed one iteration after issue.
#define NERF_LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int RING_BYTES = 4 * 32 * 128;  // four corner slots of 32 samples x 128 B per wave

__global__ __launch_bounds__(256, 2) void nerf_fused_dma_kernel(NerfParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_dma[];
  unsigned char* const ring_all = smem_dma;                 // 4 waves x RING_BYTES
  unsigned char* const Ws = smem_dma + 4 * RING_BYTES;      // CN x W_PITCH
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const ring = ring_all + wave * RING_BYTES;
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

  const float hs = hh ? 2.f : 1.f;
  const int arow0 = chan_pos(l31);
  typedef const __attribute__((address_space(4))) float cfloat;
  typedef const __attribute__((address_space(4))) int cint;
  cfloat* const cams_c = (cfloat*)(p.cams);
  cfloat* const cview_c = (cfloat*)(p.cview);
  cint* const imap_c = (cint*)(p.img_map);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Y, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.zP, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.lv, 0, 0xffffffff, 0x00020000);
  const uint32_t row_bytes = (uint32_t)p.C * 2u;
  const uint32_t lane_off = (uint32_t)(ch0 + 16 * hh) * 2u;  // this lane's 32 bytes of a 64-channel slice (block mb: + 64 B)
  // DMA lane geometry: piece j moves samples 8 j .. 8 j + 7; this lane: sample 8 j + lane / 8, LDS chunk lane % 8 <- source chunk ^ swizzle
  const int dsub = lane >> 3, dchunk = lane & 7;
  // read-back: chunk 4 mb + 2 hh + t of sample l31 sits at chunk position (id ^ ((l31 >> 1) & 7))
  const int rswz = (l31 >> 1) & 7;

  for (int tw = 0; tw < TILES_PER_WAVE; ++tw) {
    const long pt0 = (long)grp * PTS_PER_WG + (wave * TILES_PER_WAVE + tw) * 32;
    if (pt0 >= npts) break;  // wave-uniform
    const long pt = pt0 + l31;
    const bool valid = pt < npts;
    const long ptc = valid ? pt : npts - 1;
    const int k = (int)(ptc / p.S), s = (int)(ptc - (long)k * p.S);
    float P[3];
    {
      const Cam c0 = load_cam(p.cams + (long)bi * (p.n + 1) * 16);
      float o[3], d[3];
      patch_ray(c0, p.xs[k % p.r], p.ys[k / p.r], o, d);
      const float ts = p.t[(long)k * p.t_ray_stride + s];
#pragma unroll
      for (int j = 0; j < 3; ++j) P[j] = o[j] + ts * d[j];
    }

    struct Geo {
      float q[3], w[4], lvv[4], cv;  // lvv: the four gathered logit texels, raw (combined one iteration after their loads were issued)
      int pix[4];                    // table pixel (image included) of the four corners
    };
    auto geometry = [&](int iv, Geo& G) {  // projection, corners, weights of view iv; issues its four logit-texel loads
      // (the per-view uniforms are read through the constant address space: scalar loads on the LDS / SMEM counter -- as vector loads
      // they would, vmcnt being in-order, wait for every row piece in flight)
      Cam ci;
      {
        const cfloat* cp = cams_c + ((long)bi * (p.n + 1) + 1 + iv) * 16;
#pragma unroll
        for (int i = 0; i < 9; ++i) ci.R[i] = cp[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) ci.T[i] = cp[9 + i];
        ci.f[0] = cp[12]; ci.f[1] = cp[13]; ci.c[0] = cp[14]; ci.c[1] = cp[15];
      }
      world_to_view(ci, P, G.q);
      const Corner cr = bilinear_corner(grid_coord(ci.f[0], ci.c[0], G.q[0], G.q[2]), grid_coord(ci.f[1], ci.c[1], G.q[1], G.q[2]), p.r);
      const int x0 = min(max(cr.x0, 0), p.r - 1), x1 = min(max(cr.x0 + 1, 0), p.r - 1);
      const int y0 = min(max(cr.y0, 0), p.r - 1), y1 = min(max(cr.y0 + 1, 0), p.r - 1);
      const int img = bi * p.n + iv;
      const int yimg = p.img_map ? imap_c[img] : img;
      G.pix[0] = yimg * hw + y0 * p.r + x0;
      G.pix[1] = yimg * hw + y0 * p.r + x1;
      G.pix[2] = yimg * hw + y1 * p.r + x0;
      G.pix[3] = yimg * hw + y1 * p.r + x1;
      G.w[0] = (1.f - cr.tx) * (1.f - cr.ty);
      G.w[1] = cr.tx * (1.f - cr.ty);
      G.w[2] = (1.f - cr.tx) * cr.ty;
      G.w[3] = cr.tx * cr.ty;
#pragma unroll
      for (int c = 0; c < 4; ++c) if (!((cr.mask >> c) & 1)) G.w[c] = 0.f;
      G.cv = cview_c[img];
#pragma unroll
      for (int c = 0; c < 4; ++c) G.lvv[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrsrc, (uint32_t)G.pix[c] * 4u, 0, 0));
    };
    u32x4 zp[2][2];
    auto load_zp = [&](int iv) {
      const uint32_t off = (uint32_t)((bi * p.n + iv) * hw + k) * row_bytes + lane_off;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        zp[mb][0] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, off + mb * 64, 0, 0);
        zp[mb][1] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, off + mb * 64 + 16, 0, 0);
      }
    };
    // The row pieces of a view are addressed per DMA lane (sample 8 j + lane / 8): the 16 cross-lane fetches of the corner pixels are
    // done in one batch right after the geometry (one LDS round trip), not piece by piece in front of each DMA instruction
    int dpix[4][4];
    auto dma_pixels = [&](const Geo& G) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) dpix[c][j] = __builtin_amdgcn_ds_bpermute(4 * (8 * j + dsub), G.pix[c]);
    };
    auto dma_corner = [&](int c) {  // 4 pieces of 8 full rows into slot c
      // the four source offsets are computed into four DIFFERENT registers before the first piece is issued (the fence keeps them all
      // live): with one register recomputed between back-to-back pieces, rows landed wrong now and then
      uint32_t off[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int smp = 8 * j + dsub;
        off[j] = (uint32_t)dpix[c][j] * row_bytes + (uint32_t)(ch0 * 2 + ((dchunk ^ ((smp >> 1) & 7)) << 4));
      }
      asm volatile("" : "+v"(off[0]), "+v"(off[1]), "+v"(off[2]), "+v"(off[3]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, NERF_LDS_AS3(ring + c * 4096 + j * 1024), 16, off[j], 0, 0, 0);
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 g[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { g[0][i] = 0.f; g[1][i] = 0.f; }

    // prologue: view 0 requested (the previous tile's last reads of the ring are complete: its blend consumed them)
    Geo cur, nxt;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    geometry(0, cur);
    load_zp(0);
    dma_pixels(cur);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) dma_corner(c);
    __builtin_amdgcn_sched_barrier(0);

    for (int iv = 0; iv < p.n; ++iv) {
      const bool more = iv + 1 < p.n;
      if (more) {
        geometry(iv + 1, nxt);
        dma_pixels(nxt);
      } else {
        nxt = cur;
      }
      __builtin_amdgcn_sched_barrier(0);

      // ---- per-sample inputs in B-operand layout: lane half h handles frequencies 2*kfp + h ----
      const float qh[3] = {cur.q[0] * hs, cur.q[1] * hs, cur.q[2] * hs};
      f32x16 z[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) { z[0][i] = 0.f; z[1][i] = 0.f; }
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
            fw[pr] = pack_bf16x2(hh ? cur.q[2] : cur.q[0], hh ? 0.f : cur.q[1]);
          } else {
            fw[pr] = 0u;
          }
        }
        u32x4 fv = {fw[0], fw[1], fw[2], fw[3]};
        const bf16x8 fb = __builtin_bit_cast(bf16x8, fv);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ws + (mb * 32 + arow0) * W_PITCH + ks * 32 + hh * 16);
          z[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fb, z[mb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // one k-step per scheduling region: bounds the live sin / cos temporaries
      }

      // ---- online softmax over views (texels and constant of this view were requested one iteration ago) ----
      float logit = cur.cv;
#pragma unroll
      for (int c = 0; c < 4; ++c) logit = fmaf(cur.w[c], cur.lvv[c], logit);
      const float lg = logit * 1.4426950408889634f;
      const float m_new = fmaxf(m_run, lg);
      const float sc = __builtin_amdgcn_exp2f(m_run - m_new);
      const float a = __builtin_amdgcn_exp2f(lg - m_new);
      l_run = fmaf(l_run, sc, a);
      m_run = m_new;
      if (p.logits) {
        if (valid && hh == 0 && cc == 0) p.logits[((long)bi * p.n + iv) * npts + pt] = logit;
      }

      // ---- z += zP, then the four corners in turn (the same order of additions as nerf_fused_kernel) ----
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            z[mb][half * 8 + 2 * e] += bf16lo_to_f32(zp[mb][half][e]);
            z[mb][half * 8 + 2 * e + 1] += bf16hi_to_f32(zp[mb][half][e]);
          }
      // All row pieces of this view (requested one view ago, corner by corner) have landed: vmcnt(0), then the workgroup barrier the
      // LDS-DMA recipe prescribes between the wait and the first ds_read: vmcnt drops when the data has been handed to the LDS, not
      // when it is readable (with a per-corner counted wait straight in front of the reads, a few rows per hundred launches came
      // back stale; 128 idle cycles after the wait hid it).  One rendezvous per view; waves that ran out of tiles
      // have exited and do not count.
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (more) load_zp(iv + 1);  // into the registers consumed above: one iteration of flight time
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_sched_barrier(0);
        u32x4 y[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            y[mb][t] = *reinterpret_cast<const u32x4*>(ring + c * 4096 + l31 * 128 + (((4 * mb + 2 * hh + t) ^ rswz) << 4));
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              z[mb][half * 8 + 2 * e] = fmaf(cur.w[c], bf16lo_to_f32(y[mb][half][e]), z[mb][half * 8 + 2 * e]);
              z[mb][half * 8 + 2 * e + 1] = fmaf(cur.w[c], bf16hi_to_f32(y[mb][half][e]), z[mb][half * 8 + 2 * e + 1]);
            }
        __builtin_amdgcn_sched_barrier(0);  // (the fmas above consumed the ds_reads: the slot is free)
        if (more) dma_corner(c);
      }
      __builtin_amdgcn_sched_barrier(0);

      // ---- g = g*sc + a*silu(z) ----
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float zz = z[mb][r];
          const float sv = zz * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * zz));
          g[mb][r] = fmaf(a, sv, g[mb][r] * sc);
        }
      cur = nxt;
    }

    if (valid) {
      const float inv = 1.f / l_run;
      uint16_t* dst = p.g + ((long)bi * npts + pt) * p.C + ch0 + 16 * hh;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        u32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = pack_bf16x2(g[mb][2 * e] * inv, g[mb][2 * e + 1] * inv);
          o1[e] = pack_bf16x2(g[mb][8 + 2 * e] * inv, g[mb][8 + 2 * e + 1] * inv);
        }
        *reinterpret_cast<u32x4*>(dst + mb * 32) = o0;
        *reinterpret_cast<u32x4*>(dst + mb * 32 + 8) = o1;
      }
      if (p.lse && hh == 0 && cc == 0) {
        p.lse[((long)bi * npts + pt) * 2] = m_run * 0.6931471805599453f;
        p.lse[((long)bi * npts + pt) * 2 + 1] = l_run;
      }
    }
  }
}

#endif  // CD360_WHATIF

// (view, ray)-only inputs of plane_coefs.0: [enc8(plucker(target ray in ref-i frame)) 96 | dir 3]
// (nerfsd_pytorch3d.py:104-112,130-131; utils_cameraray.py:201-242,270-292).  out [b, n, hw, 104] fp32 (99 + 5 zero pad)
// BF = false: out [b, n, hw, 104] fp32 (99 + 5 zero pad); BF = true: out [b, n, hw, 128] bf16 (99 + 29 zero pad) -- the A operand of
// the zP = [enc8(plucker), dir] Wp^T + b1 table GEMM on cd360_gemm_bf16 (K % 64 == 0), written once instead of fp32 + a cast pass
template <bool BF>
__global__ void plucker_features_kernel(const float* __restrict__ cams, const float* __restrict__ xs, const float* __restrict__ ys,
                                        void* __restrict__ out_, int b, int n, int r) {
  const int hw = r * r;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)b * n * hw) return;
  const int k = (int)(gid % hw), iv = (int)((gid / hw) % n), bi = (int)(gid / ((long)hw * n));
  const Cam c0 = load_cam(cams + (long)bi * (n + 1) * 16);
  const Cam ci = load_cam(cams + ((long)bi * (n + 1) + 1 + iv) * 16);
  float o[3], d[3], co[3], cd[3];
  patch_ray(c0, xs[k % r], ys[k / r], o, d);
  world_to_view(ci, o, co);
  rotate_to_view(ci, d, cd);
  const float nrm = sqrtf(fmaf(cd[2], cd[2], fmaf(cd[1], cd[1], cd[0] * cd[0])));
  float v[6] = {cd[0] / nrm, cd[1] / nrm, cd[2] / nrm, 0.f, 0.f, 0.f};
  v[3] = co[1] * v[2] - co[2] * v[1];
  v[4] = co[2] * v[0] - co[0] * v[2];
  v[5] = co[0] * v[1] - co[1] * v[0];
  // every value goes from registers to its place in the row: no per-thread feature array (a 104-float array indexed inside the
  // partially unrolled sin/cos loops lived in 432 B of scratch)
  uint32_t* dst16 = reinterpret_cast<uint32_t*>(out_) + gid * 64;
  float* dst32 = reinterpret_cast<float*>(out_) + gid * 104;
#pragma unroll 1
  for (int kf = 0; kf < 8; ++kf) {
    const float freq = __builtin_bit_cast(float, (uint32_t)((127 + kf - 4) << 23)) * 3.14159274101257324f;
    float sn[6], cs[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float arg = v[c] * freq;
      sn[c] = sinf(arg);
      cs[c] = cosf(arg);
    }
    if (BF) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dst16[kf * 3 + c] = pack_bf16x2(sn[2 * c], sn[2 * c + 1]);
        dst16[24 + kf * 3 + c] = pack_bf16x2(cs[2 * c], cs[2 * c + 1]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        dst32[kf * 6 + c] = sn[c];
        dst32[48 + kf * 6 + c] = cs[c];
      }
    }
  }
  if (BF) {
    dst16[48] = pack_bf16x2(cd[0], cd[1]);
    dst16[49] = pack_bf16x2(cd[2], 0.f);
#pragma unroll
    for (int j = 50; j < 64; ++j) dst16[j] = 0u;
  } else {
    dst32[96] = cd[0]; dst32[97] = cd[1]; dst32[98] = cd[2];
#pragma unroll
    for (int j = 99; j < 104; ++j) dst32[j] = 0.f;
  }
}

}  // namespace

static int plucker_launch(bool bf, const void* cams, const void* xs, const void* ys, void* out, int b, int n, int r, void* stream) {
  if (!cams || !xs || !ys || !out || b <= 0 || n <= 0 || r <= 0) return CD360_ERR_ARG;
  const long total = (long)b * n * r * r;
  if (bf) hipLaunchKernelGGL(plucker_features_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                             (const float*)cams, (const float*)xs, (const float*)ys, out, b, n, r);
  else hipLaunchKernelGGL(plucker_features_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                          (const float*)cams, (const float*)xs, (const float*)ys, out, b, n, r);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

extern "C" int cd360_plucker_features(const void* cams, const void* xs, const void* ys, void* out, int b, int n, int r, void* stream) {
  CD360_TUNE_SCOPE(stream);
  return plucker_launch(false, cams, xs, ys, out, b, n, r, stream);
}

// The same features as bf16 rows of 128 (99 values + zero pad): the A operand of the zP table GEMM on cd360_gemm_bf16
extern "C" int cd360_plucker_features_bf16(const void* cams, const void* xs, const void* ys, void* out, int b, int n, int r, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if ((uintptr_t)out % 16) return CD360_ERR_ARG;
  return plucker_launch(true, cams, xs, ys, out, b, n, r, stream);
}

extern "C" int cd360_nerf_k_padded(void) { return KP; }

// Workspace of the two-pass form: 32 bytes per (batch, view, sample) + 8 per (batch, sample)
extern "C" int64_t cd360_nerf_ws_bytes(int b, int n, int r, int S) {
  if (b <= 0 || n <= 0 || r <= 0 || S <= 0) return 0;
  const int64_t npts = (int64_t)r * r * S;
  return (int64_t)b * n * npts * 32 + (int64_t)b * npts * 8;
}

static int nerf_launch_single(NerfParams& p, void* stream);

// cd360_nerf_mlp_aggregate in two passes (see nerf_geom_kernel): `ws` = cd360_nerf_ws_bytes(b, n, r, S) bytes of scratch, 16-byte aligned
// (written and read inside the call).  Shapes outside the two-pass envelope (table pixels beyond 2^26, 32-bit table offsets) and
// cd360_tuning.nerf_kernel = 0 | 1 run the one-pass kernels instead; results agree to fp32 rounding of the softmax weights.
extern "C" int cd360_nerf_mlp_aggregate_ws(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, const void* Y,
                                           const void* zP, const void* lv, const void* cview, const void* Wk, const void* img_map, void* g,
                                           void* logits, void* lse, int b, int n, int r, int S, int C, int ntab, void* ws, int64_t ws_bytes,
                                           void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!cams || !xs || !ys || !t || !Y || !zP || !lv || !cview || !Wk || !g || !ws) return CD360_ERR_ARG;
  if (b <= 0 || n <= 0 || r <= 0 || S <= 0 || C <= 0 || C % CN || ntab <= 0) return CD360_ERR_SHAPE;
  if (t_ray_stride != 0 && t_ray_stride != S) return CD360_ERR_SHAPE;
  if (((uintptr_t)Y | (uintptr_t)zP | (uintptr_t)Wk | (uintptr_t)g | (uintptr_t)ws) % 16) return CD360_ERR_ARG;
  if (ws_bytes < cd360_nerf_ws_bytes(b, n, r, S)) return CD360_ERR_ARG;
  NerfParams p;
  p.cams = (const float*)cams; p.xs = (const float*)xs; p.ys = (const float*)ys; p.t = (const float*)t;
  p.Y = (const uint16_t*)Y; p.zP = (const uint16_t*)zP; p.lv = (const float*)lv; p.cview = (const float*)cview;
  p.Wk = (const uint16_t*)Wk; p.img_map = (const int*)img_map; p.g = (uint16_t*)g; p.logits = (float*)logits; p.lse = (float*)lse;
  p.b = b; p.n = n; p.r = r; p.S = S; p.C = C; p.t_ray_stride = t_ray_stride;
  p.ncc = C / CN;
  const long npts = (long)r * r * S;
  p.ngroups = (int)((npts + PTS_PER_WG - 1) / PTS_PER_WG);
  p.rec = (u32x4*)ws;
  p.ml = (f32x2*)((unsigned char*)ws + (int64_t)b * n * npts * 32);
  const long nwg = (long)p.ncc * b * p.ngroups;
  if (nwg > 0x7fffffffL) return CD360_ERR_SHAPE;
  const int variant = cd360_tune().nerf_kernel;
  const bool fits = (long)ntab * r * r < (1L << 26) && (long)ntab * r * r * C * 2 < (1L << 32) && (long)b * n * r * r * C * 2 < (1L << 32) &&
                    (long)b * n * npts * 32 < (1L << 32);
  if (!fits || variant == 0 || variant == 1 || variant == 2) return nerf_launch_single(p, stream);
  hipLaunchKernelGGL(nerf_geom_kernel, dim3((unsigned)((npts + 255) / 256), (unsigned)b), dim3(256), 0, (hipStream_t)stream, p);
  CD360_LAUNCH_CHECK();
  // cd360_tuning.nerf_kernel: 3 = 64 channels per workgroup (two waves per SIMD), 4 = 32 channels per workgroup (three); -1 = the measured default
  if (variant == 4 || (variant < 0 && NERF_REC_HALF_DEFAULT)) {
    constexpr int LDS_REC = 4 * 4 * 32 * 64 + 32 * W_PITCH;
    p.ncc = C / 32;
    const long nwg1 = (long)p.ncc * b * p.ngroups;
    if (nwg1 > 0x7fffffffL) return CD360_ERR_SHAPE;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_fused_rec_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_REC);
    if (attr != hipSuccess) return CD360_ERR_LAUNCH;
    hipLaunchKernelGGL(nerf_fused_rec_kernel<1>, dim3((unsigned)nwg1), dim3(256), LDS_REC, (hipStream_t)stream, p);
    CD360_LAUNCH_CHECK();
    return CD360_OK;
  }
  constexpr int LDS_LINE = 4 * WAVE_LDS + CN * W_PITCH;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_fused_rec_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LINE);
  if (attr != hipSuccess) return CD360_ERR_LAUNCH;
  hipLaunchKernelGGL(nerf_fused_rec_kernel<2>, dim3((unsigned)nwg), dim3(256), LDS_LINE, (hipStream_t)stream, p);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// See NerfParams for layouts.  C must be a multiple of 64.  img_map / logits / lse may be NULL.
extern "C" int cd360_nerf_mlp_aggregate(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, const void* Y,
                                        const void* zP, const void* lv, const void* cview, const void* Wk, const void* img_map, void* g,
                                        void* logits, void* lse, int b, int n, int r, int S, int C, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!cams || !xs || !ys || !t || !Y || !zP || !lv || !cview || !Wk || !g) return CD360_ERR_ARG;
  if (b <= 0 || n <= 0 || r <= 0 || S <= 0 || C <= 0 || C % CN) return CD360_ERR_SHAPE;
  if (t_ray_stride != 0 && t_ray_stride != S) return CD360_ERR_SHAPE;
  if (((uintptr_t)Y | (uintptr_t)zP | (uintptr_t)Wk | (uintptr_t)g) % 16) return CD360_ERR_ARG;
  NerfParams p;
  p.cams = (const float*)cams; p.xs = (const float*)xs; p.ys = (const float*)ys; p.t = (const float*)t;
  p.Y = (const uint16_t*)Y; p.zP = (const uint16_t*)zP; p.lv = (const float*)lv; p.cview = (const float*)cview;
  p.Wk = (const uint16_t*)Wk; p.img_map = (const int*)img_map; p.g = (uint16_t*)g; p.logits = (float*)logits; p.lse = (float*)lse;
  p.b = b; p.n = n; p.r = r; p.S = S; p.C = C; p.t_ray_stride = t_ray_stride;
  p.ncc = C / CN;
  const long npts = (long)r * r * S;
  p.ngroups = (int)((npts + PTS_PER_WG - 1) / PTS_PER_WG);
  p.rec = nullptr;
  p.ml = nullptr;
  return nerf_launch_single(p, stream);
}

static int nerf_launch_single(NerfParams& p, void* stream) {
  const int b = p.b, n = p.n, r = p.r, C = p.C;
  const long nwg = (long)p.ncc * b * p.ngroups;
  if (nwg > 0x7fffffffL) return CD360_ERR_SHAPE;
#ifdef CD360_WHATIF
  // cd360_tuning.nerf_kernel = 2 (probe builds) selects the LDS-DMA row-gather kernel (tables addressable with 32-bit byte offsets).  It is
  // correct (bit-identical) only with the vmcnt(0) + workgroup barrier between the pieces' arrival and their first ds_read, and with that
  // rendezvous per view it measures 0 % (1280 channels) / -5 % (640 channels) against the register-gather kernel.
  if (cd360_tune().nerf_kernel == 2 && (long)b * n * r * r * C * 2 < (1L << 32)) {
    constexpr int LDS_DMA = 4 * RING_BYTES + CN * W_PITCH;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_fused_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DMA);
    if (attr != hipSuccess) return CD360_ERR_LAUNCH;
    hipLaunchKernelGGL(nerf_fused_dma_kernel, dim3((unsigned)nwg), dim3(256), LDS_DMA, (hipStream_t)stream, p);
    CD360_LAUNCH_CHECK();
    return CD360_OK;
  }
#endif
  // cd360_tuning.nerf_kernel: 0 = register gathers (32-byte pieces of 32 lines per instruction), 1 = full-line gathers staged through the
  // wave's LDS block (tables addressable with 32-bit byte offsets); -1 = the measured default
  const int variant = cd360_tune().nerf_kernel;
  const bool lines = (variant == 1 || (variant < 0 && NERF_LINE_DEFAULT)) && (long)b * n * r * r * C * 2 < (1L << 32) && (long)b * n * r * r * 4 < (1L << 32);
  if (lines) {
    constexpr int LDS_LINE = 4 * WAVE_LDS + CN * W_PITCH;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_fused_line_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LINE);
    if (attr != hipSuccess) return CD360_ERR_LAUNCH;
    hipLaunchKernelGGL(nerf_fused_line_kernel, dim3((unsigned)nwg), dim3(256), LDS_LINE, (hipStream_t)stream, p);
    CD360_LAUNCH_CHECK();
    return CD360_OK;
  }
  hipLaunchKernelGGL(nerf_fused_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}
