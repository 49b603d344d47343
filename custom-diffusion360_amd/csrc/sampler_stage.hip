// The two ends of one captured denoise step (sample.py:331-349 through EulerEDMSampler.sampler_step,
// sampling.py:85-136, DiscreteDenoiser.network_inputs, denoiser.py:47-79, and UNetModel.forward's first lines, openaimodel.py:1006-1030):
//
//   stage-in   x3 = [x | x | x];  h0 = conv3x3(bf16(c_in x3), 4 -> 320) + b        the UNet's input convolution on the scaled latent
//              emb_act = silu(time_embed(t_emb(idx(sigma))) + label_emb(y))        the input of every ResBlock's emb_layers Linear
//   step-out   x <- x + (x - d0) / sigma (sigma' - sigma),  d0 = the 3-way CFG combine of x - sigma eps_b
//
// Everything that depends on the STEP only -- sigma, sigma', c_in, the time-embedding row -- is a row of two tables the sampler fills
// once per schedule (cd360/job.py: Sampler.prepare); the kernels pick their row through a device-side step index, so a captured step
// holds no scalar arithmetic at all: round 5's graph spent ~45 torch-issued micro-kernels per step on it (sub / abs / argmin / index /
// pow / sqrt / reciprocal / arange / exp / sin / cos / cat / silu / casts / the 4 -> 64 channel pad of the input convolution).
// The input convolution is computed ONCE per diffusion sample and written to its three CFG branches (their inputs are identical).
#include "cd360_common.h"

namespace {

constexpr int TP = 64;  // pixels of one image row per workgroup

// x [bs, 4, H, W] fp32 (NCHW); tab [nsteps, 4] fp32 = (sigma, sigma_next, c_in, -); step: device int32; w [36][Cout] fp32 (k = tap * 4 + ci,
// values already rounded to bf16); bias [Cout] fp32; h [rep * bs, H * W, Cout] bf16 (branch r of sample s = image r * bs + s);
// temb [nsteps, E] bf16, lab [rep * bs, E] bf16, emb_act [rep * bs, E] bf16.  Blocks [0, nconv) convolve, the rest build emb_act.
__global__ __launch_bounds__(256) void unet_stage_in_kernel(const float* __restrict__ x, const float* __restrict__ tab, const int* __restrict__ step,
                                                            const float* __restrict__ w, const float* __restrict__ bias, uint16_t* __restrict__ h,
                                                            const uint16_t* __restrict__ temb, const uint16_t* __restrict__ lab,
                                                            uint16_t* __restrict__ emb_act, int bs, int rep, int H, int W, int Cout, int E, int nconv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int idx = *step;
  if ((int)blockIdx.x >= nconv) {  // ---- emb_act rows: 8 values per thread ----
    const long total = (long)rep * bs * (E >> 3);
    for (long i = (long)(blockIdx.x - nconv) * 256 + tid; i < total; i += (long)(gridDim.x - nconv) * 256) {
      const long row = i / (E >> 3);
      const int c8 = (int)(i - row * (E >> 3));
      const u32x4 a = *reinterpret_cast<const u32x4*>(temb + (long)idx * E + c8 * 8);
      const u32x4 b = *reinterpret_cast<const u32x4*>(lab + row * E + c8 * 8);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // emb = time_embed(..) + label_emb(y) is a bf16 tensor in the module path: round the sum, then SiLU of the rounded value
        const uint32_t s = pack_bf16x2(bf16lo_to_f32(a[e]) + bf16lo_to_f32(b[e]), bf16hi_to_f32(a[e]) + bf16hi_to_f32(b[e]));
        const float v0 = bf16lo_to_f32(s), v1 = bf16hi_to_f32(s);
        o[e] = pack_bf16x2(v0 / (1.f + __expf(-v0)), v1 / (1.f + __expf(-v1)));
      }
      *reinterpret_cast<u32x4*>(emb_act + row * E + c8 * 8) = o;
    }
    return;
  }
  float* const wl = reinterpret_cast<float*>(smem);               // [36][Cout]
  float* const xs = wl + 36 * Cout;                               // [3][TP + 2][4]: the row band with its halo, scaled and bf16-rounded
  const int tiles_w = (W + TP - 1) / TP;
  const int tw = blockIdx.x % tiles_w, y = (blockIdx.x / tiles_w) % H, s = blockIdx.x / (tiles_w * H);
  const int x0 = tw * TP;
  const float c_in = tab[idx * 4 + 2];
  for (int i = tid; i < 36 * Cout; i += 256) wl[i] = w[i];
  for (int i = tid; i < 3 * (TP + 2) * 4; i += 256) {
    const int ci = i & 3, px = (i >> 2) % (TP + 2), ry = (i >> 2) / (TP + 2);
    const int yy = y + ry - 1, xx = x0 + px - 1;
    float v = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = bf16_to_f32(f32_to_bf16(x[(((long)s * 4 + ci) * H + yy) * W + xx] * c_in));
    xs[i] = v;
  }
  __syncthreads();
  const int chunks = Cout >> 3;
  const long HW = (long)H * W;
  for (int item = tid; item < TP * chunks; item += 256) {
    const int px = item / chunks, ch = item - px * chunks;
    if (x0 + px >= W) continue;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias[ch * 8 + e];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const f32x4 xin = *reinterpret_cast<const f32x4*>(xs + ((tap / 3) * (TP + 2) + px + tap % 3) * 4);
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const float* wk = wl + (tap * 4 + ci) * Cout + ch * 8;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wk), w1 = *reinterpret_cast<const f32x4*>(wk + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[e] = fmaf(xin[ci], w0[e], acc[e]);
          acc[4 + e] = fmaf(xin[ci], w1[e], acc[4 + e]);
        }
      }
    }
    const u32x4 o = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
    for (int r = 0; r < rep; ++r)
      *reinterpret_cast<u32x4*>(h + (((long)r * bs + s) * HW + (long)y * W + x0 + px) * Cout + ch * 8) = o;
  }
}

// x [bs, 4, HW] fp32, updated IN PLACE; eps [3 bs, HW, ld] bf16 channels-last (channels 0..3 of each pixel row; u | ic | c thirds)
__global__ __launch_bounds__(256) void cfg_euler_step_cl_kernel(float* __restrict__ x, const uint16_t* __restrict__ eps, const float* __restrict__ tab,
                                                                const int* __restrict__ step, float scale, float scale_im, int bs, long HW, int ld) {
  const int idx = *step;
  const float s = tab[idx * 4], sn = tab[idx * 4 + 1];
  const long total = (long)bs * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long smp = i / HW, px = i - smp * HW;
    const u32x2 eu = *reinterpret_cast<const u32x2*>(eps + ((0 * bs + smp) * HW + px) * ld);
    const u32x2 ei = *reinterpret_cast<const u32x2*>(eps + ((1 * bs + smp) * HW + px) * ld);
    const u32x2 ec = *reinterpret_cast<const u32x2*>(eps + ((2 * bs + smp) * HW + px) * ld);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float e_u = (c & 1) ? bf16hi_to_f32(eu[c >> 1]) : bf16lo_to_f32(eu[c >> 1]);
      const float e_i = (c & 1) ? bf16hi_to_f32(ei[c >> 1]) : bf16lo_to_f32(ei[c >> 1]);
      const float e_c = (c & 1) ? bf16hi_to_f32(ec[c >> 1]) : bf16lo_to_f32(ec[c >> 1]);
      float* xp = x + (smp * 4 + c) * HW + px;
      const float xv = *xp;
      const float du = xv - s * e_u, dic = xv - s * e_i, dc = xv - s * e_c;  // (the arithmetic and its order: cfg_euler_step_kernel)
      const float d0 = du + scale * (dc - dic) + scale_im * (dic - du);
      *xp = xv + (xv - d0) / s * (sn - s);
    }
  }
}

}  // namespace

// See the kernel for layouts.  Cout % 8 == 0, E % 8 == 0; w_k36 = the input convolution's weight as [36, Cout] fp32 (k = (ky * 3 + kx) * 4 + ci).
extern "C" int cd360_unet_stage_in(const void* x, const void* step_tab, const void* step, const void* w_k36, const void* bias, void* h,
                                   const void* temb_tab, const void* lab, void* emb_act, int bs, int rep, int H, int W, int Cout, int E,
                                   void* stream) {
  if (!x || !step_tab || !step || !w_k36 || !bias || !h || !temb_tab || !lab || !emb_act) return CD360_ERR_ARG;
  if (bs <= 0 || rep <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 8 || E <= 0 || E % 8) return CD360_ERR_SHAPE;
  if (((uintptr_t)h | (uintptr_t)temb_tab | (uintptr_t)lab | (uintptr_t)emb_act | (uintptr_t)w_k36) % 16) return CD360_ERR_ARG;
  const int lds = (36 * Cout + 3 * (TP + 2) * 4) * 4;
  if (lds > 64 * 1024) return CD360_ERR_SHAPE;
  const int nconv = bs * H * ((W + TP - 1) / TP);
  const int nemb = (int)(((long)rep * bs * (E >> 3) + 255) / 256);
  hipLaunchKernelGGL(unet_stage_in_kernel, dim3((unsigned)(nconv + nemb)), dim3(256), lds, (hipStream_t)stream, (const float*)x,
                     (const float*)step_tab, (const int*)step, (const float*)w_k36, (const float*)bias, (uint16_t*)h, (const uint16_t*)temb_tab,
                     (const uint16_t*)lab, (uint16_t*)emb_act, bs, rep, H, W, Cout, E, nconv);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// x [bs, 4, HW] fp32 in place; eps [3 bs, HW, ld] bf16 (ld >= 4, ld % 4 == 0: the 320 -> 4 output convolution writes 16-channel rows)
extern "C" int cd360_cfg_euler_step_cl(void* x, const void* eps, const void* step_tab, const void* step, float scale, float scale_im, int bs,
                                       int64_t HW, int ld, void* stream) {
  if (!x || !eps || !step_tab || !step || bs <= 0 || HW <= 0 || ld < 4 || ld % 4) return CD360_ERR_ARG;
  if ((uintptr_t)eps % 8) return CD360_ERR_ARG;
  const long total = (long)bs * HW;
  hipLaunchKernelGGL(cfg_euler_step_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float*)x,
                     (const uint16_t*)eps, (const float*)step_tab, (const int*)step, scale, scale_im, bs, (long)HW, ld);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}
