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
