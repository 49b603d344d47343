// The UNet's output convolution (openaimodel.py:967-973: conv_nd(dims, model_channels, out_channels = 4, 3, padding = 1) behind GroupNorm +
// SiLU) as its own small MFMA kernel.  On the GEMM core a 3 x 3 convolution with FOUR output channels occupied a 256 x 256 tile per 256
// pixels with 240 of its 256 columns empty and streamed its input nine times: 153 us of a 26 ms denoise step for 0.45 GFLOP.  Here
//   partial[p', tap * 4 + co] = sum_c act[p', c] w[co, c, tap]        one [pixels x 36] x K = Cin product over the tile's pixels AND its halo rows
//   out[p, co] = bias[co] + sum_tap partial[p + shift(tap), tap * 4 + co]                                  nine shifted reads of the LDS
// A workgroup owns two image rows (+ one halo row above and below): 4 W <= 512 pixel rows of the activation travel ONCE, 64 channels at
// a time, by LDS-DMA into the XOR-swizzled image the GEMM core uses (rows outside the image are past the descriptor's end: zeros, and so are
// their partials -- the padding needs no test); eight waves of 64 pixels x 64 partial columns on v_mfma_f32_32x32x16_bf16; the fp32
// partials go through the LDS once.  Output: bf16 rows [images, H W, 4], what cd360_cfg_euler_step_cl reads.
#include "cd360_common.h"

namespace {

#define OC_LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))
#define OC_BARRIER()                     \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

__device__ __forceinline__ int oc_chan_pos(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

constexpr int OC_XB = 512 * 128;          // one activation stage: 512 pixel rows x 64 channels
constexpr int OC_WB = 64 * 128;           // one weight stage: 64 partial columns x 64 channels
constexpr int OC_STAGE = OC_XB + OC_WB;   // 72 KB, two stages
constexpr int OC_PPITCH = 40;             // floats per pixel row of the partial image (36 used)

// x [images, H W, Cin] bf16; w [64, Cin] bf16: row tap * 4 + co = w[co, :, tap], rows 36 .. 63 zero; bias fp32 [4]; out [images, H W, 4] bf16
__global__ __launch_bounds__(512) void out_conv4_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const float* __restrict__ bias,
                                                        uint16_t* __restrict__ out, int H, int W, int Cin) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bands = H >> 1, img = blockIdx.x / bands, y0 = (blockIdx.x - img * bands) * 2;  // output rows y0, y0 + 1
  const int npx = 4 * W;                 // pixel rows of the tile (halo row, two rows, halo row)
  const int nk = Cin >> 6;
  const long HW = (long)H * W;

  // ---- DMA geometry: piece j of a stage = tile pixel rows 64 j + 8 wave + lane / 8; LDS chunk lane % 8 <- source chunk ^ swizzle ----
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)img * HW * Cin), 0, (int)(HW * Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 64 * Cin * 2, 0x00020000);
  const int srow = wave * 8 + (lane >> 3);
  const int schunk = (lane & 7) ^ ((srow >> 1) & 7);
  uint32_t xoff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = 64 * j + srow;                       // tile pixel row
    const int yy = y0 - 1 + r / W, xx = r - (r / W) * W;
    xoff[j] = (r < npx && yy >= 0 && yy < H) ? (uint32_t)((((long)yy * W + xx) * Cin) * 2 + schunk * 16) : 0x80000000u;
  }
  const uint32_t woff = (uint32_t)(srow * Cin * 2 + schunk * 16);
  auto issue = [&](int c) {  // K chunk c into stage c & 1: eight activation pieces + one weight piece per wave
    unsigned char* st = lds + (c & 1) * OC_STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, OC_LDS_AS3(st + j * 8192), 16, xoff[j] == 0x80000000u ? xoff[j] : xoff[j] + (uint32_t)(c * 128), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, OC_LDS_AS3(st + OC_XB), 16, woff + (uint32_t)(c * 128), 0, 0, 0);
  };

  // ---- fragments: this wave's 64 pixel rows (two blocks of 32) against the 64 partial columns (two blocks of 32) ----
  const int cp = oc_chan_pos(l31);
  f32x16 acc[2][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][mb][i] = 0.f;

  issue(0);
  if (nk > 1) issue(1);
  for (int c = 0; c < nk; ++c) {
    if (c + 1 < nk) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // chunk c landed, chunk c + 1 (nine pieces) may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OC_BARRIER();  // (a raw s_barrier: __syncthreads() would drain the chunk still in flight -- its LDS-DMA is a pending LDS write)
    const unsigned char* xs = lds + (c & 1) * OC_STAGE;
    const unsigned char* ws = xs + OC_XB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fw[2], fx[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int row = nb * 32 + cp;
        fw[nb] = *reinterpret_cast<const bf16x8*>(ws + row * 128 + (((2 * ks + hh) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int row = wave * 64 + mb * 32 + l31;
        fx[mb] = *reinterpret_cast<const bf16x8*>(xs + row * 128 + (((2 * ks + hh) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nb], fx[mb], acc[nb][mb], 0, 0, 0);
    }
    OC_BARRIER();  // every wave is past its reads of this stage (its MFMAs have consumed them)
    if (c + 2 < nk) issue(c + 2);
  }

  // ---- partials -> LDS [pixel row][40] fp32 (lane: pixel mb * 32 + l31 of its wave; registers: columns nb * 32 + 16 hh + r) ----
  float* P = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    float* dst = P + (wave * 64 + mb * 32 + l31) * OC_PPITCH;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f32x4*>(dst + 16 * hh + 4 * q) = f32x4{acc[0][mb][4 * q], acc[0][mb][4 * q + 1], acc[0][mb][4 * q + 2], acc[0][mb][4 * q + 3]};
    if (hh == 0) *reinterpret_cast<f32x4*>(dst + 32) = f32x4{acc[1][mb][0], acc[1][mb][1], acc[1][mb][2], acc[1][mb][3]};
  }
  __syncthreads();

  // ---- out[p, co] = bias + nine shifted partials; thread = (output pixel of the two rows, all four channels) ----
  if (tid < 2 * W) {
    const int ry = tid / W, xx = tid - ry * W;  // tile pixel row of the output pixel: (1 + ry) W + xx
    float o[4] = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - 3 * dy;
      const int sx = xx + dx - 1;
      if (sx < 0 || sx >= W) continue;  // (rows above / below the image hold zero partials: no test)
      const f32x4 v = *reinterpret_cast<const f32x4*>(P + ((ry + dy) * W + sx) * OC_PPITCH + tap * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += v[e];
    }
    const u32x2 r = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    *reinterpret_cast<u32x2*>(out + (((long)img * H + y0 + ry) * W + xx) * 4) = r;
  }
}

}  // namespace

// x [images, H W, Cin] bf16 channels-last (the GroupNorm + SiLU output); w36 [64, Cin] bf16 with row tap * 4 + co = weight[co, :, ky, kx]
// (tap = 3 ky + kx), rows 36 .. 63 zero; bias fp32 [4] -> out [images, H W, 4] bf16.  W in {32, 64, 128}, H even, Cin % 64 == 0.
extern "C" int cd360_out_conv4_bf16(const void* x, const void* w36, const void* bias, void* out, int images, int H, int W, int Cin, void* stream) {
  if (!x || !w36 || !bias || !out || images <= 0) return CD360_ERR_ARG;
  if (H <= 0 || (H & 1) || (W != 32 && W != 64 && W != 128) || Cin <= 0 || Cin % 64) return CD360_ERR_SHAPE;
  if (((uintptr_t)x | (uintptr_t)w36) % 16 || (uintptr_t)out % 8 || (uintptr_t)bias % 4) return CD360_ERR_ARG;
  if ((long)H * W * Cin * 2 >= (1L << 31)) return CD360_ERR_SHAPE;
  constexpr int LDS = 2 * OC_STAGE;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&out_conv4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) return CD360_ERR_LAUNCH;
  hipLaunchKernelGGL(out_conv4_kernel, dim3((unsigned)(images * (H / 2))), dim3(512), LDS, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)w36,
                     (const float*)bias, (uint16_t*)out, H, W, Cin);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}
