// L2 -> LDS DMA throughput probe (buffer_load_dwordx4 ... lds): every workgroup streams `nk` K-tiles of a GEMM-like operand pair
// (BM + BN rows x 64 bf16) into two LDS buffers exactly like gemm8p.hip does, with a vmcnt(0) + barrier per tile and nothing else.
// hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))
template <int NW, int MODE>  // MODE 0: swizzled rows (gemm8p pattern), 1: linear rows, 2: fully linear 1 KiB per wave-instruction
__global__ __launch_bounds__(64 * NW) void dma_kernel(const uint16_t* a, const uint16_t* w, int K, int tiles_n, int nk, int depth, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  constexpr int PR = 8 * NW, XP = 256 / PR, NP = 2 * XP;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tm = tiles_n > 0 ? blockIdx.x / tiles_n : (blockIdx.x & 7) % (-tiles_n), tn = tiles_n > 0 ? blockIdx.x % tiles_n : 0;  // tiles_n <= 0: every XCD streams the same -tiles_n A tiles and one W tile (L2-resident)
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
  const int srow = wave * 8 + (lane >> 3);
  const int chunk = MODE == 0 ? ((lane & 7) ^ ((srow >> 1) & 7)) : (lane & 7);
  uint32_t xoff = (uint32_t)(((long)tm * 256 + srow) * K * 2 + chunk * 16), woff = (uint32_t)(((long)tn * 256 + srow) * K * 2 + chunk * 16);
  if (MODE == 2) { xoff = (uint32_t)(((long)tm * 256 * K * 2) + wave * 1024 + lane * 16); woff = (uint32_t)(((long)tn * 256 * K * 2) + wave * 1024 + lane * 16); }
  const uint32_t step = MODE == 2 ? NW * 1024 : PR * K * 2, kstep = MODE == 2 ? 65536 / 2 : 128;
  unsigned char* base = lds + wave * 1024;
  auto tile = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (i < XP) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, LDS_AS3(base + buf * 32768 + i * PR * 128), 16, xoff + kt * kstep + i * step, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, LDS_AS3(base + 65536 + buf * 32768 + (i - XP) * PR * 128), 16, woff + kt * kstep + (i - XP) * step, 0, 0, 0);
    }
  };
  tile(0, 0);
  if (depth > 1) tile(1, 1);
  for (int t = 0; t < nk; ++t) {
    if (depth > 1) { if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + depth < nk) tile(t + depth, t & 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(float*)(lds + 64);
}
template <int NW, int MODE> void run(const char* name, uint16_t* a, uint16_t* w, float* sink, int depth, int resident = 0) {
  const int M = 4096, N = 4096, K = 4096, nk = K / 64, nwg = (M / 256) * (N / 256), tiles_n = resident ? -resident : N / 256;
  hipFuncSetAttribute((const void*)&dma_kernel<NW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((dma_kernel<NW, MODE>), dim3(nwg), dim3(64 * NW), 131072, 0, a, w, K, tiles_n, nk, depth, sink);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((dma_kernel<NW, MODE>), dim3(nwg), dim3(64 * NW), 131072, 0, a, w, K, tiles_n, nk, depth, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const double bytes = (double)nwg * nk * 65536;
  printf("%-28s waves %d depth %d: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU@2.1GHz\n", name, NW, depth, ms * 1e3, bytes / ms * 1e-9, bytes / 256 / (ms * 1e-3 * 2.1e9));
}

// Mixed feed: each operand either by LDS-DMA or by buffer_load_dwordx4 into registers + ds_write_b128 one tile later (RX / RW = via registers)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
template <int NW, bool RX, bool RW>
__global__ __launch_bounds__(64 * NW) void mix_kernel(const uint16_t* a, const uint16_t* w, int K, int tiles_n, int nk, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  constexpr int PR = 8 * NW, XP = 256 / PR;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
  const int srow = wave * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ ((srow >> 1) & 7);
  const uint32_t xoff = (uint32_t)(((long)tm * 256 + srow) * K * 2 + chunk * 16), woff = (uint32_t)(((long)tn * 256 + srow) * K * 2 + chunk * 16);
  const uint32_t step = PR * K * 2;
  unsigned char* base = lds + wave * 1024;
  unsigned char* wbase = lds + wave * 1024 + lane * 16;
  u32x4 rx[XP], rw[XP];
  auto tile = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      if (RX) rx[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, xoff + kt * 128 + i * step, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, LDS_AS3(base + buf * 32768 + i * PR * 128), 16, xoff + kt * 128 + i * step, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      if (RW) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff + kt * 128 + i * step, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, LDS_AS3(base + 65536 + buf * 32768 + i * PR * 128), 16, woff + kt * 128 + i * step, 0, 0, 0);
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      if (RX) *reinterpret_cast<u32x4*>(wbase + buf * 32768 + i * PR * 128) = rx[i];
      if (RW) *reinterpret_cast<u32x4*>(wbase + 65536 + buf * 32768 + i * PR * 128) = rw[i];
    }
  };
  tile(0, 0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    put(t & 1);
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nk) tile(t + 1, (t + 1) & 1);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(float*)(lds + 64);
}
template <int NW, bool RX, bool RW> void run_mix(const char* name, uint16_t* a, uint16_t* w, float* sink) {
  const int M = 4096, N = 4096, K = 4096, nk = K / 64, tiles_n = N / 256, nwg = (M / 256) * tiles_n;
  hipFuncSetAttribute((const void*)&mix_kernel<NW, RX, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix_kernel<NW, RX, RW>), dim3(nwg), dim3(64 * NW), 131072, 0, a, w, K, tiles_n, nk, sink);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mix_kernel<NW, RX, RW>), dim3(nwg), dim3(64 * NW), 131072, 0, a, w, K, tiles_n, nk, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const double bytes = (double)nwg * nk * 65536;
  printf("%-28s waves %d (depth 1): %7.1f us  %6.2f TB/s  %5.1f B/clk/CU@2.1GHz\n", name, NW, ms * 1e3, bytes / ms * 1e-9, bytes / 256 / (ms * 1e-3 * 2.1e9));
}
int main() {
  uint16_t *a, *w; float* sink;
  hipMalloc(&a, 4096L * 4096 * 2 + 65536); hipMalloc(&w, 4096L * 4096 * 2 + 65536); hipMalloc(&sink, 4096);
  hipMemset(a, 1, 4096L * 4096 * 2); hipMemset(w, 2, 4096L * 4096 * 2);
  run<4, 0>("swizzled rows", a, w, sink, 2); run<8, 0>("swizzled rows", a, w, sink, 2);
  run<4, 1>("linear rows", a, w, sink, 2); run<8, 1>("linear rows", a, w, sink, 2);
  run<4, 2>("contiguous 1 KiB", a, w, sink, 2); run<8, 2>("contiguous 1 KiB", a, w, sink, 2);
  run<4, 0>("swizzled rows", a, w, sink, 1); run<8, 0>("swizzled rows", a, w, sink, 1);
  run<8, 0>("L2-resident (1 A tile)", a, w, sink, 2, 1); run<8, 0>("L2-resident (2 A tiles)", a, w, sink, 2, 2);
  run<4, 0>("L2-resident (1 A tile)", a, w, sink, 2, 1); run<8, 2>("L2-resident contiguous", a, w, sink, 2, 1);
  run_mix<8, false, false>("mix: DMA + DMA", a, w, sink);
  run_mix<8, true, false>("mix: regs + DMA", a, w, sink);
  run_mix<8, true, true>("mix: regs + regs", a, w, sink);
  run_mix<4, true, false>("mix: regs + DMA", a, w, sink);
  run_mix<4, true, true>("mix: regs + regs", a, w, sink);
  return 0;
}
