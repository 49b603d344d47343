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
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
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
template <int NW, int MODE> void run(const char* name, uint16_t* a, uint16_t* w, float* sink, int depth) {
  const int M = 4096, N = 4096, K = 4096, nk = K / 64, tiles_n = N / 256, nwg = (M / 256) * tiles_n;
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
int main() {
  uint16_t *a, *w; float* sink;
  hipMalloc(&a, 4096L * 4096 * 2 + 65536); hipMalloc(&w, 4096L * 4096 * 2 + 65536); hipMalloc(&sink, 4096);
  hipMemset(a, 1, 4096L * 4096 * 2); hipMemset(w, 2, 4096L * 4096 * 2);
  run<4, 0>("swizzled rows", a, w, sink, 2); run<8, 0>("swizzled rows", a, w, sink, 2);
  run<4, 1>("linear rows", a, w, sink, 2); run<8, 1>("linear rows", a, w, sink, 2);
  run<4, 2>("contiguous 1 KiB", a, w, sink, 2); run<8, 2>("contiguous 1 KiB", a, w, sink, 2);
  run<4, 0>("swizzled rows", a, w, sink, 1); run<8, 0>("swizzled rows", a, w, sink, 1);
  return 0;
}
