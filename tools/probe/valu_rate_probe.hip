// Issue rate of a few VALU instructions on gfx950 (cycles per wave64 instruction, one and two waves per SIMD), measured with
// s_memtime around an unrolled independent stream.  hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(INS) \
  for (int it = 0; it < iters; ++it) { asm volatile(REP16(INS) REP16(INS) REP16(INS) REP16(INS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(p0), "+v"(p1) : "v"(c0), "v"(c1)); }
template <int WHICH>
__global__ void probe(float* out, long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b0 = 0.5f, b1 = 0.25f, c0 = 1.0001f, c1 = 0.9999f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3};
  const long t0 = __builtin_readcyclecounter();
  if (WHICH == 0) BODY("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n")
  if (WHICH == 1) BODY("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
  if (WHICH == 2) BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
  if (WHICH == 3) BODY("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n")
  if (WHICH == 4) BODY("v_pk_fma_f32 %6, %6, %6, %7\n v_pk_fma_f32 %7, %7, %7, %6\n v_pk_fma_f32 %6, %6, %7, %7\n v_pk_fma_f32 %7, %7, %6, %6\n")
  if (WHICH == 5) BODY("v_dot2_f32_bf16 %0, %4, %5, %0\n v_dot2_f32_bf16 %1, %4, %5, %1\n v_dot2_f32_bf16 %2, %4, %5, %2\n v_dot2_f32_bf16 %3, %4, %5, %3\n")
  if (WHICH == 6) BODY("v_cvt_f32_bf16 %0, %4\n v_cvt_f32_bf16 %1, %5\n v_cvt_f32_bf16 %2, %4\n v_cvt_f32_bf16 %3, %5\n")
  if (WHICH == 7) BODY("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n")
  if (WHICH == 8) BODY("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0\n")
  if (WHICH == 9) BODY("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1\n")
  const long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + p0[0] + p1[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int W> void run(const char* name, float* out, long* cyc) {
  for (int waves = 1; waves <= 2; ++waves) {  // waves per SIMD on one CU (a single workgroup of 4 or 8 waves)
    const int iters = 64;
    hipLaunchKernelGGL(probe<W>, dim3(1), dim3(256 * waves), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-20s %d wave(s)/SIMD: %6.2f shader-clock ticks per instruction per wave (readcyclecounter)\n", name, waves, (double)c / (iters * 256.0));
  }
}
int main() {
  float* out; long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
  run<0>("v_fma_f32", out, cyc); run<1>("v_exp_f32", out, cyc); run<2>("v_rcp_f32", out, cyc); run<3>("v_sin_f32", out, cyc);
  run<4>("v_pk_fma_f32", out, cyc); run<5>("v_dot2_f32_bf16", out, cyc); run<6>("v_cvt_f32_bf16", out, cyc); run<7>("v_mul_lo_u32", out, cyc);
  run<8>("v_cvt_pk_bf16_f32", out, cyc); run<9>("v_max3_f32", out, cyc);
  return 0;
}
