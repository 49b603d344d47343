// Layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands, unit block scales) on gfx950:
//   assumed: A lane l holds row (l & 31), k = 32 (l >> 5) + 4 j + t for byte t of dword j; B lane l holds column (l & 31), same k;
//            D register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column (l & 31)  (the 32x32 accumulator layout of the bf16 MFMAs).
// Prints the max abs difference against a host matmul of the same fp8 values.  hipcc --offload-arch=gfx950 f8_mfma_probe.hip -o f8_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const unsigned char* A, const unsigned char* B, float* D) {  // A [32][64], B [64][32] fp8 bytes
  const int l = threadIdx.x, row = l & 31, kg = l >> 5;
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) {
    unsigned av = 0, bv = 0;
    for (int t = 0; t < 4; ++t) {
      const int k = 32 * kg + 4 * j + t;
      av |= (unsigned)A[row * 64 + k] << (8 * t);
      bv |= (unsigned)B[k * 32 + row] << (8 * t);
    }
    a[j] = (int)av;
    b[j] = (int)bv;
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f, 0, 0x7f);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + row] = acc[r];
}

static float e4m3(unsigned char v) {  // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main() {
  unsigned char hA[32 * 64], hB[64 * 32];
  srand(1);
  for (int i = 0; i < 2048; ++i) {  // finite values of moderate size: exponent field 4 .. 10, any mantissa, any sign
    hA[i] = (unsigned char)(((rand() & 1) << 7) | ((4 + rand() % 7) << 3) | (rand() & 7));
    hB[i] = (unsigned char)(((rand() & 1) << 7) | ((4 + rand() % 7) << 3) | (rand() & 7));
  }
  unsigned char *dA, *dB;
  float* dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  float hD[1024];
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double worst = 0, big = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0;
      for (int k = 0; k < 64; ++k) s += (double)e4m3(hA[i * 64 + k]) * e4m3(hB[k * 32 + j]);
      worst = fmax(worst, fabs(s - hD[i * 32 + j]));
      big = fmax(big, fabs(s));
    }
  printf("f8f6f4 32x32x64 probe: max |D - ref| = %g (max |ref| %g) -> %s\n", worst, big, worst <= 1e-4 * big ? "LAYOUT CONFIRMED" : "LAYOUT MISMATCH");
  return worst <= 1e-4 * big ? 0 : 1;
}
