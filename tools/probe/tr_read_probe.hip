#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
// LDS image: row-major [64 rows][64 cols] of uint16, value = row*64+col.  Each lane passes address of (row = lane % 16 ... ) we try a few mappings.
__global__ void probe(const int* rowsel, const int* colsel, uint16_t* out) {
  __shared__ uint16_t lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  const uint16_t* p = lds + rowsel[lane] * 64 + colsel[lane];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
  int hr[64], hc[64];
  // mapping A: lane l -> row = l % 16 (key), col = 4 * (l / 16) (d group)
  for (int l = 0; l < 64; ++l) { hr[l] = l % 16; hc[l] = 4 * (l / 16); }
  int *dr, *dc; uint16_t* dout; uint16_t hout[256];
  hipMalloc(&dr, 256); hipMalloc(&dc, 256); hipMalloc(&dout, 512);
  hipMemcpy(dr, hr, 256, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 256, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dr, dc, dout);
  hipMemcpy(hout, dout, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d (row %2d col %2d):", l, hr[l], hc[l]); for (int j = 0; j < 4; ++j) printf(" r%dc%d", hout[l*4+j] / 64, hout[l*4+j] % 64); printf("\n"); }
  return 0;
}
