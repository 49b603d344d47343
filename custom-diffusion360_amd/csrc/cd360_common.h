// Shared device helpers for the cd360 HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define CD360_OK 0
#define CD360_ERR_ARG (-1)
#define CD360_ERR_SHAPE (-2)
#define CD360_ERR_LAUNCH (-3)

#define CD360_LAUNCH_CHECK()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return CD360_ERR_LAUNCH;            \
  } while (0)

// bf16 <-> f32 (bit tricks; bf16 is the top half of an f32)
__device__ __forceinline__ float bf16lo_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t packed) { return __builtin_bit_cast(float, packed & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __builtin_bit_cast(float, ((uint32_t)v) << 16); }
// round-to-nearest-even pack of two floats -> one dword of 2 x bf16 (v_cvt_pk_bf16_f32); lo = a
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float a) { return (uint16_t)(pack_bf16x2(a, 0.f) & 0xffffu); }

// XCD-aware block remap (8 XCDs; block b is dispatched to XCD b % 8): give each XCD a contiguous
// range of logical tiles so neighbouring tiles share that XCD's L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
