// out[N, K] = A[M, N]^T @ B[M, K]: the weight gradient of a Linear (dW = dY^T X, nn.Linear layout [out_features, in_features]) -- the
// "TN" member of the GEMM family, for the fine-tuning path (BASELINE config 4).  The reference gets these products from torch autograd
// of nn.Linear / its own per-sample Linear (sgm/modules/attention.py:515-516 pose_emb_layers, sgm/modules/nerfsd_pytorch3d.py:40-51
// plane_coefs / nviews / decoder, trained by diffusion.py:139-144 trainkeys = pose).
//
// Both operands are row-major over the CONTRACTION index m, so neither is in MFMA fragment order in memory.  A 64-row K-tile of each
// operand travels L2 -> LDS by LDS-DMA as 128-byte rows (one 64-column panel per 8 KB, two panels per operand) with the 32-byte XOR
// swizzle of the attention kernels' V tile on the per-lane SOURCE address, and fragments are taken with gfx950's transposing LDS read
// (ds_read_b64_tr_b16): lane i of a 16-lane group receives, for ITS column i, four consecutive rows -- two reads give the eight
// contraction values of a v_mfma_f32_32x32x16_bf16 operand.  The same (permuted) m order on both operands, so the contraction is exact.
// Tile 128 (n) x 128 (k), four waves of 64 x 64, two LDS stages (64 KB: two workgroups per CU), one barrier per K-tile.
// M is split over `slabs` workgroups per output tile (M is 10^4 .. 10^6 rows against a 640^2 .. 1280 x 2560 output: without the split
// only 25-200 workgroups would exist); each writes an fp32 partial tile, a second kernel sums the slabs in a FIXED order (deterministic,
// no atomics) and rounds once to bf16 (or leaves fp32).
#include "cd360_common.h"

namespace {

struct TnParams {
  const uint16_t* a;  // [M, N] bf16, row stride lda
  const uint16_t* b;  // [M, K] bf16, row stride ldb
  float* part;        // [slabs, N, K] fp32 partial sums (slabs == 1: the final fp32 result when out_bf16 is null)
  long lda, ldb;
  int M, N, K;
  int tiles_n, tiles_k, slabs, mtiles_per_slab;
};

#define TN_LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int TBM = 64;                 // contraction rows per K-tile
constexpr int PANEL = 64 * 128;         // one 64-column panel of a K-tile: 64 rows of 128 bytes
constexpr int STAGE = 4 * PANEL;        // A panels 0, 1 then B panels 0, 1

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// fragment of a panel for the 32-column block `cb` (0 | 1) and the 16-row group kk (0 .. 3): rows 16 kk + 4 hh + {0..3} and + 8
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* panel, int off_cb, int kk) {
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(panel + off_cb + kk * 16 * 128));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(panel + off_cb + kk * 16 * 128 + 8 * 128));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;  // this wave: A panel wr (64 n) x B panel wc (64 k)

  // work item: [slab][tile_n][tile_k], slab slowest so that the workgroups of one slab (sharing their A / B rows) run together
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int per_slab = p.tiles_n * p.tiles_k;
  const int slab = wg / per_slab, rem = wg - slab * per_slab;
  const int tn = rem / p.tiles_k, tk = rem - tn * p.tiles_k;
  const int n0 = tn * 128, k0 = tk * 128;
  const int mt0 = slab * p.mtiles_per_slab;
  const int total_mt = (p.M + TBM - 1) / TBM;
  const int mt1 = mt0 + p.mtiles_per_slab < total_mt ? mt0 + p.mtiles_per_slab : total_mt;
  const int nt = mt1 - mt0;

  // LDS-DMA geometry: one wave instruction = 8 rows x 128 B; the wave's 8 pieces per panel cover rows 8 j + lane / 8 ... per panel 64
  // rows = 8 pieces, 4 panels = 32 pieces per K-tile, 8 per wave (wave w moves panel w).  LDS chunk lane % 8 of row r <- source chunk
  // (lane % 8) ^ (((r >> 1) & 3) << 1): the 32-byte swizzle the transposing reads undo.  Rows past M / columns past the row end are past
  // the buffer descriptor (zeros) or belong to output columns that are never stored.
  const bool a_panel = wave < 2;
  const int pcol0 = (a_panel ? n0 : k0) + (wave & 1) * 64;
  const long ld = a_panel ? p.lda : p.ldb;
  const int width = a_panel ? p.N : p.K;
  const uint16_t* base = a_panel ? p.a : p.b;
  const long span = ((long)p.M - 1) * ld * 2 + (long)width * 2;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(span < 0x7fffffffL ? span : 0x7fffffffL), 0x00020000);
  const int drow = lane >> 3, dchunk = lane & 7;
  uint32_t src_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = 8 * j + drow;
    const int chunk = dchunk ^ (((r >> 1) & 3) << 1);
    // a panel that starts past the operand's width (the last tile of N or K not a multiple of 128) reads nothing: offset past the end
    src_off[j] = pcol0 < width ? (uint32_t)((long)r * ld * 2 + (long)(pcol0 + chunk * 8) * 2) : 0x80000000u;
  }
  auto issue = [&](int t, int buf) {
    const uint32_t tile_off = (uint32_t)((long)(mt0 + t) * TBM * ld * 2);
    unsigned char* dst = lds + buf * STAGE + wave * PANEL;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t o;
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(o) : "s"(tile_off), "v"(src_off[j]));
      if (src_off[j] == 0x80000000u) o = 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, TN_LDS_AS3(dst + j * 1024), 16, o, 0, 0, 0);
    }
  };

  // fragment read offsets inside a panel (attn_fwd.hip: v_frag_offset)
  int foff[2];
  {
    const int i = l31 & 15;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
      foff[cb] = (4 * hh + (i >> 2)) * 128 + (((2 * cb + (l31 >> 4)) ^ ((2 * hh + (i >> 3)) & 3)) << 5) + 8 * (i & 3);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  if (nt > 0) issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    // tile t has landed (this wave's pieces: vmcnt(0); everyone's: the barrier, which also says every wave is done reading the other stage)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
    const unsigned char* st = lds + (t & 1) * STAGE;
    const unsigned char* pa = st + wr * PANEL;
    const unsigned char* pb = st + (2 + wc) * PANEL;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        fa[cb] = tn_frag(pa, foff[cb], kk);
        fb[cb] = tn_frag(pb, foff[cb], kk);
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[x], fb[y], acc[x][y], 0, 0, 0);
    }
  }

  // D[row = n (A rows), col = k (B columns)]: lane l31 = column, register r = row (r & 3) + 8 (r >> 2) + 4 hh of the 32 x 32 block
  float* const dst = p.part + (long)slab * p.N * p.K;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int kcol = k0 + wc * 64 + y * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nrow = n0 + wr * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (nrow < p.N && kcol < p.K) dst[(long)nrow * p.K + kcol] = acc[x][y][r];
      }
    }
}

// out = sum over slabs of part[s] in slab order, rounded once
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, uint16_t* __restrict__ out_bf16, float* __restrict__ out_f32,
                                                        long elems, int slabs) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < slabs; ++q) s += part[(long)q * elems + i];
    if (out_bf16) out_bf16[i] = f32_to_bf16(s);
    else out_f32[i] = s;
  }
}

int tn_slabs(long M, int N, int K) {
  // Slabs of the contraction range per output tile.  512 workgroups run at once (two per CU); a launch of 600 takes TWO rounds, the second
  // one at 17 % occupancy -- round 3 chose ceil(512 / tiles) and paid exactly that on the fine-tune step's dominant shapes (100 tiles x 6
  // slabs, 25 tiles x 21 slabs: half the kernel's time).  Pick the count that minimises rounds x (K-tiles per workgroup + its fixed
  // prologue / partial-tile store, ~6 K-tiles' worth) plus the second kernel's pass over the slabs.
  const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128);
  const long mt = (M + TBM - 1) / TBM;
  long best = 1;
  double best_cost = 1e30;
  for (long s = 1; s <= 64 && s <= mt; ++s) {
    const long rounds = (tiles * s + 511) / 512, per = (mt + s - 1) / s;
    const double cost = (double)rounds * (double)(per + 6) + 0.5 * (double)s;
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return (int)best;
}

}  // namespace

// bytes of fp32 workspace cd360_gemm_tn_bf16 needs for this shape (the per-slab partial tiles)
extern "C" int64_t cd360_gemm_tn_workspace_bytes(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)tn_slabs(M, N, K) * N * K * 4;
}

// out[N, K] = A[M, N]^T @ B[M, K] (fp32 accumulation).  A, B bf16 with row strides lda >= N, ldb >= K (elements, multiples of 8),
// 16-byte aligned; N, K multiples of 8.  out_dtype 0: out is fp32 [N, K]; 1: bf16 [N, K].  ws = cd360_gemm_tn_workspace_bytes(M, N, K)
// bytes.  Deterministic (fixed summation order).
extern "C" int cd360_gemm_tn_bf16(const void* a, const void* b, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldb, int out_dtype,
                                  void* ws, void* stream) {
  if (!a || !b || !out || !ws || M <= 0 || N <= 0 || K <= 0) return CD360_ERR_ARG;
  if (N % 8 || K % 8 || lda % 8 || ldb % 8 || lda < N || ldb < K || (out_dtype != 0 && out_dtype != 1)) return CD360_ERR_SHAPE;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | (uintptr_t)ws) % 16) return CD360_ERR_ARG;
  if (M > 0x7fffffffL || (M + 64) * lda * 2 >= (1L << 31) || (M + 64) * ldb * 2 >= (1L << 31)) return CD360_ERR_SHAPE;  // 32-bit buffer offsets
  TnParams p;
  p.a = (const uint16_t*)a; p.b = (const uint16_t*)b; p.part = (float*)ws;
  p.lda = lda; p.ldb = ldb; p.M = (int)M; p.N = N; p.K = K;
  p.tiles_n = (N + 127) / 128; p.tiles_k = (K + 127) / 128;
  p.slabs = tn_slabs(M, N, K);
  const int mt = (int)((M + TBM - 1) / TBM);
  p.mtiles_per_slab = (mt + p.slabs - 1) / p.slabs;
  p.slabs = (mt + p.mtiles_per_slab - 1) / p.mtiles_per_slab;  // no empty slab
  if (p.slabs == 1 && out_dtype == 0) p.part = (float*)out;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  if (attr != hipSuccess) return CD360_ERR_LAUNCH;
  const long nwg = (long)p.tiles_n * p.tiles_k * p.slabs;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)nwg), dim3(256), 2 * STAGE, (hipStream_t)stream, p);
  CD360_LAUNCH_CHECK();
  if (!(p.slabs == 1 && out_dtype == 0)) {
    const long elems = (long)N * K;
    const long blocks = (elems + 255) / 256;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, (const float*)ws,
                       out_dtype == 1 ? (uint16_t*)out : nullptr, out_dtype == 0 ? (float*)out : nullptr, elems, p.slabs);
    CD360_LAUNCH_CHECK();
  }
  return CD360_OK;
}
