// Flash attention forward, bf16 in / fp32 accumulate / bf16 out, head dim 64, no mask.
//
// Replaces the reference's only working attention operator,
//   xformers.ops.memory_efficient_attention(q, k, v, attn_bias=None)      (sgm/modules/attention.py:393-408)
// for all three call shapes of the pose path:
//   self-attn  Nq = Nk = hw,  text cross-attn Nk = 77,  pose-token cross-attn Nq = hw*S (up to 98 304), Nk = 77.
//
// CDNA4 design (one wave = 32 query rows, 4 waves per workgroup, 64-key tiles, 3 workgroups per CU):
//   * "swapped" scores  S^T = K Q^T  with v_mfma_f32_32x32x16_bf16, so every lane owns ONE query column:
//     the row max / row sum of the online softmax are in-register reductions plus one exchange with lane^32.
//   * O^T = V^T P^T: the P^T B-operand is taken straight from the lane's own S^T accumulator registers
//     (the contraction order over keys is permuted identically on the V^T side), so P never goes through LDS.
//   * K tile in LDS is XOR-swizzled at 16-B granularity (conflict-free ds_read_b128).  V is taken ROW-MAJOR ([.., key, d], a slice
//     of the same merged q|k|v projection GEMM as q and k), staged like K and read as the V^T A-operand with gfx950's transposing
//     LDS read ds_read_b64_tr_b16 (lane mapping measured with tools/probe/tr_read_probe.hip): inside a group of 16 lanes, lane s
//     supplies the address of 4 consecutive d of key s>>2 (d-chunk s&3) and lane i receives, for its d = i, the 4 keys.  A 32-B
//     XOR swizzle ((row>>1)&3) keeps the 512 bytes of one wave-instruction at exactly two accesses per bank.
//   * all tensors are addressed through explicit strides, so the kernel reads the projection outputs
//     [b, N, H*64] in place and writes [b, N, H*64] directly: no head split/merge copies.
#include "cd360_common.h"
#include "cd360_tuning.h"
#include <stdlib.h>

#define LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

struct AttnParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* o;
  float* lse;  // optional [B*H, Nq] fp32: log-sum-exp of the scaled scores (natural log), what cd360_attn_bwd_bf16 reads
  int B, H, Nq, Nk;
  long q_sb, q_sh, q_sn;  // element strides, d contiguous
  long k_sb, k_sh, k_sn;
  long v_sb, v_sh, v_sn;  // v[b][h][key][d], d contiguous (like k)
  long o_sb, o_sh, o_sn;
  float scale_log2e;
  int n_qtiles;
  int fast;  // 1: full tiles may use 32-bit buffer offsets (K and V of one head span < 2 GiB)
  float inv_sq, inv_sk, inv_sv, o_scale;  // fp8-MFMA variant: 1 / per-tensor e4m3 scales, and sv / 256 for the output
};

constexpr int BN = 64;        // keys per tile
constexpr int K_PITCH = 128;  // bytes per K row in LDS (64 d * 2 B), XOR-swizzled
constexpr int V_PITCH = 128;  // bytes per V row (key) in LDS, 32-B XOR-swizzled for the transposing read
constexpr int TILE_BYTES = BN * K_PITCH + BN * V_PITCH;

// LDS byte offset of 16-byte chunk `chunk` of V row `row` (32-B granules swapped by (row >> 1) & 3)
__device__ __forceinline__ int v_swz(int row, int chunk) { return row * V_PITCH + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }

// One K/V tile on its way from HBM to LDS: every thread carries 2 x 16 B of K and 2 x 16 B of V in registers, so the
// loads of tile t+1 are in flight while tile t is being consumed (split issue / write, LDS double-buffered).
struct TileRegs {
  u32x4 k[2], v[2];
};

__device__ __forceinline__ void tile_load(const AttnParams& p, const uint16_t* kp, const uint16_t* vp, int kt0, int tid, TileRegs& r) {
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {  // rows (keys) beyond Nk are zero in both tiles: their P is exactly 0 and 0 * 0 adds nothing
    const int row = (tid >> 3) + 32 * pass, chunk = tid & 7;
    u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
    const int key = kt0 + row;
    if (key < p.Nk) {
      kv = *reinterpret_cast<const u32x4*>(kp + (long)key * p.k_sn + chunk * 8);
      vv = *reinterpret_cast<const u32x4*>(vp + (long)key * p.v_sn + chunk * 8);
    }
    r.k[pass] = kv;
    r.v[pass] = vv;
  }
}

// Fast path of tile_load for a tile whose 64 keys all exist: buffer loads with per-thread byte offsets computed once and a
// scalar tile offset, so the K-loop spends no VALU on addresses or bounds.
__device__ __forceinline__ void tile_load_full(__amdgpu_buffer_rsrc_t krsrc, __amdgpu_buffer_rsrc_t vrsrc, int kgo, int vgo, int ksoff,
                                               int vsoff, int kpass, int vpass, TileRegs& r) {
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {  // the second pass (rows + 32) is a scalar offset as well
    r.k[pass] = __builtin_amdgcn_raw_buffer_load_b128(krsrc, kgo, ksoff + pass * kpass, 0);
    r.v[pass] = __builtin_amdgcn_raw_buffer_load_b128(vrsrc, vgo, vsoff + pass * vpass, 0);
  }
}

template <int BUF>
__device__ __forceinline__ void tile_store_to(unsigned char* lds, int kso, int vso, const TileRegs& r) {
  unsigned char* Ks = lds + BUF * TILE_BYTES;
  unsigned char* Vs = Ks + BN * K_PITCH;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {  // rows + 32: both swizzle terms ((row >> 1) & 7, & 3) are unchanged, so a constant offset
    *reinterpret_cast<u32x4*>(Ks + kso + pass * 32 * K_PITCH) = r.k[pass];
    *reinterpret_cast<u32x4*>(Vs + vso + pass * 32 * V_PITCH) = r.v[pass];
  }
}

typedef __attribute__((ext_vector_type(4))) short s16x4;

// per-lane byte offset (inside a V tile) of the transposing reads for the 32-channel block `db`; key group kk adds kk * 16 rows
__device__ __forceinline__ int v_frag_offset(int l31, int hh, int db) {
  const int i = l31 & 15;
  return (4 * hh + (i >> 2)) * V_PITCH + (((2 * db + (l31 >> 4)) ^ ((2 * hh + (i >> 3)) & 3)) << 5) + 8 * (i & 3);
}

// V^T A-fragment (rows = 32 channels of block db, k = 16 keys of group kk, in the key order of the P registers) from a row-major
// V tile: two ds_read_b64_tr_b16, keys 16kk + 4hh + {0..3} and 16kk + 8 + 4hh + {0..3}
__device__ __forceinline__ bf16x8 v_frag(const unsigned char* Vs, int voff_db, int kk) {
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Vs + voff_db + kk * 16 * V_PITCH));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Vs + voff_db + kk * 16 * V_PITCH + 8 * V_PITCH));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

struct WaveState {
  bf16x8 qf[4];    // this wave's 32 query rows as MFMA B fragments
  f32x16 oT[2];    // O^T accumulator: 64 d x 32 queries
  float m_run, l_run;
};

// One 64-key tile from LDS buffer BUF against the wave's 32 queries.  FULL = every key of the tile exists (no masking, both
// 32-key halves multiplied); the ragged last tile takes the other instantiation.  BUF is a template parameter so that every LDS
// address is `per-lane offset computed before the loop + immediate`.
template <int BUF, bool FULL>
__device__ __forceinline__ void consume_tile(const AttnParams& p, const unsigned char* lds, int kt0, const int (&koff)[4], const int (&voff)[2],
                                             int hh, float c, WaveState& w) {
  const unsigned char* Ks = lds + BUF * TILE_BYTES;
  const unsigned char* Vs = Ks + BN * K_PITCH;
  const int nkb = FULL ? 2 : ((p.Nk - kt0 > 32) ? 2 : 1);  // 32-key blocks with at least one valid key
  // ---- S^T = K Q^T ----
  f32x16 sT[2];
  __builtin_amdgcn_s_setprio(1);
  if (FULL) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + kb * 32 * K_PITCH + koff[ks]);
        sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w.qf[ks], ks == 0 ? zero : sT[kb], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) { sT[0][i] = 0.f; sT[1][i] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb < nkb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + kb * 32 * K_PITCH + koff[ks]);
          sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w.qf[ks], sT[kb], 0, 0, 0);
        }
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);

  // ---- online softmax (this lane: one query column; keys spread over registers and lane^32) ----
  if (!FULL) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= p.Nk) sT[kb][r] = -INFINITY;
      }
  }
  float mx = sT[0][0];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sT[kb][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float m_new = fmaxf(w.m_run, mx);  // raw (unscaled) score units
  const float alpha = __builtin_amdgcn_exp2f((w.m_run - m_new) * c);
  const float mc = -m_new * c;
  float rs = 0.f;
  uint32_t pk[16];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r], c, mc));
      const float p1 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r + 1], c, mc));
      rs += p0 + p1;
      pk[kb * 8 + (r >> 1)] = pack_bf16x2(p0, p1);
    }
  rs += __shfl_xor(rs, 32);
  w.l_run = w.l_run * alpha + rs;
  w.m_run = m_new;
  // (skipping this rescale when no running max moved -- `if (__any(m_new > m_old))` -- measured 3-7 % SLOWER: the branch
  // breaks the MFMA/VALU interleave; the 32 multiplies stay unconditional)
#pragma unroll
  for (int i = 0; i < 16; ++i) { w.oT[0][i] *= alpha; w.oT[1][i] *= alpha; }

  // ---- O^T += V^T P^T ----
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (FULL || kk < 2 * nkb) {
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        u32x4 pw = {pk[kk * 4 + 0], pk[kk * 4 + 1], pk[kk * 4 + 2], pk[kk * 4 + 3]};
        w.oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_frag(Vs, voff[db], kk), __builtin_bit_cast(bf16x8, pw), w.oT[db], 0, 0, 0);
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

// 4 waves x 32 queries per workgroup, 3 workgroups per CU.  (A 2-q-block-per-wave variant -- half the LDS fragment traffic, one
// wave per SIMD -- measured 15 % slower at every SDXL shape and was removed.)
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = tile / p.n_qtiles, qt = tile - bh * p.n_qtiles;
  const int b = bh / p.H, h = bh - b * p.H;
  const uint16_t* qp = p.q + b * p.q_sb + h * p.q_sh;
  const uint16_t* kp = p.k + b * p.k_sb + h * p.k_sh;
  const uint16_t* vp = p.v + b * p.v_sb + h * p.v_sh;
  uint16_t* op = p.o + b * p.o_sb + h * p.o_sh;

  WaveState w;
  const int qrow = qt * 128 + wave * 32 + l31;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (qrow < p.Nq) v = *reinterpret_cast<const u32x4*>(qp + (long)qrow * p.q_sn + 16 * ks + 8 * hh);
    w.qf[ks] = __builtin_bit_cast(bf16x8, v);
  }
  w.m_run = -INFINITY;
  w.l_run = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { w.oT[0][i] = 0.f; w.oT[1][i] = 0.f; }
  const float c = p.scale_log2e;

  // per-lane LDS offsets, constant over the K-loop
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * K_PITCH + (((2 * ks + hh) ^ ((l31 >> 1) & 7)) << 4);
  const int voff[2] = {v_frag_offset(l31, hh, 0), v_frag_offset(l31, hh, 1)};
  const int srow = tid >> 3, schunk = tid & 7;  // staging: thread -> (row, 16-byte chunk), rows + 32 on the second pass
  const int kso = srow * K_PITCH + ((schunk ^ ((srow >> 1) & 7)) << 4), vso = v_swz(srow, schunk);
  const int kgo = (int)(srow * p.k_sn * 2) + schunk * 16, vgo = (int)(srow * p.v_sn * 2) + schunk * 16;
  const int kpass = (int)(32 * p.k_sn * 2), vpass = (int)(32 * p.v_sn * 2);
  const int n_tiles = (p.Nk + BN - 1) / BN;
  const int n_full = p.fast ? p.Nk / BN : 0;  // tiles that take the unguarded buffer-load / unmasked path
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, p.fast ? (int)((long)p.Nk * p.k_sn * 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, p.fast ? (int)((long)p.Nk * p.v_sn * 2) : 0, 0x00020000);
  const int kstep = (int)(BN * p.k_sn * 2), vstep = (int)(BN * p.v_sn * 2);

  TileRegs tr;
  tile_load(p, kp, vp, 0, tid, tr);
  tile_store_to<0>(lds, kso, vso, tr);
  __syncthreads();

  // the loads of tile t+1 are in flight during tile t's MFMAs and land in the other LDS buffer (last read before the previous
  // barrier); the loop is unrolled over the two buffers so BUF is a compile-time constant
#define CD360_ATTN_STEP(BUF, T)                                                                         \
  {                                                                                                     \
    const int t_ = (T), kt0 = t_ * BN;                                                                  \
    const bool more = t_ + 1 < n_tiles;                                                                 \
    if (more) {                                                                                         \
      if (t_ + 1 < n_full) tile_load_full(krsrc, vrsrc, kgo, vgo, (t_ + 1) * kstep, (t_ + 1) * vstep, kpass, vpass, tr); \
      else tile_load(p, kp, vp, kt0 + BN, tid, tr);                                                     \
    }                                                                                                   \
    if (t_ < n_full) consume_tile<BUF, true>(p, lds, kt0, koff, voff, hh, c, w);                        \
    else consume_tile<BUF, false>(p, lds, kt0, koff, voff, hh, c, w);                                   \
    if (more) tile_store_to<(BUF) ^ 1>(lds, kso, vso, tr);                                              \
    __syncthreads();                                                                                    \
  }
  for (int t = 0; t < n_tiles; t += 2) {
    CD360_ATTN_STEP(0, t)
    if (t + 1 >= n_tiles) break;
    CD360_ATTN_STEP(1, t + 1)
  }
#undef CD360_ATTN_STEP

  // (routing these stores through LDS as full 128-byte rows, as attn_smallk_kernel does, measured no gain here -- the epilogue is
  // a percent of a 16-64 tile K-loop -- and cost a register spill)
  if (qrow < p.Nq) {
    const float inv = 1.f / w.l_run;
    if (p.lse && hh == 0) p.lse[(long)bh * p.Nq + qrow] = (w.m_run * c + __log2f(w.l_run)) * 0.6931471805599453f;
    uint16_t* orow = op + (long)qrow * p.o_sn;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hh;
        u32x2 wv = {pack_bf16x2(w.oT[db][4 * g + 0] * inv, w.oT[db][4 * g + 1] * inv),
                    pack_bf16x2(w.oT[db][4 * g + 2] * inv, w.oT[db][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2*>(orow + d) = wv;
      }
    }
  }
}

// ---- self-attention, second generation (Nq % (128 QB) == 0, Nk % 64 == 0) -----------------------------------------------------------
// At head dim 64 the softmax, not the MFMA, is the long pole of the tile above: per 32 x 32 block of scores 8 MFMAs (256 cycles of the
// matrix pipe) stand against max + scale + exp + sum + pack + rescale-O = ~6 VALU instructions per score (~350 issue cycles).  This
// kernel removes three of them and takes the staging work off the VALU:
//   * q is scaled by (scale log2 e) once, when its fragments are built: the scores leave the MFMA in exp2 units;
//   * the running maximum is LAZY: the accumulator of S^T = K Q^T starts at -m_ref (a 16-register tuple per query block that is
//     only rewritten when m_ref moves), so p = exp2(S') needs no subtract; m_ref moves -- and O, l and the scores at hand are rescaled,
//     in a wave-uniform rarely taken branch -- only when a tile's maximum exceeds it by more than 2^8 (p <= 256: harmless in bf16 /
//     fp32, the final O / l is independent of m_ref).  The per-tile O *= alpha is gone;
//   * max by v_max3_f32, the row sum stays per lane until the end (no per-tile exchange with lane ^ 32);
//   * K / V tiles travel L2 -> LDS by LDS-DMA into a 4-deep ring (two tiles in flight, no staging registers, no LDS-write pass);
//     the XOR swizzles of the two tiles are applied on the per-lane SOURCE address (the DMA destination is lane-linear);
//   * software pipeline inside the wave: the S^T MFMAs of tile t+1 are issued before the softmax of tile t, whose VALU work
//     runs under them, then P V of tile t.  One workgroup barrier per tile.
// QB = query blocks (32 rows) per wave: 2 halves the LDS fragment traffic per MFMA (K and V^T fragments serve both blocks) and is
// used when the launch still fills the chip (the 64^2 level), 1 otherwise.
// NW = waves per workgroup (4 | 8): the K / V tile and its DMA pieces are shared by NW * 32 * QB queries.
template <int QB, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void attn_self_kernel(AttnParams p) {
  constexpr int NB = 4;           // ring slots: tiles t, t+1 in use, t+2 landing, t+3 being issued
  constexpr bool PIPE = QB == 1;  // two query blocks leave no registers for the look-ahead scores: the second wave of the SIMD covers
  constexpr int PPT = 8 / NW;     // DMA pieces per thread, tile and operand (a piece of the workgroup = NW * 8 rows)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NB * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = tile / p.n_qtiles, qt = tile - bh * p.n_qtiles;
  const int b = bh / p.H, h = bh - b * p.H;
  const uint16_t* qp = p.q + b * p.q_sb + h * p.q_sh;
  const uint16_t* kp = p.k + b * p.k_sb + h * p.k_sh;
  const uint16_t* vp = p.v + b * p.v_sb + h * p.v_sh;
  uint16_t* op = p.o + b * p.o_sb + h * p.o_sh;
  const float c = p.scale_log2e;

  // ---- Q fragments, pre-scaled ----
  bf16x8 qf[QB][4];
  const int qrow0 = qt * (NW * 32 * QB) + wave * (32 * QB) + l31;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(qp + (long)(qrow0 + 32 * qb) * p.q_sn + 16 * ks + 8 * hh);
      u32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(bf16lo_to_f32(v[i]) * c, bf16hi_to_f32(v[i]) * c);
      qf[qb][ks] = __builtin_bit_cast(bf16x8, o);
    }

  // ---- LDS-DMA geometry: piece j of a tile covers rows 32 j + tid / 8; LDS chunk tid % 8 <- source chunk ^ swizzle(row) ----
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (int)((long)p.Nk * p.k_sn * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (int)((long)p.Nk * p.v_sn * 2), 0x00020000);
  const int srow = tid >> 3, sch = tid & 7;
  const uint32_t kgo = (uint32_t)(srow * p.k_sn * 2) + (uint32_t)((sch ^ ((srow >> 1) & 7)) << 4);
  const uint32_t vgo = (uint32_t)(srow * p.v_sn * 2) + (uint32_t)((sch ^ (((srow >> 1) & 3) << 1)) << 4);
  const uint32_t kpass = (uint32_t)(NW * 8 * p.k_sn * 2), vpass = (uint32_t)(NW * 8 * p.v_sn * 2);
  const uint32_t kstep = (uint32_t)(BN * p.k_sn * 2), vstep = (uint32_t)(BN * p.v_sn * 2);
  unsigned char* const dma_base = lds + wave * 1024;
  auto issue = [&](int t) {  // K rows then V rows of tile t into ring slot t % NB, PPT pieces each
    unsigned char* slot = dma_base + (t % NB) * TILE_BYTES;
#pragma unroll
    for (int j = 0; j < PPT; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(krsrc, LDS_AS3(slot + j * (NW * 1024)), 16, kgo, t * kstep + j * kpass, 0, 0);
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, LDS_AS3(slot + BN * K_PITCH + j * (NW * 1024)), 16, vgo, t * vstep + j * vpass, 0, 0);
  };
  auto wait_tiles_in_flight = [&](int later) {  // all but the `later` most recent tiles have landed
    if (later <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPT) : "memory");
  };

  // per-lane fragment offsets inside a tile
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * K_PITCH + (((2 * ks + hh) ^ ((l31 >> 1) & 7)) << 4);
  const int voff[2] = {BN * K_PITCH + v_frag_offset(l31, hh, 0), BN * K_PITCH + v_frag_offset(l31, hh, 1)};

  f32x16 negm[QB], oT[QB][2], sc[QB][2], sn[PIPE ? QB : 1][2];
  float l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    l_run[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { negm[qb][i] = 0.f; oT[qb][0][i] = 0.f; oT[qb][1][i] = 0.f; }
  }

  auto qk = [&](const unsigned char* T, auto& s) {  // S'^T = K Q^T - m_ref for one tile
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(T + kb * 32 * K_PITCH + koff[ks]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[qb][ks], ks == 0 ? negm[qb] : s[qb][kb], 0, 0, 0);
      }
  };

  const int n_tiles = p.Nk / BN;
  // tile maximum of one query block (this lane's column; both key halves through lane ^ 32)
  auto tile_max = [&](const f32x16 (&s)[2]) {
    float m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(s[j >> 1][8 * (j & 1)], s[j >> 1][8 * (j & 1) + 1]), s[j >> 1][8 * (j & 1) + 2]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = fmaxf(fmaxf(m[j], s[j >> 1][8 * (j & 1) + 3]), s[j >> 1][8 * (j & 1) + 4]);
      m[j] = fmaxf(fmaxf(m[j], s[j >> 1][8 * (j & 1) + 5]), s[j >> 1][8 * (j & 1) + 6]);
    }
    const float ma = fmaxf(fmaxf(m[0], m[1]), s[0][7]), mb = fmaxf(fmaxf(m[2], m[3]), s[0][15]);
    const float mx = fmaxf(fmaxf(ma, mb), fmaxf(s[1][7], s[1][15]));
    // both key halves: v_permlane32_swap (VALU) instead of a ds_bpermute round trip on the way to the end-of-tile branch
    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, mx), false, false);
    return fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
  };
  // the rare path: move m_ref of query block qb by d (>= 0 except on the first tile) and rescale what hangs on it
  auto move_ref = [&](int qb, float d, f32x16 (&s)[2]) {
    // d < 0 happens on the first tile only (m_ref starts at 0, l and O at 0): nothing to rescale yet, and exp2(-d) of a first tile whose
    // scores all sit below -128 would be +inf (0 * inf = NaN in l and O) -- clamp the factor's exponent at 0
    const float alpha = __builtin_amdgcn_exp2f(-fmaxf(d, 0.f));
    l_run[qb] *= alpha;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      negm[qb][i] -= d;
      s[0][i] -= d; s[1][i] -= d;
      oT[qb][0][i] *= alpha; oT[qb][1][i] *= alpha;
    }
  };
  // p = exp2(S') of tile scores s -> packed bf16 P (MFMA B operand order), row sum into l_run
  auto softmax_p = [&](int qb, const f32x16 (&s)[2], uint32_t (&pk)[16]) {
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(s[kb][r]);
        const float p1 = __builtin_amdgcn_exp2f(s[kb][r + 1]);
        r0 += p0; r1 += p1;
        pk[kb * 8 + (r >> 1)] = pack_bf16x2(p0, p1);
      }
    l_run[qb] += r0 + r1;
  };
  auto pv = [&](const unsigned char* T, const uint32_t (&pk)[QB][16]) {  // O^T += V^T P^T
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const bf16x8 vfr = v_frag(T, voff[db], kk);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const u32x4 pw = {pk[qb][kk * 4 + 0], pk[qb][kk * 4 + 1], pk[qb][kk * 4 + 2], pk[qb][kk * 4 + 3]};
          oT[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, __builtin_bit_cast(bf16x8, pw), oT[qb][db], 0, 0, 0);
        }
      }
  };
  auto rendezvous = [&](int t, int need) {  // tile `need` landed and visible; ring slot (t + NB - 1) % NB = (t - 1) % NB consumed by all waves
    if (need < n_tiles) {
      const int last = n_tiles - 1 < t + NB - 2 ? n_tiles - 1 : t + NB - 2;  // newest tile issued so far
      wait_tiles_in_flight(last - need);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t + NB - 1 < n_tiles) issue(t + NB - 1);
  };

#pragma unroll
  for (int t = 0; t < NB - 1; ++t)
    if (t < n_tiles) issue(t);
  if (PIPE) {
    // prologue: scores of tile 0, m_ref = their maximum
    wait_tiles_in_flight((n_tiles < NB - 1 ? n_tiles : NB - 1) - 1);
    __syncthreads();
    qk(lds, sc);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) move_ref(qb, tile_max(sc[qb]), sc[qb]);
    // steady state, ONE basic block per tile: S' MFMAs of tile t+1 | exp / sum / pack of tile t | P V MFMAs of tile t | max of t+1
    for (int t = 0; t + 1 < n_tiles; ++t) {
      rendezvous(t, t + 1);
      uint32_t pk[QB][16];
      float mx[QB];
      qk(lds + ((t + 1) % NB) * TILE_BYTES, sn);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) softmax_p(qb, sc[qb], pk[qb]);
      pv(lds + (t % NB) * TILE_BYTES, pk);
      bool any = false;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        mx[qb] = tile_max(sn[qb]);
        any = any || mx[qb] > 8.f;
      }
      if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) move_ref(qb, mx[qb] > 8.f ? mx[qb] : 0.f, sn[qb]);
      }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) { sc[qb][0] = sn[qb][0]; sc[qb][1] = sn[qb][1]; }
    }
    {  // last tile: no look-ahead
      const int t = n_tiles - 1;
      rendezvous(t, n_tiles);
      uint32_t pk[QB][16];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) softmax_p(qb, sc[qb], pk[qb]);
      pv(lds + (t % NB) * TILE_BYTES, pk);
    }
  } else {
    for (int t = 0; t < n_tiles; ++t) {
      rendezvous(t, t);
      const unsigned char* Tc = lds + (t % NB) * TILE_BYTES;
      uint32_t pk[QB][16];
      qk(Tc, sc);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float mx = tile_max(sc[qb]);
        if (__builtin_amdgcn_ballot_w64(t == 0 || mx > 8.f) != 0) move_ref(qb, (t == 0 || mx > 8.f) ? mx : 0.f, sc[qb]);
        softmax_p(qb, sc[qb], pk[qb]);
      }
      pv(Tc, pk);
    }
  }

  // O^T registers (lane = query, four consecutive channels per register group) -> a wave-private 4-KB block in a ring slot nobody reads
  // any more (the last tile sits in slot (n_tiles - 1) % NB; waves 0-3 take the next slot, waves 4-7 the one after) -> FULL 128-byte rows,
  // one dwordx4 per lane: the natural layout stores 8-byte pieces of 32 different rows per instruction, 16 instructions per query block
  // (store-issue-bound tail of a 30 us kernel); through the LDS it is 4.
  unsigned char* const Os = lds + ((n_tiles + (wave >> 2)) % NB) * TILE_BYTES + (wave & 3) * 4096;
  const int lrow = lane >> 3, lchunk = lane & 7;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l = l_run[qb] + __shfl_xor(l_run[qb], 32);
    const float inv = __builtin_amdgcn_rcpf(l);
    const int qrow = qrow0 + 32 * qb;
    if (p.lse && hh == 0) p.lse[(long)bh * p.Nq + qrow] = (__log2f(l) - negm[qb][0]) * 0.6931471805599453f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dbyte = (db * 32 + 8 * g + 4 * hh) * 2;
        const u32x2 wv = {pack_bf16x2(oT[qb][db][4 * g + 0] * inv, oT[qb][db][4 * g + 1] * inv),
                          pack_bf16x2(oT[qb][db][4 * g + 2] * inv, oT[qb][db][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2*>(Os + l31 * 128 + (((dbyte >> 4) ^ ((l31 >> 1) & 7)) << 4) + (dbyte & 8)) = wv;
      }
    uint16_t* obase = op + (long)(qrow0 - l31 + 32 * qb) * p.o_sn;  // first row of this wave's query block
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + lrow;
      const u32x4 v = *reinterpret_cast<const u32x4*>(Os + row * 128 + ((lchunk ^ ((row >> 1) & 7)) << 4));
      *reinterpret_cast<u32x4*>(obase + (long)row * p.o_sn + lchunk * 8) = v;
    }
  }
}

// ---- Nk <= 96 (the text context: 77 keys) -------------------------------------------------------------------------------------
// Cross-attention over the text tokens (A2) and over the pose tokens (A3, up to 98 304 queries per head against the same 77 keys)
// is a streaming problem: 256 B of Q/O traffic per (query, head) against 2 x 12 MFMAs.  The whole K and V^T of a head fit in
// REGISTERS as MFMA A-fragments (NKB*4 + NKB*4 fragments of 4 VGPRs), so after one staging pass through LDS the loop over query
// blocks has no LDS traffic, no barrier and no online-softmax state: S^T for all keys at once, one max / exp2 / sum, P straight
// from the accumulator registers into the PV MFMAs, the next block's Q rows already in flight.  Keys >= Nk are masked for free:
// the accumulator of the last 32-key block starts at -1e30 in the masked rows instead of 0.
//
// FP8 = true is the fp8-MFMA variant (BASELINE config 5): the same bf16 tensors in and out, but Q, K, V^T and P are rounded to
// OCP e4m3 in registers (per-tensor scales amax / 448 supplied by the caller, P scaled by 256 into the e4m3 normal range) and both
// contractions run on v_mfma_f32_32x32x16_fp8_fp8.  It exists to measure what fp8 QK^T / PV costs in accuracy; HBM traffic is
// unchanged (bf16 rows), so it is not faster on this HBM-bound shape.
__device__ __forceinline__ long bf16x8_to_fp8x8(bf16x8 v, float inv_scale) {
  const u32x4 w = __builtin_bit_cast(u32x4, v);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo_to_f32(w[0]) * inv_scale, bf16hi_to_f32(w[0]) * inv_scale, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo_to_f32(w[1]) * inv_scale, bf16hi_to_f32(w[1]) * inv_scale, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo_to_f32(w[2]) * inv_scale, bf16hi_to_f32(w[2]) * inv_scale, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo_to_f32(w[3]) * inv_scale, bf16hi_to_f32(w[3]) * inv_scale, hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int NKB, bool FP8>
__global__ __launch_bounds__(256, 2) void attn_smallk_kernel(AttnParams p) {
  constexpr int NKEYS = NKB * 32;
  // K / V staging (read once into registers) + per wave one 32 x 128 B block each for Q (in) and O (out): rows move between
  // HBM and LDS as FULL 128-byte lines (8 lanes x 16 B per row) and are re-read in MFMA fragment shape from LDS -- loading the
  // fragments straight from global memory touches every line four times with 32-byte pieces and capped the kernel at 2.5 TB/s.
  __shared__ __attribute__((aligned(16))) unsigned char lds[NKEYS * K_PITCH + NKEYS * V_PITCH + 4 * 2 * 32 * K_PITCH];
  unsigned char* Ks = lds;
  unsigned char* Vs = lds + NKEYS * K_PITCH;
  unsigned char* Qs = lds + NKEYS * K_PITCH + NKEYS * V_PITCH + (threadIdx.x >> 6) * (2 * 32 * K_PITCH);  // this wave's Q block
  unsigned char* Os = Qs + 32 * K_PITCH;                                                          // ... and O block

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.x / p.n_qtiles, chunk_id = blockIdx.x - bh * p.n_qtiles;  // n_qtiles = query chunks per head here
  const int b = bh / p.H, h = bh - b * p.H;
  const uint16_t* qp = p.q + b * p.q_sb + h * p.q_sh;
  const uint16_t* kp = p.k + b * p.k_sb + h * p.k_sh;
  const uint16_t* vp = p.v + b * p.v_sb + h * p.v_sh;
  uint16_t* op = p.o + b * p.o_sb + h * p.o_sh;

  // ---- stage K and V [Nk, 64] (rows >= Nk zero), each in its own swizzle ----
  for (int i = tid; i < NKEYS * 8; i += 256) {
    const int row = i >> 3, chunk = i & 7;
    u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
    if (row < p.Nk) {
      kv = *reinterpret_cast<const u32x4*>(kp + (long)row * p.k_sn + chunk * 8);
      vv = *reinterpret_cast<const u32x4*>(vp + (long)row * p.v_sn + chunk * 8);
    }
    *reinterpret_cast<u32x4*>(Ks + row * K_PITCH + ((chunk ^ ((row >> 1) & 7)) << 4)) = kv;
    *reinterpret_cast<u32x4*>(Vs + v_swz(row, chunk)) = vv;
  }
  __syncthreads();

  // ---- K and V^T fragments into registers, mask vector for the last key block ----
  bf16x8 kf[NKB][4], vf[2][2 * NKB];
  long kf8[NKB][4], vf8[2][2 * NKB];  // the e4m3 copies (FP8 only; dead otherwise)
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int krow = kb * 32 + l31;
      kf[kb][ks] = *reinterpret_cast<const bf16x8*>(Ks + krow * K_PITCH + (((2 * ks + hh) ^ ((krow >> 1) & 7)) << 4));
      if (FP8) kf8[kb][ks] = bf16x8_to_fp8x8(kf[kb][ks], p.inv_sk);
    }
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int kk = 0; kk < 2 * NKB; ++kk) {
      vf[db][kk] = v_frag(Vs, v_frag_offset(l31, hh, db), kk);
      if (FP8) vf8[db][kk] = bf16x8_to_fp8x8(vf[db][kk], p.inv_sv);
    }
  f32x16 init_last;  // accumulator start of the last key block: 0 for real keys, -1e30 for padding
#pragma unroll
  for (int r = 0; r < 16; ++r) init_last[r] = ((NKB - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh < p.Nk) ? 0.f : -1e30f;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float c = p.scale_log2e;

  // ---- this wave's 32-query blocks: first, first + 4, ... inside the workgroup's chunk ----
  const int nqb = (p.Nq + 31) >> 5;
  const int per = (nqb + p.n_qtiles - 1) / p.n_qtiles;  // blocks per chunk
  const int qb_end = min(nqb, (chunk_id + 1) * per);
  int qb = chunk_id * per + wave;
  // line-shaped access: pass i moves rows 8 i + lane / 8, 16-byte chunk lane % 8
  const int lrow = lane >> 3, lchunk = lane & 7;
  auto load_q = [&](int blk, u32x4 (&dst)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qrow = blk * 32 + 8 * i + lrow;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (blk < qb_end && qrow < p.Nq) v = *reinterpret_cast<const u32x4*>(qp + (long)qrow * p.q_sn + lchunk * 8);
      dst[i] = v;
    }
  };
  u32x4 qn[4];
  load_q(qb, qn);
  for (; qb < qb_end; qb += 4) {
    // rows -> this wave's LDS block (16-B XOR swizzle) -> B fragments (row l31, chunk 2 ks + hh); wave-private, no barrier
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + lrow;
      *reinterpret_cast<u32x4*>(Qs + row * K_PITCH + ((lchunk ^ ((row >> 1) & 7)) << 4)) = qn[i];
    }
    load_q(qb + 4, qn);  // next block's rows travel during this block's MFMAs
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[ks] = *reinterpret_cast<const bf16x8*>(Qs + l31 * K_PITCH + (((2 * ks + hh) ^ ((l31 >> 1) & 7)) << 4));

    f32x16 sT[NKB];
    long qf8[4];
    if (FP8) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf8[ks] = bf16x8_to_fp8x8(qf[ks], p.inv_sq);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x16 acc0 = ks == 0 ? (kb == NKB - 1 ? init_last : zero) : sT[kb];
        sT[kb] = FP8 ? __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf8[kb][ks], qf8[ks], acc0, 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks], qf[ks], acc0, 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);

    float mx = sT[0][0];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sT[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mc = -mx * c;
    float rs = 0.f;
    uint32_t pk[NKB * 8];   // bf16 pairs
    int pk8[NKB * 4];       // e4m3 quads of 256 p (FP8)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r], c, mc));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r + 1], c, mc));
        const float p2 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r + 2], c, mc));
        const float p3 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r + 3], c, mc));
        rs += (p0 + p1) + (p2 + p3);
        if (FP8) {
          int d8 = __builtin_amdgcn_cvt_pk_fp8_f32(p0 * 256.f, p1 * 256.f, 0, false);
          pk8[kb * 4 + (r >> 2)] = __builtin_amdgcn_cvt_pk_fp8_f32(p2 * 256.f, p3 * 256.f, d8, true);
        } else {
          pk[kb * 8 + (r >> 1)] = pack_bf16x2(p0, p1);
          pk[kb * 8 + (r >> 1) + 1] = pack_bf16x2(p2, p3);
        }
      }
    rs += __shfl_xor(rs, 32);

    f32x16 oT[2];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2 * NKB; ++kk) {
      u32x4 pw = {0u, 0u, 0u, 0u};
      long pw8 = 0;
      if (FP8) pw8 = (long)(((unsigned long)(unsigned)pk8[kk * 2 + 1] << 32) | (unsigned)pk8[kk * 2]);
      else pw = u32x4{pk[kk * 4 + 0], pk[kk * 4 + 1], pk[kk * 4 + 2], pk[kk * 4 + 3]};
#pragma unroll
      for (int db = 0; db < 2; ++db)
        oT[db] = FP8 ? __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf8[db][kk], pw8, kk == 0 ? zero : oT[db], 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][kk], __builtin_bit_cast(bf16x8, pw), kk == 0 ? zero : oT[db], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);

    // O^T registers (lane = query, 4 consecutive d per group) -> this wave's LDS block -> full 128-byte rows to HBM
    {
      const float inv = (FP8 ? p.o_scale : 1.f) / rs;
      if (!FP8 && p.lse && hh == 0 && qb * 32 + l31 < p.Nq) p.lse[(long)bh * p.Nq + qb * 32 + l31] = (mx * c + __log2f(rs)) * 0.6931471805599453f;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dbyte = (db * 32 + 8 * g + 4 * hh) * 2;  // byte offset in the row: chunk = dbyte >> 4, half = dbyte & 8
          u32x2 wv = {pack_bf16x2(oT[db][4 * g + 0] * inv, oT[db][4 * g + 1] * inv),
                      pack_bf16x2(oT[db][4 * g + 2] * inv, oT[db][4 * g + 3] * inv)};
          *reinterpret_cast<u32x2*>(Os + l31 * K_PITCH + ((((dbyte >> 4)) ^ ((l31 >> 1) & 7)) << 4) + (dbyte & 8)) = wv;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + lrow, qrow = qb * 32 + row;
        const u32x4 v = *reinterpret_cast<const u32x4*>(Os + row * K_PITCH + ((lchunk ^ ((row >> 1) & 7)) << 4));
        if (qrow < p.Nq) *reinterpret_cast<u32x4*>(op + (long)qrow * p.o_sn + lchunk * 8) = v;
      }
    }
  }
}

}  // namespace

// fp8_amax = {max|q|, max|k|, max|v|} selects the fp8-MFMA variant (Nk <= 96 only); NULL = bf16 MFMA
static int attn_launch(const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk, const int64_t* q_strides,
                       const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, float scale, const float* fp8_amax,
                       float* lse, void* stream, bool prescaled = false) {
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return CD360_ERR_ARG;
  AttnParams p;
  p.q = (const uint16_t*)q; p.k = (const uint16_t*)k; p.v = (const uint16_t*)v; p.o = (uint16_t*)o; p.lse = lse;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.q_sb = q_strides[0]; p.q_sh = q_strides[1]; p.q_sn = q_strides[2];
  p.k_sb = k_strides[0]; p.k_sh = k_strides[1]; p.k_sn = k_strides[2];
  p.v_sb = v_strides[0]; p.v_sh = v_strides[1]; p.v_sn = v_strides[2];
  p.o_sb = o_strides[0]; p.o_sh = o_strides[1]; p.o_sn = o_strides[2];
  // 16-byte vector access requirements
  const int64_t all[] = {p.q_sb, p.q_sh, p.q_sn, p.k_sb, p.k_sh, p.k_sn, p.v_sb, p.v_sh, p.v_sn};
  for (int64_t s : all) if (s % 8) return CD360_ERR_SHAPE;
  if (p.o_sb % 4 || p.o_sh % 4 || p.o_sn % 4) return CD360_ERR_SHAPE;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 || (uintptr_t)o % 8) return CD360_ERR_ARG;
  p.scale_log2e = prescaled ? 1.f : scale * 1.4426950408889634f;  // prescaled: q already carries scale * log2(e)
  // (its 16-byte output stores need 16-byte aligned output rows; otherwise the tiled kernel with 8-byte stores serves the call)
  bool smallk = Nk <= 96 && !(p.o_sb % 8 || p.o_sh % 8 || p.o_sn % 8 || (uintptr_t)o % 16);
  if (fp8_amax) {
    if (!smallk) return CD360_ERR_SHAPE;
    for (int i = 0; i < 3; ++i) if (!(fp8_amax[i] > 0.f)) return CD360_ERR_ARG;
    const float sq = fp8_amax[0] / 448.f, sk = fp8_amax[1] / 448.f, sv = fp8_amax[2] / 448.f;  // OCP e4m3: largest finite value 448
    p.inv_sq = 1.f / sq; p.inv_sk = 1.f / sk; p.inv_sv = 1.f / sv;
    p.scale_log2e *= sq * sk;     // scores come out in units of sq * sk
    p.o_scale = sv / 256.f;       // V in units of sv, P in units of 1 / 256
  } else if (cd360_tune().attn_smallk == 0) {
    smallk = false;  // tuning override: always the tiled kernel
  }
  if (smallk) {
    // one workgroup = one (batch, head) x one chunk of 32-query blocks; ~2 workgroups per CU in total, at least one block per wave
    const int nqb = (Nq + 31) / 32;
    long target = 512;  // two resident workgroups per CU (swept 240 ... 8192: 480-512 best for the pose-token shapes, text shapes flat)
    if (cd360_tune().attn_smallk_wgs > 0) target = cd360_tune().attn_smallk_wgs;  // tuning override
    long chunks = (target + (long)B * H - 1) / ((long)B * H);
    const long max_chunks = (nqb + 3) / 4;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    p.n_qtiles = (int)chunks;
    p.fast = 0;
    const long nwg = chunks * B * H;
    if (nwg > 0x7fffffffL) return CD360_ERR_SHAPE;
#define CD360_SMALLK(NKB_) \
  do { \
    if (fp8_amax) hipLaunchKernelGGL((attn_smallk_kernel<NKB_, true>), dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p); \
    else hipLaunchKernelGGL((attn_smallk_kernel<NKB_, false>), dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p); \
  } while (0)
    if (Nk <= 32) CD360_SMALLK(1);
    else if (Nk <= 64) CD360_SMALLK(2);
    else CD360_SMALLK(3);
#undef CD360_SMALLK
    CD360_LAUNCH_CHECK();
    return CD360_OK;
  }
  // Second-generation self-attention kernel (whole tiles).  Its q pre-scaling costs one more bf16 rounding of q unless the caller
  // folded the scale into the projection (cd360_attn_fwd_prescaled_bf16: exact), so plain calls keep the first generation unless
  // cd360_tuning.attn_self says otherwise: 0 = first generation, 1 = 4 waves x 32 queries, 2 = 8 waves x 64 queries, unset = by shape.
  int gen2 = (Nk % 64 == 0 && Nq % 128 == 0 && (long)Nk * p.k_sn * 2 < (1L << 31) && (long)Nk * p.v_sn * 2 < (1L << 31) &&
              !(p.o_sb % 8 || p.o_sh % 8 || p.o_sn % 8 || (uintptr_t)o % 16))  // (its output leaves as 16-byte pieces of full rows)
                 ? 1 : 0;
  int pick = prescaled ? -1 : 0;
  if (cd360_tune().attn_self >= 0) pick = cd360_tune().attn_self;
  if (gen2 && pick) {
    bool wide = Nq % 512 == 0 && (long)B * H * (Nq / 512) >= 240;  // enough 512-query workgroups for every CU
    if (pick > 0) wide = pick == 2 && Nq % 512 == 0;
    // 3 (A/B, round 5): eight waves x 32 queries = 256-query workgroups with the look-ahead scores -- one workgroup per CU at the 32^2 level
    // (240 of them) instead of two of 128 queries, i.e. one K / V stream per CU instead of two.  Measured (tools/bench_attn_self.py, same box,
    // us, variant 1 / 3): 32^2 level 25.6-28.8 / 27.7-28.5, 64^2 level 165.8-168.0 / 169.3-170.6, bit-identical -- no gain: the K / V stream is
    // not what this kernel waits for.  By cd360_tuning.attn_self = 3 only.
    const bool mid = pick == 3 && Nq % 256 == 0;
    p.n_qtiles = Nq / (mid ? 256 : wide ? 512 : 128);
    const long nwg2 = (long)p.n_qtiles * B * H;
    if (nwg2 > 0x7fffffffL) return CD360_ERR_SHAPE;
    if (mid) hipLaunchKernelGGL((attn_self_kernel<1, 8>), dim3((unsigned)nwg2), dim3(512), 0, (hipStream_t)stream, p);
    else if (wide) hipLaunchKernelGGL((attn_self_kernel<2, 8>), dim3((unsigned)nwg2), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_self_kernel<1, 4>), dim3((unsigned)nwg2), dim3(256), 0, (hipStream_t)stream, p);
    CD360_LAUNCH_CHECK();
    return CD360_OK;
  }
  p.n_qtiles = (Nq + 127) / 128;
  p.fast = ((long)Nk * p.k_sn * 2 < (1L << 31) && (long)Nk * p.v_sn * 2 < (1L << 31)) ? 1 : 0;
  if (cd360_tune().attn_fast == 0) p.fast = 0;  // tuning / debug: forces the guarded path
  const long nwg = (long)p.n_qtiles * B * H;
  if (nwg > 0x7fffffffL) return CD360_ERR_SHAPE;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

extern "C" int cd360_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                                   const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                   const int64_t* o_strides, float scale, void* stream) {
  CD360_TUNE_SCOPE(stream);
  return attn_launch(q, k, v, o, B, H, Nq, Nk, q_strides, k_strides, v_strides, o_strides, scale, nullptr, nullptr, stream);
}

// The same operator for a q that already carries the softmax scale AND the base change: q' = q * (scale * log2 e), o = softmax_2(q' k^T) v
// with softmax_2 the base-2 softmax (identical to softmax(scale q k^T) v).  The transformer blocks multiply the q rows of the merged
// q|k|v projection (weights, bias, LayerNorm-fold terms) by that factor when they pack it, so q' is rounded to bf16 once -- exactly
// like the reference's q -- and the whole-tile self-attention kernel (attn_self_kernel) needs no per-score multiply.
extern "C" int cd360_attn_fwd_prescaled_bf16(const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                                             const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                             const int64_t* o_strides, void* stream) {
  CD360_TUNE_SCOPE(stream);
  return attn_launch(q, k, v, o, B, H, Nq, Nk, q_strides, k_strides, v_strides, o_strides, 1.f, nullptr, nullptr, stream, true);
}

// Training forward: the same kernels, additionally writing lse [B*H, Nq] fp32 (natural-log log-sum-exp of the scaled scores of every
// query row), the only statistic cd360_attn_bwd_bf16 needs besides q, k, v, o.
extern "C" int cd360_attn_fwd_lse_bf16(const void* q, const void* k, const void* v, void* o, void* lse, int B, int H, int Nq, int Nk,
                                       const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                       const int64_t* o_strides, float scale, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!lse) return CD360_ERR_ARG;
  return attn_launch(q, k, v, o, B, H, Nq, Nk, q_strides, k_strides, v_strides, o_strides, scale, nullptr, (float*)lse, stream);
}

// fp8-MFMA variant for Nk <= 96 (BASELINE config 5): same bf16 tensors, Q / K / V^T / P rounded to e4m3 in registers.
// amax = {max|q|, max|k|, max|v|} (host floats) define the per-tensor scales.  CD360_ERR_SHAPE when Nk > 96.
extern "C" int cd360_attn_fwd_fp8mfma_bf16(const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                                           const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                           const int64_t* o_strides, float scale, const float* amax, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!amax) return CD360_ERR_ARG;
  return attn_launch(q, k, v, o, B, H, Nq, Nk, q_strides, k_strides, v_strides, o_strides, scale, amax, nullptr, stream);
}

// xformers-layout convenience entry: q, k, v, o all contiguous [BH, N, 64] (attention.py:393-408), consumed in place
extern "C" int cd360_attn_fwd_xformers_bf16(const void* q, const void* k, const void* v, void* o, int BH, int Nq, int Nk, float scale,
                                            void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (BH <= 0) return CD360_ERR_ARG;
  const int64_t qs[3] = {0, (int64_t)Nq * 64, 64}, ks[3] = {0, (int64_t)Nk * 64, 64};
  return cd360_attn_fwd_bf16(q, k, v, o, 1, BH, Nq, Nk, qs, ks, ks, qs, scale, stream);
}
