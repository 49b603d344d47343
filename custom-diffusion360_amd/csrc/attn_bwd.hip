// Flash attention backward, bf16 in / fp32 accumulate / bf16 out, head dim 64, no mask.
//
// The reference trains through xformers.ops.memory_efficient_attention's autograd (sgm/modules/attention.py:393-408; the
// fine-tuning loop of configs/train_co3d_concept.yaml, BASELINE config 4).  This is that backward for the layouts of
// cd360_attn_fwd_bf16: given q, k, v, o, dO and the forward's lse,
//     P  = exp(scale q k^T - lse)          dV = P^T dO          dP = dO v^T
//     dS = P o (dP - delta)                dQ = scale dS k      dK = scale dS^T q         delta = rowsum(dO o O)
//
// One kernel skeleton, two roles (template DKV), both in the forward's "swapped" MFMA convention (v_mfma_f32_32x32x16_bf16,
// lane = one column of the STATIONARY side, registers = rows of the STREAMED side):
//   DKV = false  stationary = 32 queries per wave (B fragments of q and dO, lse / delta one scalar per lane);
//                streamed   = 64-key tiles of k and v through LDS;    accumulates dQ^T  (64 d x 32 queries).
//   DKV = true   stationary = 32 keys per wave (B fragments of k and v);
//                streamed   = 64-query tiles of q and dO (+ their lse / delta) through LDS;  accumulates dK^T and dV^T.
// Per tile:   S^T  = X F1^T  (row fragments of X = k | q)          dP^T = Y F2^T  (row fragments of Y = v | dO)
//             P, dS in registers, packed to bf16 in the accumulator's own row order, and used directly as MFMA B operands of
//             acc1 += X^T dS   (dQ^T | dK^T)        acc2 += Y^T P   (dV^T, DKV only)
// with X^T / Y^T read from the same row-major LDS tiles by gfx950's transposing read ds_read_b64_tr_b16 (lane mapping as in
// attn_fwd.hip: tools/probe/tr_read_probe.hip).  Neither P nor dS ever goes through LDS or HBM, and nothing is atomically
// accumulated: dQ rows belong to one workgroup of the first launch, dK / dV rows to one workgroup of the second.
// Tiles are register-prefetched and double-buffered in LDS (one barrier per tile); the stationary rows and the output rows move
// between HBM and LDS as full 128-byte lines.  The transposed reads still use the K swizzle (some bank conflicts): next step.
#include "cd360_common.h"

namespace {

struct AttnBwdParams {
  const uint16_t *q, *k, *v, *dout, *o;
  const float* lse;    // [B*H, Nq]
  float* delta;        // [B*H, Nq] workspace: written by the dQ role (or attn_bwd_delta_kernel), read by the dK/dV role
  long o_s[3];
  int fused_delta;     // 1: the dQ role computes delta = rowsum(dO o O) of its own rows itself
  uint16_t *dq, *dk, *dv;
  int B, H, Nq, Nk;
  long q_s[3], k_s[3], v_s[3], do_s[3], dq_s[3], dk_s[3], dv_s[3];  // (batch, head, row) element strides, d contiguous
  float scale, scale_log2e;
  int n_tiles;  // 128-row stationary tiles per (batch, head)
};

constexpr int PITCH = 128;  // bytes per LDS row: 64 d * 2 B, 16-byte chunks XOR-swizzled by (row >> 1) & 7
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ int lds_off(int row, int d) { return row * PITCH + ((((d >> 3) ^ ((row >> 1) & 7))) << 4) + (d & 7) * 2; }

// A fragment, rows = tile rows 32 kb + lane%32, k = d 16 ks + 8 hh .. + 7
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* T, int kb, int ks, int l31, int hh) {
  return *reinterpret_cast<const bf16x8*>(T + lds_off(kb * 32 + l31, 16 * ks + 8 * hh));
}

// A fragment of the TRANSPOSED tile: rows = d 32 db + lane%32, k = the 16 tile rows of group kk in the order of the accumulator
// registers (16 kk + 4 hh + {0..3}, then + 8).  In each 16-lane group lane s supplies the address of 4 consecutive d (chunk s & 3)
// of tile row s >> 2, and receives its own d column of the 4 rows.
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* T, int db, int kk, int l31, int hh) {
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int s = l31 & 15;
  const int row = 16 * kk + 4 * hh + (s >> 2), d0 = 32 * db + 16 * (l31 >> 4) + 4 * (s & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(T + lds_off(row, d0)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(T + lds_off(row + 8, d0)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// delta[bh, row] = sum_d dO * O   (one 8-lane group per row: 16-byte pieces of full 128-byte lines)
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const uint16_t* o, const uint16_t* dout, float* delta, int H, int Nq, long o_sb,
                                                             long o_sh, long o_sn, long do_sb, long do_sh, long do_sn, long rows) {
  const long gid = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int chunk = threadIdx.x & 7;
  float acc = 0.f;
  if (gid < rows) {
    const long bh = gid / Nq, r = gid - bh * Nq;
    const long b = bh / H, h = bh - b * H;
    const u32x4 a = *reinterpret_cast<const u32x4*>(o + b * o_sb + h * o_sh + r * o_sn + chunk * 8);
    const u32x4 g = *reinterpret_cast<const u32x4*>(dout + b * do_sb + h * do_sh + r * do_sn + chunk * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += bf16lo_to_f32(a[i]) * bf16lo_to_f32(g[i]) + bf16hi_to_f32(a[i]) * bf16hi_to_f32(g[i]);
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  if (gid < rows && chunk == 0) delta[gid] = acc;
}

template <bool DKV>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnBwdParams p) {
  // two tile buffers (X | Y, 64 rows x 128 B each) + the tile's lse / delta (DKV); before the loop the same 32 KB stage the
  // workgroup's 128 stationary rows of both tensors, after it each wave's 4 KB slice stages its output rows
  constexpr int BUF_BYTES = 2 * 64 * PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF_BYTES + 2 * 2 * 64 * 4];
  float* st_all = reinterpret_cast<float*>(lds + 2 * BUF_BYTES);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = tile / p.n_tiles, t = tile - bh * p.n_tiles;
  const int b = bh / p.H, h = bh - b * p.H;
  const int Ns = DKV ? p.Nk : p.Nq, Nt = DKV ? p.Nq : p.Nk;  // stationary / streamed row counts
  const uint16_t* f1p = (DKV ? p.k + b * p.k_s[0] + h * p.k_s[1] : p.q + b * p.q_s[0] + h * p.q_s[1]);
  const uint16_t* f2p = (DKV ? p.v + b * p.v_s[0] + h * p.v_s[1] : p.dout + b * p.do_s[0] + h * p.do_s[1]);
  const long f1_sn = DKV ? p.k_s[2] : p.q_s[2], f2_sn = DKV ? p.v_s[2] : p.do_s[2];
  const uint16_t* xp = (DKV ? p.q + b * p.q_s[0] + h * p.q_s[1] : p.k + b * p.k_s[0] + h * p.k_s[1]);
  const uint16_t* yp = (DKV ? p.dout + b * p.do_s[0] + h * p.do_s[1] : p.v + b * p.v_s[0] + h * p.v_s[1]);
  const long x_sn = DKV ? p.q_s[2] : p.k_s[2], y_sn = DKV ? p.do_s[2] : p.v_s[2];
  const float* lsep = p.lse + (long)bh * p.Nq;
  const float* delp = p.delta + (long)bh * p.Nq;  // (dK/dV role; dQ role without fused delta)
  constexpr float LOG2E = 1.4426950408889634f;
  const int lrow = tid >> 3, lchunk = tid & 7;  // line-shaped access: 8 lanes x 16 B = one 128-byte row

  // ---- stationary side: rows travel as full 128-byte lines into LDS and come back as B fragments (column lane%32, k = d 16 ks + 8 hh ..);
  // loading the fragments straight from HBM touches every line four times with 32-byte pieces ----
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = lrow + 32 * pass, grow = t * 128 + row;
    u32x4 a = {0u, 0u, 0u, 0u}, c2 = {0u, 0u, 0u, 0u};
    if (grow < Ns) {
      a = *reinterpret_cast<const u32x4*>(f1p + (long)grow * f1_sn + lchunk * 8);
      c2 = *reinterpret_cast<const u32x4*>(f2p + (long)grow * f2_sn + lchunk * 8);
    }
    *reinterpret_cast<u32x4*>(lds + lds_off(row, lchunk * 8)) = a;
    *reinterpret_cast<u32x4*>(lds + 128 * PITCH + lds_off(row, lchunk * 8)) = c2;
  }
  __syncthreads();
  const int srow = t * 128 + wave * 32 + l31;
  bf16x8 f1[4], f2[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    f1[ks] = *reinterpret_cast<const bf16x8*>(lds + lds_off(wave * 32 + l31, 16 * ks + 8 * hh));
    f2[ks] = *reinterpret_cast<const bf16x8*>(lds + 128 * PITCH + lds_off(wave * 32 + l31, 16 * ks + 8 * hh));
  }
  float lse2_lane = 0.f, delta_lane = 0.f;
  if (!DKV && srow < Ns) {
    lse2_lane = lsep[srow] * LOG2E;
    if (!p.fused_delta) delta_lane = delp[srow];
  }
  __syncthreads();  // the staging area becomes the tile buffers
  if (!DKV && p.fused_delta) {
    // delta = rowsum(dO o O) of this workgroup's own query rows: one more trip of 128 lines through LDS instead of a separate
    // kernel that reads dO and O a second time; the dK/dV role (launched afterwards) finds it in the workspace
    const uint16_t* op = p.o + b * p.o_s[0] + h * p.o_s[1];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = lrow + 32 * pass, grow = t * 128 + row;
      u32x4 a = {0u, 0u, 0u, 0u};
      if (grow < Ns) a = *reinterpret_cast<const u32x4*>(op + (long)grow * p.o_s[2] + lchunk * 8);
      *reinterpret_cast<u32x4*>(lds + lds_off(row, lchunk * 8)) = a;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 ov = *reinterpret_cast<const u32x4*>(lds + lds_off(wave * 32 + l31, 16 * ks + 8 * hh));
      const u32x4 gv = __builtin_bit_cast(u32x4, f2[ks]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        delta_lane += bf16lo_to_f32(ov[e]) * bf16lo_to_f32(gv[e]) + bf16hi_to_f32(ov[e]) * bf16hi_to_f32(gv[e]);
    }
    delta_lane += __shfl_xor(delta_lane, 32);
    if (hh == 0 && srow < Ns) p.delta[(long)bh * p.Nq + srow] = delta_lane;
    __syncthreads();
  }

  f32x16 acc1[2], acc2[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc1[0][i] = 0.f; acc1[1][i] = 0.f; acc2[0][i] = 0.f; acc2[1][i] = 0.f; }
  const float c = p.scale_log2e;

  // ---- streamed tiles: the loads of tile i + 1 are in flight while tile i is consumed; two LDS buffers, one barrier per tile ----
  u32x4 xr[2], yr[2];
  float sr = 0.f;
  auto tile_load = [&](int t0) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = lrow + 32 * pass;
      u32x4 xv = {0u, 0u, 0u, 0u}, yv = {0u, 0u, 0u, 0u};
      if (t0 + row < Nt) {
        xv = *reinterpret_cast<const u32x4*>(xp + (long)(t0 + row) * x_sn + lchunk * 8);
        yv = *reinterpret_cast<const u32x4*>(yp + (long)(t0 + row) * y_sn + lchunk * 8);
      }
      xr[pass] = xv;
      yr[pass] = yv;
    }
    if (DKV && tid < 128) {
      const int row = tid & 63;
      sr = 0.f;
      if (t0 + row < Nt) sr = tid < 64 ? lsep[t0 + row] * LOG2E : delp[t0 + row];
    }
  };
  auto tile_store = [&](int buf) {
    unsigned char* Xb = lds + buf * BUF_BYTES;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = lrow + 32 * pass;
      *reinterpret_cast<u32x4*>(Xb + lds_off(row, lchunk * 8)) = xr[pass];
      *reinterpret_cast<u32x4*>(Xb + 64 * PITCH + lds_off(row, lchunk * 8)) = yr[pass];
    }
    if (DKV && tid < 128) st_all[buf * 128 + tid] = sr;
  };
  tile_load(0);
  tile_store(0);
  __syncthreads();

  const int n_t = (Nt + 63) / 64;
  for (int it = 0; it < n_t; ++it) {
    const int t0 = it * 64, buf = it & 1;
    const bool more = it + 1 < n_t;
    if (more) tile_load(t0 + 64);
    const unsigned char* Xs = lds + buf * BUF_BYTES;
    const unsigned char* Ys = Xs + 64 * PITCH;
    const float* st = st_all + buf * 128;

    // ---- S^T (streamed rows x stationary columns) and dP^T ----
    f32x16 sT[2], dpT[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { sT[0][i] = 0.f; sT[1][i] = 0.f; dpT[0][i] = 0.f; dpT[1][i] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Xs, kb, ks, l31, hh), f1[ks], sT[kb], 0, 0, 0);
        dpT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ys, kb, ks, l31, hh), f2[ks], dpT[kb], 0, 0, 0);
      }

    // ---- P and dS, packed in register order (tile row of register r of block kb: 32 kb + (r & 3) + 8 (r >> 2) + 4 hh) ----
    uint32_t pk[16], dsk[16];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float pv[2], dv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int rl = kb * 32 + ((r + e) & 3) + 8 * ((r + e) >> 2) + 4 * hh;
          const float l2 = DKV ? st[rl] : lse2_lane;
          const float dl = DKV ? st[64 + rl] : delta_lane;
          float pe = __builtin_amdgcn_exp2f(fmaf(sT[kb][r + e], c, -l2));
          if (t0 + rl >= Nt) pe = 0.f;
          pv[e] = pe;
          dv[e] = pe * (dpT[kb][r + e] - dl);
        }
        pk[kb * 8 + (r >> 1)] = pack_bf16x2(pv[0], pv[1]);
        dsk[kb * 8 + (r >> 1)] = pack_bf16x2(dv[0], dv[1]);
      }

    // ---- acc1 += X^T dS,  acc2 += Y^T P ----
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const u32x4 dsw = {dsk[kk * 4 + 0], dsk[kk * 4 + 1], dsk[kk * 4 + 2], dsk[kk * 4 + 3]};
      const u32x4 pw = {pk[kk * 4 + 0], pk[kk * 4 + 1], pk[kk * 4 + 2], pk[kk * 4 + 3]};
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        acc1[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Xs, db, kk, l31, hh), __builtin_bit_cast(bf16x8, dsw), acc1[db], 0, 0, 0);
        if (DKV)
          acc2[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ys, db, kk, l31, hh), __builtin_bit_cast(bf16x8, pw), acc2[db], 0, 0, 0);
      }
    }
    if (more) tile_store(buf ^ 1);  // (last read of that buffer: tile it - 1, before the previous barrier)
    __syncthreads();
  }

  // ---- epilogue: lane = stationary row, register r of block db = d 32 db + (r & 3) + 8 (r >> 2) + 4 hh.  The accumulators go
  // through this wave's 4 KB of LDS and leave as full 128-byte rows (8 lanes x 16 B) when the output rows are 16-byte aligned ----
  unsigned char* Os = lds + wave * (32 * PITCH);
  auto write_out = [&](const f32x16 (&acc)[2], float mul, uint16_t* obase, const long (&os)[3]) {
    uint16_t* orows = obase + b * os[0] + h * os[1];
    const bool lines = (os[0] % 8 == 0) && (os[1] % 8 == 0) && (os[2] % 8 == 0) && ((uintptr_t)obase % 16 == 0);
    if (lines) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * hh;
          const u32x2 wv = {pack_bf16x2(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul), pack_bf16x2(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul)};
          *reinterpret_cast<u32x2*>(Os + lds_off(l31, d)) = wv;
        }
      // (wave-private region: no barrier; the LDS writes of this wave are visible to its own later reads)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3), grow = t * 128 + wave * 32 + row;
        const u32x4 v = *reinterpret_cast<const u32x4*>(Os + lds_off(row, (lane & 7) * 8));
        if (grow < Ns) *reinterpret_cast<u32x4*>(orows + (long)grow * os[2] + (lane & 7) * 8) = v;
      }
    } else if (srow < Ns) {
      uint16_t* o1 = orows + (long)srow * os[2];
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * hh;
          const u32x2 wv = {pack_bf16x2(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul), pack_bf16x2(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul)};
          *reinterpret_cast<u32x2*>(o1 + d) = wv;
        }
    }
  };
  if (DKV) {
    write_out(acc1, p.scale, p.dk, p.dk_s);
    write_out(acc2, 1.f, p.dv, p.dv_s);
  } else {
    write_out(acc1, p.scale, p.dq, p.dq_s);
  }
}

}  // namespace

// Backward of cd360_attn_fwd_lse_bf16.  q, k, v, o, dout: bf16, the forward's layouts (element strides {batch, head, row}, d
// contiguous, 16-byte aligned rows); lse [B*H, Nq] fp32 from the forward; delta_ws: caller-allocated B*H*Nq floats of workspace.
// dq (may be NULL) and dk / dv (both or neither) are written in bf16 with their own strides (8-byte aligned rows), e.g. straight
// into the column slices of one d(q|k|v) buffer so the projection's backward is a single GEMM.
extern "C" int cd360_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* dout, const void* lse,
                                   void* dq, void* dk, void* dv, void* delta_ws, int B, int H, int Nq, int Nk, const int64_t* q_strides,
                                   const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, const int64_t* do_strides,
                                   const int64_t* dq_strides, const int64_t* dk_strides, const int64_t* dv_strides, float scale,
                                   void* stream) {
  if (!q || !k || !v || !o || !dout || !lse || !delta_ws || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return CD360_ERR_ARG;
  if ((dk == nullptr) != (dv == nullptr)) return CD360_ERR_ARG;
  if ((dq && !dq_strides) || (dk && (!dk_strides || !dv_strides))) return CD360_ERR_ARG;
  AttnBwdParams p;
  p.q = (const uint16_t*)q; p.k = (const uint16_t*)k; p.v = (const uint16_t*)v; p.dout = (const uint16_t*)dout;
  p.lse = (const float*)lse; p.delta = (float*)delta_ws; p.o = (const uint16_t*)o;
  p.dq = (uint16_t*)dq; p.dk = (uint16_t*)dk; p.dv = (uint16_t*)dv;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  for (int i = 0; i < 3; ++i) {
    p.q_s[i] = q_strides[i]; p.k_s[i] = k_strides[i]; p.v_s[i] = v_strides[i]; p.do_s[i] = do_strides[i];
    p.dq_s[i] = dq ? dq_strides[i] : 0; p.dk_s[i] = dk ? dk_strides[i] : 0; p.dv_s[i] = dv ? dv_strides[i] : 0;
    p.o_s[i] = o_strides[i];
    if (p.q_s[i] % 8 || p.k_s[i] % 8 || p.v_s[i] % 8 || p.do_s[i] % 8 || o_strides[i] % 8) return CD360_ERR_SHAPE;
    if (p.dq_s[i] % 4 || p.dk_s[i] % 4 || p.dv_s[i] % 4) return CD360_ERR_SHAPE;
  }
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout) % 16) return CD360_ERR_ARG;
  if (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 8) return CD360_ERR_ARG;
  p.scale = scale;
  p.scale_log2e = scale * 1.4426950408889634f;
  const long rows = (long)B * H * Nq;
  if (rows > 0x7fffffffL * 32) return CD360_ERR_SHAPE;
  p.fused_delta = dq ? 1 : 0;  // the dQ launch (first) leaves delta in the workspace for the dK/dV launch
  if (!dq) {
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)o,
                       (const uint16_t*)dout, (float*)delta_ws, H, Nq, (long)o_strides[0], (long)o_strides[1], (long)o_strides[2], p.do_s[0],
                       p.do_s[1], p.do_s[2], rows);
    CD360_LAUNCH_CHECK();
  }
  if (dq) {
    p.n_tiles = (Nq + 127) / 128;
    const long nwg = (long)p.n_tiles * B * H;
    if (nwg > 0x7fffffffL) return CD360_ERR_SHAPE;
    hipLaunchKernelGGL((attn_bwd_kernel<false>), dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    CD360_LAUNCH_CHECK();
  }
  if (dk) {
    p.n_tiles = (Nk + 127) / 128;
    const long nwg = (long)p.n_tiles * B * H;
    if (nwg > 0x7fffffffL) return CD360_ERR_SHAPE;
    hipLaunchKernelGGL((attn_bwd_kernel<true>), dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    CD360_LAUNCH_CHECK();
  }
  return CD360_OK;
}
