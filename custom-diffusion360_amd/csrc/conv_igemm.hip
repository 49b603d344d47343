// 3x3 (stride 1, pad 1) convolution and plain GEMM on channels-last bf16 as ONE implicit-GEMM MFMA kernel with fused
// epilogues -- the convs that bracket the injected transformer blocks:
//   ResBlock  in_layers[2]  : conv3x3(SiLU(GN(x))) + bias + emb[n, :]          (sgm/modules/diffusionmodules/openaimodel.py:352-374)
//   ResBlock  out_layers[3] : conv3x3(SiLU(GN(h))) + bias + skip               (openaimodel.py:375-376)
//   Upsample.conv, UNet.out : conv3x3 + bias                                    (openaimodel.py:161-164, 967-973)
// MIOpen serves these shapes with split-K asm kernels (fp32 atomics + separate zero-fill and cast passes, ~270 TF/s effective
// at SDXL sizes, see profiles/r01a_*); here the 9 taps are 9 shifted views of the same NHWC tensor accumulated into one MFMA
// accumulator, bias / time-embedding / residual are applied in registers and the bf16 result is written once.
//
//   out^T[co, pixel] = sum_{tap, ci} Wp[co][tap][ci] * x[pixel + shift(tap)][ci]       (zero outside the image)
// Tile 128 co x 128 pixels x 64 ci per step, 4 waves (2 x 2), each 64 x 64 on v_mfma_f32_32x32x16_bf16; the weight operand
// rows are loaded through the permutation `chan_pos` so that a lane ends up with 16 CONSECUTIVE output channels of one pixel
// (32-byte stores); both LDS tiles use the 16-B XOR swizzle of attn_fwd.hip (conflict-free ds_read_b128); the next K-step's
// tiles travel HBM -> registers while the current one is multiplied (LDS double-buffered, one barrier per step).
#include "cd360_common.h"
#include "cd360_tuning.h"
#include <type_traits>
#include <stdlib.h>

namespace {

struct ConvParams {
  const uint16_t* x;      // [M, Cin] pixels (n, y, x) row-major, channels contiguous
  const uint16_t* w;      // [Cout, taps * Cin] packed: k = tap * Cin + ci, tap = ky * 3 + kx
  const float* bias;      // [Cout] or null
  const uint16_t* emb;    // [N, Cout] bf16 per-image addend (row stride emb_stride elements) or null
  long emb_stride;
  const uint16_t* res;    // [M, Cout] bf16 residual or null
  uint16_t* out;          // [M, Cout]
  float* stats;           // optional [n_mtiles * slabs_per_tile, Cout, 2]: per-slab channel sums / sums of squares of `out`
  int N, H, W, Cin, Cout, taps;
  int stride, Ho, Wo;     // output pixel (yo, xo) reads input (stride*yo + ky - 1, stride*xo + kx - 1); Ho = H / stride, Wo = W / stride
  long M;
  int n_mtiles, n_ntiles, w_major;
  int kgroup;  // K order: 64-channel chunks per group (cd360_conv_k_order): group outer, tap middle, chunk-in-group inner
};

constexpr int BM = 128;  // pixels per tile (both tilings)

__device__ __forceinline__ int chan_pos(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

// BK = 64 input channels per K-step: 2 x 32 KB LDS stages per 256-thread group (BK = 32 with more workgroups per CU measured slower).
// SPLIT = 1: 256 threads own the tile.  SPLIT = 2: 512 threads; the second group of 4 waves runs the ODD K-steps of the same
// tile through its own LDS stages and the two accumulators are summed through LDS at the end -- an in-workgroup split-K for
// launches with no more tiles than CUs (the 32x32 level: 240 tiles), where one 4-wave workgroup per CU leaves one wave per SIMD
// and nothing to hide LDS / MFMA latency behind.  No atomics, no workspace, deterministic.
// Wave tiling: WAVES_CO x (4 / WAVES_CO) waves, each NCB x NPB MFMA tiles of 32 channels x 32 pixels.
//   <2, 2, 2>: 128 channels x 128 pixels, waves 2 x 2 of 64 x 64 -- the default;
//   <1, 5, 1>: 160 channels x 128 pixels, 4 waves of 160 x 32 -- for Cout = 320 (2 exact tiles instead of 2.5 of 128: no
//              half-empty third tile, a third fewer workgroups for the same pixels, weights read from LDS once per 32 pixels).
template <int SPLIT, int WAVES_CO, int NCB, int NPB>
__global__ __launch_bounds__(256 * SPLIT) void conv_igemm_kernel(ConvParams p) {
  constexpr int WAVES_PX = 4 / WAVES_CO;
  constexpr int BNC = WAVES_CO * NCB * 32;  // channels per tile
  static_assert(WAVES_PX * NPB * 32 == BM, "pixel tile is 128");
  constexpr int BK = 64;
  constexpr int PITCH = BK * 2;             // bytes per LDS row
  constexpr int CPR = BK / 8;               // 16-byte chunks per row
  constexpr int SH = 1;                     // rows per 256-byte bank row = 1 << SH
  constexpr int RPP = 256 / CPR;            // rows staged per pass
  constexpr int NPASS = BM / RPP;           // staging passes of the pixel tile
  constexpr int WPASS = BNC / RPP;          // ... of the weight tile
  constexpr int STAGE = (BM + BNC) * PITCH;
  constexpr int KS = BK / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];  // SPLIT * 2 * STAGE bytes (64 KB / 128 KB)
  auto swz = [](int row, int chunk) { return row * PITCH + ((chunk ^ ((row >> SH) & (CPR - 1))) << 4); };

  const int grp = SPLIT == 1 ? 0 : (int)(threadIdx.x >> 8);  // K-split group
  unsigned char* lds = lds_all + grp * 2 * STAGE;
  const int tid = threadIdx.x & 255, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wco = wave / WAVES_PX, wpx = wave % WAVES_PX;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  // tile order: each XCD walks a contiguous range of tiles (see the host side for which operand is re-streamed)
  const int nt = p.w_major ? tile / p.n_mtiles : tile % p.n_ntiles;
  const int mt = p.w_major ? tile % p.n_mtiles : tile / p.n_ntiles;
  const long m0 = (long)mt * BM;
  const int co0 = nt * BNC;
  const int HW = p.Ho * p.Wo;  // OUTPUT pixels per image
  const int chunk = tid % CPR, lrow = tid / CPR;

  // ---- per-thread staging geometry: NPASS weight rows and NPASS pixel rows (fixed for the whole K loop) ----
  // Loads go through buffer descriptors: a row that does not exist (channel >= Cout, pixel >= M, tap outside the image) gets an
  // offset beyond num_records and the hardware returns zeros -- no branches, no zero-fill moves in the K loop.
  constexpr uint32_t OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long)p.Cout * p.taps * p.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long)p.N * p.H * p.W * p.Cin * 2), 0x00020000);
  uint32_t woff[WPASS], xbase[NPASS];
  int py[NPASS], px[NPASS];
  bool pok[NPASS];
#pragma unroll
  for (int ps = 0; ps < WPASS; ++ps) {
    const int co = co0 + lrow + RPP * ps;
    woff[ps] = co < p.Cout ? (uint32_t)(((long)co * p.taps * p.Cin + chunk * 8) * 2) : OOB;
  }
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const long m = m0 + lrow + RPP * ps;
    pok[ps] = m < p.M;
    const long mc = pok[ps] ? m : 0;
    const int img = (int)(mc / HW), rem = (int)(mc - (long)img * HW);
    const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
    py[ps] = yo * p.stride;  // input coordinates of the tap centre
    px[ps] = xo * p.stride;
    xbase[ps] = (uint32_t)(((((long)img * p.H + py[ps]) * p.W + px[ps]) * p.Cin + chunk * 8) * 2);
  }

  const int kchunks = p.Cin / BK;
  const int nsteps_all = p.taps * kchunks;                       // K-steps of the tile
  const int nsteps = (nsteps_all - grp + SPLIT - 1) / SPLIT;     // ... of which this group runs grp, grp + SPLIT, ...
  const int niter = (nsteps_all + SPLIT - 1) / SPLIT;            // barrier count, the same for both groups

  // K order: groups of G = p.kgroup 64-channel chunks OUTER, tap MIDDLE, chunk-in-group INNER (step s = (cg * taps + tap) * G + j).
  // The nine taps of one group re-read the same (128 + halo) pixels x G*64 channels (<= 100 KB per pixel tile) back to back, so
  // eight of the nine reads are L2 hits -- pure tap-outer order (G = all chunks) re-streamed the whole pixel tile from the fabric
  // for every tap (2-4x the fetch traffic) -- while the per-tap shifts / image-border tests are still hoisted out of G
  // consecutive steps and each pixel row is read G*128 contiguous bytes at a time (G = 1, tap innermost, measured 7 % slower).
  const int G = p.kgroup;
  uint32_t xtap[NPASS];
  auto set_tap = [&](int tap) {
    const int dy = p.taps == 9 ? tap / 3 - 1 : 0, dx = p.taps == 9 ? tap - (tap / 3) * 3 - 1 : 0;
    const int xoff = (dy * p.W + dx) * p.Cin * 2;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const bool ok = pok[ps] && (unsigned)(py[ps] + dy) < (unsigned)p.H && (unsigned)(px[ps] + dx) < (unsigned)p.W;
      xtap[ps] = ok ? (uint32_t)((int)xbase[ps] + xoff) : OOB;
    }
  };
  // (cg, tap, j) and the linear index of the NEXT step this K-split group loads: steps grp, grp + SPLIT, ...
  int ld_s = grp, ld_j = grp % G, ld_tap = (grp / G) % p.taps, ld_cg = grp / (G * p.taps);
  set_tap(ld_tap);

  // One register set in flight (a second set -- loads two steps ahead -- was measured 10-25 % SLOWER: 256 VGPRs, worse interleave)
  u32x4 wreg[WPASS], xreg[NPASS];
  auto load_next = [&]() {
    const uint32_t wstep = (uint32_t)(ld_s * BK * 2);                 // weights are packed in step order: consecutive 128 B
    const uint32_t koff = (uint32_t)((ld_cg * G + ld_j) * BK * 2);    // channel chunk of the pixel rows
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) wreg[ps] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff[ps] + wstep, 0, 0);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) xreg[ps] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xtap[ps] + koff, 0, 0);
    ld_s += SPLIT;
    ld_j += SPLIT;
    if (ld_j >= G) {
      do { ld_j -= G; ++ld_tap; } while (ld_j >= G);
      while (ld_tap >= p.taps) { ld_tap -= p.taps; ++ld_cg; }
      set_tap(ld_tap);
    }
  };
  auto store_step = [&](unsigned char* base) {
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) *reinterpret_cast<u32x4*>(base + swz(lrow + RPP * ps, chunk)) = wreg[ps];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) *reinterpret_cast<u32x4*>(base + BNC * PITCH + swz(lrow + RPP * ps, chunk)) = xreg[ps];
  };

  f32x16 acc[NCB][NPB];  // [channel block][pixel block]
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][pb][i] = 0.f;

  // Schedule: the registers hold tile t+1 when iteration t starts (its loads were issued one whole iteration earlier).  They
  // are written to the other LDS buffer FIRST -- that buffer's last readers passed the previous barrier -- and re-issued at once
  // for tile t+2, so the ds_writes drain under this iteration's MFMAs and the single barrier at the end of the iteration never
  // waits for a load or a write pass.  (Load-at-the-top / write-before-the-barrier measured the same within 2 % at 2
  // workgroups per CU and 1-2 % slower at 1.)
  load_next();
  store_step(lds);
  if (nsteps > 1) load_next();
  __syncthreads();

  // Cout = 320 leaves the upper 64 channels of the third 128-channel tile empty: those two waves only help staging and skip
  // the fragment reads and MFMAs, which hands their SIMDs' matrix pipes to the co-resident workgroup
  const bool wave_has_channels = co0 + wco * NCB * 32 < p.Cout;
  const int wrow0 = wco * NCB * 32 + chan_pos(l31);  // + 32 * cb : weight-tile row feeding MFMA A-operand row l31
  const int prow0 = wpx * NPB * 32 + l31;            // + 32 * pb : pixel-tile row feeding MFMA B-operand column l31
  auto k_loop = [&](auto compute_tag) {
    constexpr bool COMPUTE = decltype(compute_tag)::value;
    for (int step = 0; step < nsteps; ++step) {
      const unsigned char* Ws = lds + (step & 1) * STAGE;
      const unsigned char* Xs = Ws + BNC * PITCH;
      if (step + 1 < nsteps) {
        store_step(lds + ((step + 1) & 1) * STAGE);
        if (step + 2 < nsteps) load_next();
      }
      if (COMPUTE) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 a[NCB], b[NPB];
#pragma unroll
          for (int i = 0; i < NCB; ++i) a[i] = *reinterpret_cast<const bf16x8*>(Ws + swz(wrow0 + 32 * i, 2 * ks + hh));
#pragma unroll
          for (int i = 0; i < NPB; ++i) b[i] = *reinterpret_cast<const bf16x8*>(Xs + swz(prow0 + 32 * i, 2 * ks + hh));
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb], b[pb], acc[cb][pb], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      __syncthreads();
    }
  };
  if (wave_has_channels) k_loop(std::true_type{}); else k_loop(std::false_type{});
  if (SPLIT == 2 && nsteps < niter) __syncthreads();  // odd step count: the other group had one more iteration

  if (SPLIT == 2) {  // sum the two K-halves through LDS (64 fp32 per thread = 64 KB, in the now idle stages), group 0 finishes
    float* red = reinterpret_cast<float*>(lds_all);
    if (grp == 1) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((cb * NPB + pb) * 16 + r) * 256 + tid] = acc[cb][pb][r];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][pb][r] += red[((cb * NPB + pb) * 16 + r) * 256 + tid];
  }

  // ---- epilogue: lane = pixel (l31), registers = 16 consecutive channels (16*hh + r) of each 32-channel block ----
  // Optional GroupNorm pre-pass (p.stats): per (pixel slab, channel) sum and sum of squares of the bf16 outputs this wave writes,
  // slab = (pixel tile, wpx) = NPB*32 pixels -- exactly the [N, nslab, C, 2] partials gn_finalize_kernel consumes, so the
  // GroupNorm that follows the conv skips its own read pass over the tensor.  Fixed reduction order: deterministic.
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int co = co0 + wco * NCB * 32 + cb * 32 + 16 * hh;
    const bool co_ok = co < p.Cout;  // Cout is a multiple of 16
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[r] = 0.f; ssq[r] = 0.f; }
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      const long m = m0 + wpx * NPB * 32 + pb * 32 + l31;
      if (m >= p.M || !co_ok) continue;
      const int img = (int)(m / HW);
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[cb][pb][r];
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += p.bias[co + r];
      }
      if (p.emb) {
        const u32x4 e0 = *reinterpret_cast<const u32x4*>(p.emb + (long)img * p.emb_stride + co);
        const u32x4 e1 = *reinterpret_cast<const u32x4*>(p.emb + (long)img * p.emb_stride + co + 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] += bf16lo_to_f32(e0[e]); v[2 * e + 1] += bf16hi_to_f32(e0[e]);
          v[8 + 2 * e] += bf16lo_to_f32(e1[e]); v[8 + 2 * e + 1] += bf16hi_to_f32(e1[e]);
        }
      }
      if (p.res) {
        const u32x4 e0 = *reinterpret_cast<const u32x4*>(p.res + m * p.Cout + co);
        const u32x4 e1 = *reinterpret_cast<const u32x4*>(p.res + m * p.Cout + co + 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] += bf16lo_to_f32(e0[e]); v[2 * e + 1] += bf16hi_to_f32(e0[e]);
          v[8 + 2 * e] += bf16lo_to_f32(e1[e]); v[8 + 2 * e + 1] += bf16hi_to_f32(e1[e]);
        }
      }
      u32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        o1[e] = pack_bf16x2(v[8 + 2 * e], v[8 + 2 * e + 1]);
      }
      uint16_t* dst = p.out + m * p.Cout + co;
      *reinterpret_cast<u32x4*>(dst) = o0;
      *reinterpret_cast<u32x4*>(dst + 8) = o1;
      if (p.stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = bf16lo_to_f32(o0[e]), a1 = bf16hi_to_f32(o0[e]), b0 = bf16lo_to_f32(o1[e]), b1 = bf16hi_to_f32(o1[e]);
          ssum[2 * e] += a0; ssq[2 * e] = fmaf(a0, a0, ssq[2 * e]);
          ssum[2 * e + 1] += a1; ssq[2 * e + 1] = fmaf(a1, a1, ssq[2 * e + 1]);
          ssum[8 + 2 * e] += b0; ssq[8 + 2 * e] = fmaf(b0, b0, ssq[8 + 2 * e]);
          ssum[8 + 2 * e + 1] += b1; ssq[8 + 2 * e + 1] = fmaf(b1, b1, ssq[8 + 2 * e + 1]);
        }
      }
    }
    if (p.stats) {  // wave-uniform
      // Sum over this wave's 32 pixel lanes through LDS (the K-loop stages are idle after the last barrier): every lane stores its
      // 16 channel sums / sums of squares as one padded row (36 floats: conflict-free ds_write_b128), then lane j adds half a
      // column -- 16 ds_read_b32 per quantity -- and one exchange with lane ^ 32 finishes it.  (Butterfly shuffles: 160 per
      // quantity, measured 3x the cost.)  Fixed order: deterministic.
      // (SPLIT = 2: the first 64 KB may still be read by slower waves finishing the K-split reduction -- use the second half)
      float* red = reinterpret_cast<float*>(lds_all + (SPLIT == 2 ? 2 * STAGE : 0)) + wave * (2 * 32 * 36);
      float* mine = red + l31 * 36 + 16 * hh;
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        *reinterpret_cast<f32x4*>(mine + r) = f32x4{ssum[r], ssum[r + 1], ssum[r + 2], ssum[r + 3]};
        *reinterpret_cast<f32x4*>(mine + 32 * 36 + r) = f32x4{ssq[r], ssq[r + 1], ssq[r + 2], ssq[r + 3]};
      }
      // wave-private region: no workgroup barrier; the compiler's lgkmcnt wait orders the ds_write before the ds_read
      const int c = lane & 31, half = lane >> 5;
      float ts = 0.f, tq = 0.f;
#pragma unroll
      for (int row = 0; row < 16; ++row) {
        ts += red[(half * 16 + row) * 36 + c];
        tq += red[32 * 36 + (half * 16 + row) * 36 + c];
      }
      ts += __shfl_xor(ts, 32);
      tq += __shfl_xor(tq, 32);
      const int cch = co0 + wco * NCB * 32 + cb * 32 + c;
      if (half == 0 && cch < p.Cout) {
        float* dst = p.stats + (((long)mt * WAVES_PX + wpx) * p.Cout + cch) * 2;
        dst[0] = ts;
        dst[1] = tq;
      }
    }
  }
}

}  // namespace

// x [N*H*W, Cin] channels-last bf16; w_packed [Cout, taps*Cin] bf16 in the kernel's K order (tap = ky*3+kx, 3x3 / stride 1 / pad 1):
// with G = cd360_conv_k_order(Cin, taps), k = ((cg * taps + tap) * G + j) * 64 + ci % 64 where ci / 64 = cg * G + j;
// taps = 1: the plain [Cout, Cin] matrix, out = x @ w^T; bias fp32 [Cout] | NULL; emb bf16 [N, Cout] | NULL (added per image; rows
// emb_stride elements apart, so a column slice of a wider matrix can be passed);
// res bf16 [N*Ho*Wo, Cout] | NULL; out bf16 [N*Ho*Wo, Cout].  Cin % 64 == 0, Cout % 16 == 0.  stride = 1, or 2 (3x3, pad 1, even H and W:
// Downsample.op, openaimodel.py:190-213) with Ho = H / 2, Wo = W / 2.
// K order of w_packed for a conv with `Cin` input channels: returns G, the number of 64-channel chunks per group; the K index of
// (tap, ci) is k = ((cg * taps + tap) * G + j) * 64 + ci % 64 with chunk = ci / 64 = cg * G + j.  G = Cin / 64 is plain tap-major
// order (k = tap * Cin + ci), G = 1 chunk-major.  Default: the largest divisor of Cin / 64 that is <= 5 (5 for every SDXL width).
extern "C" int cd360_conv_k_order(int Cin, int taps) {
  const int kchunks = Cin / 64;
  if (taps != 9 || kchunks <= 0) return kchunks > 0 ? kchunks : 1;
  int gmax = 5;
  if (cd360_tune_default().conv_kgroup > 0) gmax = cd360_tune_default().conv_kgroup;  // process-wide override (set before weights are packed)
  for (int g = gmax < kchunks ? gmax : kchunks; g > 1; --g)
    if (kchunks % g == 0) return g;
  return 1;
}

// Pixel slabs per 128-pixel tile of the optional `tile_stats` output (the wave tiling the launch will use for this Cout)
extern "C" int cd360_conv_stats_slabs(int Cout) { return (Cout % 160 == 0 && Cout % 128 != 0 && cd360_tune().conv_wide != 0) ? 4 : 2; }

// tile_stats (optional): fp32 [N*H*W / 128 * cd360_conv_stats_slabs(Cout), Cout, 2] = per pixel slab (32 or 64 consecutive pixels)
// and channel, the sum and the sum of squares of the bf16 outputs -- the first pass of the GroupNorm that follows the conv
// (cd360_gn_silu_bf16's `tile_stats`).  Requires H*W % 128 == 0 (slabs must not straddle images).
extern "C" int cd360_gemm_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                               const void* bias, const void* res, int64_t ldr, const void* ln_stats, int ln_parts, int ln_dim, float ln_eps,
                               const void* wsum, void* stats_out, int flags, void* stream);
extern "C" int cd360_conv_dma_slab_rows(int N, int H, int W, int Cin, int Cout, int taps, int stride);
extern "C" int cd360_conv3x3_dma_bf16(const void* x, const void* w_packed, const void* bias, const void* emb, int64_t emb_stride, const void* res,
                                      void* out, int N, int H, int W, int Cin, int Cout, void* tile_stats, void* stream);

// Pixels per slab of the `tile_stats` output for this convolution: decided by the kernel that will serve the call (the 3 x 3 / stride 1
// convolutions run on the LDS-DMA GEMM core of gemm8p.hip, everything else on conv_igemm_kernel)
extern "C" int cd360_conv_stats_rows(int N, int H, int W, int Cin, int Cout, int taps, int stride) {
  const int rows = cd360_conv_dma_slab_rows(N, H, W, Cin, Cout, taps, stride);
  return rows > 0 ? rows : 128 / cd360_conv_stats_slabs(Cout);
}

extern "C" int cd360_conv_igemm_bf16(const void* x, const void* w_packed, const void* bias, const void* emb, int64_t emb_stride, const void* res,
                                     void* out, int N, int H, int W, int Cin, int Cout, int taps, int stride, void* tile_stats, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!x || !w_packed || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return CD360_ERR_ARG;
  if (cd360_conv_dma_slab_rows(N, H, W, Cin, Cout, taps, stride) > 0) {  // 3 x 3 / stride 1: the LDS-DMA core (gemm8p.hip, EPI 5)
    const int rc = cd360_conv3x3_dma_bf16(x, w_packed, bias, emb, emb_stride, res, out, N, H, W, Cin, Cout, tile_stats, stream);
    // outside that entry's envelope (e.g. an emb row stride it cannot read): the register-staged kernel below serves the call -- unless
    // statistics were asked for, whose slab size cd360_conv_stats_rows() has already promised from the DMA tiling
    if (rc != CD360_ERR_SHAPE || tile_stats) return rc;
  }
  // 1 x 1 (the ResBlock skip_connection convs, openaimodel.py:335-343): a plain Linear over the pixels -- the same core through its GEMM entry
  if (taps == 1 && stride == 1 && !emb && !tile_stats && Cin % 64 == 0 && Cout % 16 == 0 && cd360_tune().conv_dma != 0) {
    const int rc = cd360_gemm_bf16(x, w_packed, out, (int64_t)N * H * W, Cout, Cin, Cin, Cin, Cout, bias, res, Cout, nullptr, 0, 0, 0.f, nullptr, nullptr, 0, stream);
    if (rc != CD360_ERR_SHAPE) return rc;
  }
  if ((taps != 9 && taps != 1) || Cin % 64 || Cout % 16) return CD360_ERR_SHAPE;
  if ((stride != 1 && stride != 2) || (stride == 2 && (taps != 9 || H % 2 || W % 2))) return CD360_ERR_SHAPE;
  if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)out | (uintptr_t)emb | (uintptr_t)res) % 16) return CD360_ERR_ARG;
  if ((long)N * H * W * Cin * 2 >= (1L << 31) || (long)Cout * taps * Cin * 2 >= (1L << 31)) return CD360_ERR_SHAPE;  // 32-bit buffer offsets
  ConvParams p;
  p.x = (const uint16_t*)x; p.w = (const uint16_t*)w_packed; p.bias = (const float*)bias; p.emb = (const uint16_t*)emb;
  p.res = (const uint16_t*)res; p.out = (uint16_t*)out; p.stats = (float*)tile_stats;
  p.emb_stride = emb ? emb_stride : 0;
  if (emb && (emb_stride < Cout || emb_stride % 8)) return CD360_ERR_SHAPE;  // rows of >= Cout elements, 16-byte aligned
  if (tile_stats && ((long)(H / stride) * (W / stride)) % BM) return CD360_ERR_SHAPE;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.taps = taps;
  p.stride = stride; p.Ho = H / stride; p.Wo = W / stride;
  p.M = (long)N * p.Ho * p.Wo;  // output pixels
  p.n_mtiles = (int)((p.M + BM - 1) / BM);
  // 160-channel tiles (4 waves of 160 x 32) when Cout is a multiple of 160 but not of 128 (Cout = 320): exact tiling
  bool wide = Cout % 160 == 0 && Cout % 128 != 0;
  if (cd360_tune().conv_wide == 0) wide = false;  // tuning override: always 128-channel tiles
  const int bnc = wide ? 160 : 128;
  p.n_ntiles = (Cout + bnc - 1) / bnc;
  p.kgroup = cd360_conv_k_order(Cin, taps);
  // Tile order across the 8 XCDs (each walks a contiguous range of tiles and has its own 4 MB L2): pixel-tile-major makes every
  // XCD stream ALL weights once per resident set, weight-tile-major makes every XCD stream all pixels, nine taps per K group --
  // which only stays in L2 while the group footprint of ALL pixel tiles (n_mtiles x G x 20 KB) fits.  So: weight-major when the
  // weights are at least ~0.9x the pixels AND that footprint is < ~3 MB (the 32x32 level: FETCH_SIZE 180 -> 81 MB per launch);
  // 1280->1280 at 64x64 (96 pixel tiles) went 287 -> 1080 MB weight-major with G = 5 and stays pixel-major.  Time is unaffected.
  p.w_major = (long)Cout * taps * 10 >= p.M * 9 && (long)p.n_mtiles * p.kgroup <= 160;
  if (cd360_tune().conv_wmajor >= 0) p.w_major = cd360_tune().conv_wmajor == 1;  // tuning override
  const long nwg = (long)p.n_mtiles * p.n_ntiles;
  if (nwg > 0x7fffffffL) return CD360_ERR_SHAPE;
  // in-workgroup split-K when the launch has no more tiles than CUs (one 4-wave workgroup per CU otherwise)
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n;
  }();
  int split = (!wide && nwg <= cus && taps * (Cin / 64) >= 8) ? 2 : 1;
  if (cd360_tune().conv_split == 1) split = 1;  // tuning override: 1 or 2
  if (cd360_tune().conv_split == 2 && !wide) split = 2;
  const int stage_bytes = (BM + bnc) * 64 * 2;
  if (wide) {  // 2 x 36 KB stages = 72 KB: above the 64 KB default, still two workgroups per CU
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<1, 1, 5, 1>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + 160) * 64 * 2);
    if (attr != hipSuccess) return CD360_ERR_LAUNCH;
    hipLaunchKernelGGL((conv_igemm_kernel<1, 1, 5, 1>), dim3((unsigned)nwg), dim3(256), 2 * stage_bytes, (hipStream_t)stream, p);
  } else if (split == 2) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<2, 2, 2, 2>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (BM + 128) * 64 * 2);  // 128 KB
    if (attr != hipSuccess) return CD360_ERR_LAUNCH;
    hipLaunchKernelGGL((conv_igemm_kernel<2, 2, 2, 2>), dim3((unsigned)nwg), dim3(512), 4 * stage_bytes, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<1, 2, 2, 2>), dim3((unsigned)nwg), dim3(256), 2 * stage_bytes, (hipStream_t)stream, p);
  }
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// A11: pose_emb_layers(cat[x, xref]) = Linear(2C -> C, no bias) (sgm/modules/attention.py:515-516,634) without the concat:
// out = x wa^T + xref wb^T with wa = W[:, :C], wb = W[:, C:] (each [C, C] contiguous, bf16).  Two GEMM-mode launches of the implicit
// GEMM kernel, the second accumulating onto the first through the residual epilogue (res aliases out: every element is read and
// written by the same lane).  x, xref, out: [rows, C] bf16; C % 64 == 0.
extern "C" int cd360_pose_embed_bf16(const void* x, const void* xref, const void* wa, const void* wb, void* out, int64_t rows, int C,
                                     void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!x || !xref || !wa || !wb || !out || rows <= 0 || C <= 0) return CD360_ERR_ARG;
  if (rows > 0x7fffffffL) return CD360_ERR_SHAPE;
  const int rc = cd360_conv_igemm_bf16(x, wa, nullptr, nullptr, 0, nullptr, out, 1, (int)rows, 1, C, C, 1, 1, nullptr, stream);
  if (rc != CD360_OK) return rc;
  return cd360_conv_igemm_bf16(xref, wb, nullptr, nullptr, 0, out, out, 1, (int)rows, 1, C, C, 1, 1, nullptr, stream);
}
