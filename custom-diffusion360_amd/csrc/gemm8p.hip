// out[M, N] = A[M, K] @ W[N, K]^T on bf16 MFMA with fused epilogues -- every Linear of the transformer blocks that bracket the
// FeatureNeRF injection (sgm/modules/attention.py:89-115 GEGLU feed-forward, :323-329 / :368-372 / :422 attention projections,
// :515-516 pose_emb_layers, :748 / :786 proj_in / proj_out) as ONE hand-scheduled kernel family instead of library GEMMs plus
// separate LayerNorm / GEGLU / residual passes.
//
// Structure (CDNA4), template <WM, WN, NCB, NMB>: WM x WN waves, each wave owns NCB x NMB blocks of 32 channels x 32 tokens on
// v_mfma_f32_32x32x16_bf16 in the "swapped" convention of the other kernels here (MFMA rows = output channels through `chan_pos`,
// MFMA columns = tokens) so a lane ends up with 16 consecutive output channels of one token.
//   <2, 4, 2, 4>: 256 x 256 tile, eight waves of 128 x 64 (two per SIMD: one wave's MFMAs cover the other's LDS / DMA issue);
//   <4, 2, 3, 2>: 256 x 192, <4, 2, 2, 2>: 256 x 128 -- tile counts that fill the 256 CUs where 256^2 leaves a ragged last round;
//   <2, 4, 1, 2>, <2, 2, 2, 2>: 128 x 128 with eight / four waves -- the 640- and 1280-wide projections (M = 3072: 240 tiles).
//   (A four-wave 256^2 variant with 128 x 128 per wave and all 512 registers -- the library kernel's shape -- measured 5-10 % slower
//   than the eight-wave one under hipcc's schedule and was dropped.)
// Operands travel L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction, no staging registers); the LDS
// image is 128-byte rows with a 16-byte XOR swizzle applied on the per-lane SOURCE address (the DMA destination is lane-linear),
// read back conflict-free with ds_read_b128 (SQ_LDS_BANK_CONFLICT = 0).
// Schedule: ONE workgroup barrier per 64-deep K-tile.  Per 16-deep k-step the wave issues the fragment reads of step ks+1 and the
// MFMAs of step ks; after step 2 it waits for its own reads (the whole buffer is then consumed) and for its DMA pieces of the next
// tile, the workgroup meets once, and step 3 runs with the next tile's first fragments and the DMA pieces of tile t+2 (into the
// buffer that just became free) issued between its MFMAs -- the matrix pipe never waits for LDS or for a landing tile.  The two
// LDS buffers alternate by flipping one bit of the eight fragment offsets (the loop body is one tile).
// (Measured and dropped: phase-per-barrier schedules with two wave groups one barrier apart -- 8 barrier events per tile -- and a
// 3-deep ring of token buffers: all within 3 % of each other at 1130-1180 TF/s on 4096^3, the library kernel at 1400.)
// Epilogues (all in registers, one bf16 write): + bias, LayerNorm folded in front of the GEMM
//   LN(x) W^T = rstd (x (gamma o W)^T - mu rowsum(gamma o W)) + (beta W^T + b)    -- row statistics from the PRODUCER's epilogue
// (per-row partial sums of the residual stream written by the GEMM that produced it: `stats_out`), + residual, GEGLU
// (x * gelu(gate), weight rows interleaved per 64 at pack time so value and gate of a column sit in the same lane).
#include "cd360_common.h"
#include "cd360_tuning.h"
#include "cd360_prefetch.h"
#include "gemm4w_loop.inc"
#include <stdlib.h>
#include <type_traits>
#ifndef CD360_GEMM_SCHED
#define CD360_GEMM_SCHED 0  // probe builds (tools/probe/gemm_sched_ab.sh): 1 s_setprio around the MFMA runs, 2 static priority for waves NW/2 .., 4 reads interleaved with the MFMAs
#endif

namespace {

struct GemmParams {
  const uint16_t* a;    // [M, K] bf16, row stride lda
  const uint16_t* w;    // [N, K] bf16, row stride ldw (nn.Linear weight layout)
  uint16_t* out;        // [M, Nout] bf16, row stride ldo (Nout = N, or N / 2 with GEGLU)
  const float* bias;    // [N] fp32 or null (added after the LayerNorm fold)
  const uint16_t* res;  // [M, N] bf16 residual (row stride ldr) or null
  const float* ln_stats;  // [M, ln_parts, 2] fp32 partial (sum, sum of squares) of the A rows, or null = no LayerNorm fold
  const float* wsum;      // [N] fp32 row sums of w (only with ln_stats)
  float* stats_out;       // [M, ceil(N / BN), 2] fp32: per-row (sum, sumsq) of the bf16 outputs of each N tile, or null
  long lda, ldw, ldo, ldr;
  int M, N, K;
  int ln_parts, ln_dim;
  float ln_eps;
  int geglu;
  int tiles_m, tiles_n, group_m;
  // EPI 2: the projection output is the QUERY of a cross-attention over Nk <= 96 keys; it never leaves the registers
  const uint16_t* ak;  // K [B, >= Nk, heads*64] (element strides ak_sb, ak_sn; head h at columns 64 h)
  const uint16_t* av;  // V likewise (row-major: transposed by the LDS reads)
  long ak_sb, ak_sn, av_sb, av_sn;
  int a_nq, a_nk;      // queries per batch element (M = B * a_nq, a_nq % 256 == 0), keys
  float a_scale_log2e;
  // EPI 10 (BASELINE configs[4]): the two attention contractions on fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales) --
  // a_kv8 = the packed e4m3 image of K and V^T per (batch, head) (cd360_kv_pack_fp8: 14336 bytes each), a_kvs [B, heads, 2] fp32 =
  // the (K, V) tensor scales of that head; ak / av unused
  const unsigned char* a_kv8;
  const float* a_kvs;
  int a_heads;
  int a_dup_from, a_dup;  // query batch elements >= a_dup_from attend to TWO key / value sets (batch i and i + a_dup; output rows of
                          // batch i and i + a_dup): the de-duplicated CFG branch, whose q is projected once
  int wt;   // 1: the output tile leaves by write-through (sc1) buffer stores (set by the launcher: cd360_tuning.store_wt, outputs below 2 GiB)
  int abl;  // what-if timing knob (cd360_tuning.whatif, -DCD360_WHATIF probe builds only; results are wrong when set): 8 no DMA wait,
            // 16 no barrier, 32 no LDS wait, 4 no DMA, 64 no stores, 512 no fragment reads / MFMAs / epilogue (a pure L2 -> LDS streamer)
  // EPI 5: A is the implicit im2col matrix of a 3x3 / stride 1 / pad 1 convolution over a channels-last [images, H, W, Cin] tensor
  // (lda = Cin, K = 9 Cin in cd360_conv_k_order order: K-tile kt = (group * 9 + tap) * cv_kg + j reads channel chunk group * cv_kg + j
  // of the pixel shifted by the tap); out [M = images H W, N = Cout]
  int cv_H, cv_W, cv_kg;
  int cv_up;  // 1: nearest-neighbour 2x upsample folded into the convolution (Upsample.forward, openaimodel.py:114-181): the launch covers the
              // four output phases (a, b) = (row, column parity), phase slowest in the tile order; phase (a, b) is a 2 x 2-tap convolution
              // of the SOURCE image (taps dy = (t >> 1) + a - 1, dx = (t & 1) + b - 1; K = 4 Cin; weights w + phase * N * ldw, the 3 x 3
              // taps that read the same source pixel summed at pack time) whose row (n, i, j) is output pixel (n, 2 i + a, 2 j + b)
  const uint16_t* emb;  // [images, Cout] bf16 per-image addend (row stride emb_stride elements) or null
  long emb_stride;
  float* cstats;        // [M / (NMB 32), Cout, 2] fp32: per slab of NMB * 32 pixels and channel, (sum, sumsq) of the stored outputs, or null
#ifdef CD360_GEMM_STAMP
  uint32_t* stamp;      // probe builds (tools/probe/gemm_stamp.sh MODE): 1 = [workgroup][wave][K-tile][8] s_memtime stamps of the K loop, 2 = [workgroup][wave][8] phase stamps; or null
#endif
};

__device__ __forceinline__ int chan_pos(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
// erf-form GELU with erf from Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below the bf16 rounding of the result): one v_rcp,
// one v_exp and a degree-5 polynomial instead of libm's branchy erff -- the epilogue is exposed time (one workgroup per CU)
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = 1.f - poly * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);  // erf(|x| / sqrt 2)
  return 0.5f * x * (1.f + copysignf(e, x));
}

// the same on two values (v_pk_fma_f32 / v_pk_mul_f32 where the scalar form has fma / mul: identical roundings, half the instructions;
// the two transcendentals stay scalar) -- the GEGLU epilogue is VALU time on a CU with nothing else to run
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 z = ax * f32x2{0.70710678118654752f, 0.70710678118654752f};
  const f32x2 d = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, z, f32x2{1.f, 1.f});
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 poly = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
  poly = __builtin_elementwise_fma(poly, t, f32x2{1.421413741f, 1.421413741f});
  poly = __builtin_elementwise_fma(poly, t, f32x2{-0.284496736f, -0.284496736f});
  poly = __builtin_elementwise_fma(poly, t, f32x2{0.254829592f, 0.254829592f});
  const f32x2 a = (-z) * z * f32x2{1.4426950408889634f, 1.4426950408889634f};
  const f32x2 ex = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const f32x2 e = f32x2{1.f, 1.f} - poly * t * ex;
  const f32x2 se = {copysignf(e[0], x[0]), copysignf(e[1], x[1])};
  return f32x2{0.5f, 0.5f} * x * (f32x2{1.f, 1.f} + se);
}

#define LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define BARRIER()                  \
  do {                             \
    FENCE();                       \
    __builtin_amdgcn_s_barrier();  \
    FENCE();                       \
  } while (0)

template <int WM, int WN, int NCB, int NMB, int NBUF, int KS, int MV, int EPI>
__global__ __launch_bounds__(64 * (WM * WN * KS + MV)) void gemm_mfma_kernel(GemmParams p) {
  // MV > 0: MV extra "mover" waves issue ALL the LDS-DMA pieces and the WM x WN x KS others only read fragments and multiply.  On the
  // 128 x 128 tilings the CU's address path needs 0.35 us to take the 32 pieces of a K-tile and the fragment reads + MFMAs 0.40 us, but
  // with every wave doing both (and all of them meeting once per tile) a tile costs 0.6 us (tools/bench_gemm.py whatif): a wave that is
  // queueing pieces is not feeding the matrix pipe.  The movers queue them back to back; the hand-off stays what it was -- each mover
  // waits for its own pieces (counted vmcnt), then the workgroup barrier, which all waves take once per K-tile.
  // KS = 2: two groups of WM x WN waves share the tile, group kg multiplying k-steps 2 kg, 2 kg + 1 of every 64-deep K-tile (an
  // in-workgroup split of K): a wave then owns twice the output block for the same number of waves per SIMD -- a 128 x 128 tile is
  // eight waves of 64 x 64 on half the K-steps instead of eight of 64 x 32 on all of them, 64 KB instead of 96 KB of fragment reads per
  // K-tile (the LDS, shared with the DMA writes, is what the 128 x 128 tilings run out of) -- and the two partial accumulators meet
  // once, through the LDS, in front of the epilogue: each group hands the other one half of its channel blocks and finishes the half it
  // keeps, so all eight waves run the epilogue on 64 x 32 as in the unsplit tiling.
  static_assert(KS == 1 || KS == 2, "k-step groups");
  static_assert(KS == 1 || ((EPI == 0 || EPI == 5 || EPI == 6 || EPI == 11) && NCB % KS == 0), "split K: linear / convolution epilogues");
  constexpr int ECB = NCB / KS;                    // channel blocks a wave finishes in the epilogue
  constexpr int NWT = WM * WN;                     // waves of one k-step group (one output block each)
  constexpr int NWC = NWT * KS;                    // waves that multiply
  constexpr int NW = NWC + MV;                     // waves
  constexpr int NWD = MV ? MV : NW;                // waves that move data
  static_assert(MV == 0 || (EPI >= 0 && EPI <= 11), "mover waves: every epilogue (in the attention epilogues they also fetch K / V)");
  static_assert(EPI != 11 || (MV > 0 && NBUF == 4), "halo convolution: mover waves issue the pieces, four channel buffers");
  constexpr int KPW = 4 / KS;                      // k-steps of a K-tile that one wave multiplies
  constexpr int BM = WM * NMB * 32, BN = WN * NCB * 32;
  // bytes of one buffer of each operand (64-deep K-tile, 128-byte rows), rounded UP to whole DMA pieces: a tile side that is not a
  // multiple of the 8 * (moving waves) rows one piece covers (192 x 320 with six waves: 320 = 6.67 pieces) gets a last piece whose tail
  // rows -- past the operand's end: zeros; or the next tile's rows: never read -- land in padding rows of the buffer
  constexpr int PR_ = 8 * (MV ? MV : WM * WN * KS);
  constexpr uint32_t XB = ((BM + PR_ - 1) / PR_) * PR_ * 128, WB = ((BN + PR_ - 1) / PR_) * PR_ * 128;
  // epilogue: 0 linear, 1 GEGLU, 2 / 3 / 4 small-Nk attention on the projected tile with 2 / 4 / 6 groups of 16 keys, 7 / 8 / 9 the same
  // with 5 / 3 / 1 groups (the last 32-key block of scores is half used: 77 text keys are 5 groups, not 6)
  // EPI 6 = linear epilogue + the per-slab channel statistics of EPI 5 (a Linear whose output feeds a GroupNorm: SpatialTransformer.proj_out)
  // 10 = the 6-group attention epilogue with both contractions on fp8 MFMA (K / V pre-packed by cd360_kv_pack_fp8)
  constexpr bool F8 = EPI == 10;
  // EPI 11 = the convolution (EPI 5's epilogue) with the A operand read from a HALO image of the tile's input pixels (see `HALO` below)
  constexpr bool HALO = EPI == 11;
  constexpr bool GEGLU = EPI == 1, ATTN = (EPI >= 2 && EPI <= 4) || (EPI >= 7 && EPI <= 10), CONV = EPI == 5 || HALO, CSTATS = EPI == 5 || EPI == 6 || HALO;
  // ASM4 (round 6): 256 x 256 as FOUR waves of 128 x 128, one per SIMD, on the generated instruction stream of gemm4w_loop.inc
  // (tools/gen_gemm4w_loop.py): operands global -> registers -> LDS with one tile of slack per piece instead of LDS-DMA.
  constexpr bool ASM4 = WM == 2 && WN == 2 && NCB == 4 && NMB == 4 && NBUF == 2 && KS == 1 && MV == 0 && (EPI == 0 || EPI == 1);
  // LDS map: NBUF token buffers, then NBUF channel buffers.  The attention epilogues interleave them instead (buffer b = tokens, then
  // channels, at b * (XB + WB)) and rotate the ring so that the LAST K-tile sits in buffer 0: everything behind buffer 0 is then free
  // one tile before the loop ends, and the K / V rows of the tile's heads are fetched into it under the last K-tile's MFMAs.
  constexpr uint32_t BB = XB + WB;
  constexpr uint32_t XSTR = ATTN ? BB : XB, WSTR = ATTN ? BB : WB;  // byte distance between consecutive buffers of one operand
  constexpr int PR = 8 * NWD;                      // rows one DMA piece of all moving waves covers (8 per wave)
  // HALO (round 6): the nine taps of a 3 x 3 convolution read the SAME input pixels nine times -- as nine K-tiles of the implicit im2col
  // matrix that is nine trips through the L2 -> LDS path, which is what the 128 x 128 tilings run out of.  Here the
  // token operand of a 64-channel chunk is ONE image in the LDS: the tile's BM / W image rows plus one row above / below and one
  // pixel left / right ((BM / W + 2) (W + 2) rows of 128 bytes, zeros outside the image), fetched once per chunk; the nine taps are nine
  // row-shifted views of it (fragment row = pixel + dy (W + 2) + dx).  K runs chunk-major here (all nine taps of a chunk back to back --
  // the weights stay in cd360_conv_k_order's order, K-tile (chunk, tap) is simply read from where that order puts it); two halo
  // buffers alternate by chunk, the channel operand keeps its ring.  Per chunk the path carries 26-36 KB of pixels instead of 9 x 16.
  constexpr int HPMAX = 9;                          // pieces (of PR rows) of one halo image, at most
  constexpr uint32_t HB = HALO ? HPMAX * PR * 128 : 0;  // bytes of one halo buffer
  constexpr uint32_t XREG = 0, WREG = HALO ? 2 * HB : (ATTN ? XB : NBUF * XB);
  constexpr int XP = HALO ? 2 : (BM + PR - 1) / PR, WP = (BN + PR - 1) / PR;  // pieces per wave per K-tile (HALO: up to two halo pieces ride with a K-tile)
  constexpr int NP = XP + WP, NMMA = NCB * NMB;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  // (LDS-DMA destinations of the attention epilogues are formed in address space 3 from here: a generic pointer that reaches the cast
  // through a select makes hipcc 7.2 emit an illegal "V_CMP_NE_U32 0, src_shared_base")
  __attribute__((address_space(3))) unsigned char* const lds3 = (__attribute__((address_space(3))) unsigned char*)lds;

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mover = MV > 0 && wave >= NWC;
#if defined(CD360_GEMM_STAMP) && CD360_GEMM_STAMP == 2  // phase stamps of a launch (100 MHz counter): [workgroup][wave][8] at p.stamp, lane 0 of every wave (tools/probe/gemm4w_stamp.py)
#define G4STAMP(i)                                                                                                            \
  do {                                                                                                                        \
    if constexpr (!ATTN) {                                                                                                    \
      if (p.stamp && (tid & 63) == 0) p.stamp[((long)blockIdx.x * NW + wave) * 8 + (i)] = (uint32_t)__builtin_amdgcn_s_memrealtime(); \
    }                                                                                                                         \
  } while (0)
#else
#define G4STAMP(i) do { } while (0)
#endif
  G4STAMP(0);
  const int dwave = MV ? (mover ? wave - NWC : 0) : wave;  // index among the moving waves
  const int kg = mover ? 0 : wave / NWT, wv = mover ? 0 : wave - kg * NWT;  // k-step group, wave inside it
  const int wr = wv / WN, wc = wv % WN;      // token / channel position of the wave in the tile

  // ---- tile of this workgroup: XCD-contiguous ranges, group_m token tiles per group with the channel tile varying slowest ----
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  int up_a = 0, up_b = 0, up_phase = 0;
  if constexpr (EPI == 5) {
    if (p.cv_up) {
      const int per_phase = p.tiles_m * p.tiles_n;
      up_phase = tile / per_phase;
      tile -= up_phase * per_phase;
      up_a = up_phase >> 1;
      up_b = up_phase & 1;
    }
  }
  const int per_group = p.group_m * p.tiles_n;
  const int grp = tile / per_group, in_grp = tile - grp * per_group;
  const int first_m = grp * p.group_m;
  const int gm = (p.tiles_m - first_m) < p.group_m ? (p.tiles_m - first_m) : p.group_m;
  const int tm = first_m + in_grp % gm, tn = in_grp / gm;
  const long m0 = (long)tm * BM;
  const int n0 = tn * BN;

  // ---- LDS-DMA geometry: piece j covers rows j*PR + wave*8 + lane/8 of the operand tile; LDS chunk lane%8 <- source chunk ^ swizzle.
  // Rows past M / N are past the end of the buffer descriptor: the hardware returns zeros (they only feed masked outputs).
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)(CONV ? (long)p.M * p.lda * 2 : ((long)p.M - 1) * p.lda * 2 + (long)p.K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (long)up_phase * p.N * p.ldw), 0, (int)(((long)p.N - 1) * p.ldw * 2 + (long)p.K * 2), 0x00020000);
  const int srow = dwave * 8 + (lane >> 3);
  const int schunk = (lane & 7) ^ ((srow >> 1) & 7);
  const uint32_t xoff0 = (uint32_t)((m0 + srow) * p.lda * 2 + schunk * 16);
  const uint32_t woff0 = (uint32_t)(((long)n0 + srow) * p.ldw * 2 + schunk * 16);
  const uint32_t xstep = (uint32_t)(PR * p.lda * 2), wstep = (uint32_t)(PR * p.ldw * 2);
  unsigned char* const dma_base = lds + dwave * 1024;
  // Convolution: bit `tap` of xmask[i] = the pixel of this lane's row of piece i has an in-image neighbour under that tap (rows past M:
  // none).  The shifted pixel is the same row offset plus a wave-uniform tap offset; a padding neighbour reads from an offset past the
  // end of the buffer descriptor, i.e. zeros.
  uint32_t xmask[(CONV && !HALO) ? XP : 1];
  if constexpr (CONV && !HALO) {
    const int hw = p.cv_H * p.cv_W;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const long m = m0 + i * PR + srow;
      uint32_t mk = 0;
      if (m < p.M) {
        const int rem = (int)(m % hw), y = rem / p.cv_W, x = rem - y * p.cv_W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int yy = p.cv_up ? y + (tap >> 1) + up_a - 1 : y + tap / 3 - 1, xx = p.cv_up ? x + (tap & 1) + up_b - 1 : x + tap % 3 - 1;
          if (yy >= 0 && yy < p.cv_H && xx >= 0 && xx < p.cv_W && (!p.cv_up || tap < 4)) mk |= 1u << tap;
        }
      }
      xmask[i] = mk;
    }
  }
  int cv_j = 0, cv_tap = 0, cv_cg = 0;  // K-tile about to be issued (tiles are issued in order)
  int is_tap = 0;
  uint32_t is_tapoff = 0;
  auto conv_next = [&]() {  // wave-uniform: byte offset of the next K-tile's (tap shift, channel chunk) relative to the output pixel's row
    is_tap = cv_tap;
    const int dy = p.cv_up ? (cv_tap >> 1) + up_a : cv_tap / 3, dx = p.cv_up ? (cv_tap & 1) + up_b : cv_tap - 3 * (cv_tap / 3);
    is_tapoff = (uint32_t)((((dy - 1) * p.cv_W + (dx - 1)) * (int)p.lda + (cv_cg * p.cv_kg + cv_j) * 64) * 2);
    if (++cv_j == p.cv_kg) {
      cv_j = 0;
      if (++cv_tap == (p.cv_up ? 4 : 9)) {
        cv_tap = 0;
        ++cv_cg;
      }
    }
  };
  // HALO: geometry of the tile's halo image and the chunk-major walk of the K-tiles.  Issue side (tiles are issued in order): is_hc / is_ht =
  // chunk / tap of the K-tile about to be issued, is_wcol = byte column of its weights in cd360_conv_k_order's layout.
  const int hW2 = HALO ? p.cv_W + 2 : 1;                                  // row pitch of the halo image in pixels
  const int hHR = HALO ? (BM / (HALO ? p.cv_W : 1) + 2) * hW2 : 0;         // its rows
  const int hNC = HALO ? (int)(p.lda >> 6) : 0;                            // 64-channel chunks
  int h_img0 = 0;                                                          // (image index * H + first image row of the tile - 1) -- row of halo row 0
  if constexpr (HALO) {
    const int hw = p.cv_H * p.cv_W;
    h_img0 = (int)(m0 / hw) * p.cv_H + (int)(m0 % hw) / p.cv_W - 1;
  }
  const int h_ylo = HALO ? (int)(m0 / (p.cv_H * p.cv_W)) * p.cv_H : 0;     // first / one-past-last image row index (image included) that exists
  const int h_yhi = h_ylo + (HALO ? p.cv_H : 0);
  const uint32_t h_inv = HALO ? (uint32_t)((1u << 20) / (uint32_t)hW2 + 1u) : 0u;  // row / hW2 = (row * h_inv) >> 20 for row < 1024 (checked by the launcher)
  int is_hc = 0, is_ht = 0, is_hg = 0, is_hj = 0;
  uint32_t is_wcol = 0;
  auto halo_next = [&]() {  // wave-uniform
    is_hc = is_hg * p.cv_kg + is_hj;
    is_ht = cv_tap;
    is_wcol = (uint32_t)((((is_hg * 9 + cv_tap) * p.cv_kg + is_hj) * 64) * 2);
    if (++cv_tap == 9) {
      cv_tap = 0;
      if (++is_hj == p.cv_kg) {
        is_hj = 0;
        ++is_hg;
      }
    }
  };
  // halo piece h (PR rows from row h PR) of chunk c into halo buffer c & 1: this lane's row = h PR + srow, its pixel = (h_img0 + row / hW2,
  // row % hW2 - 1); a pixel outside the image (or a row past the image's end) is fetched from past the end of the descriptor: zeros
  auto halo_piece = [&](int c, int h) {
    const int row = h * PR + srow;
    const int hy = (int)(((uint32_t)row * h_inv) >> 20), hx = row - hy * hW2 - 1;
    const int yy = h_img0 + hy;
    uint32_t o = (uint32_t)(((long)yy * p.cv_W + hx) * p.lda * 2 + c * 128 + schunk * 16);
    if (row >= hHR || yy < h_ylo || yy >= h_yhi || hx < 0 || hx >= p.cv_W) o = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, LDS_AS3(dma_base + (c & 1) * HB + h * (PR * 128)), 16, o, 0, 0, 0);
  };
  const int hPM = HALO ? (hHR + PR - 1) / PR : 0;                          // pieces of a halo image
  // DMA piece i (0 .. NP-1) of K-tile kt into the buffers at byte offsets bx / bw (0 | one buffer).  The wave-uniform part of the
  // source offset is added with an opaque v_add (otherwise the compiler keeps NP strength-reduced per-piece offsets live in VGPRs).
  auto piece = [&](int kt, int i, uint32_t bx, uint32_t bw) {
    uint32_t o;
    if constexpr (HALO) {
      if (i < XP) {
        // up to two pieces of the NEXT chunk's halo ride with the K-tiles of taps NBUF .. 8 of this chunk: issued after the barrier that
        // ended the chunk before (whose buffer they overwrite), landed -- they are older than this chunk's last channel pieces -- before
        // the first K-tile of the next chunk is waited for
        const int h = 2 * (is_ht - NBUF) + i;
        if (is_ht >= NBUF && h < hPM && is_hc + 1 < hNC) halo_piece(is_hc + 1, h);
        return;
      }
    }
    if (i < XP) {
      const uint32_t su = (CONV ? is_tapoff : (uint32_t)(kt * 128)) + (uint32_t)i * xstep;
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(o) : "s"(su), "v"(xoff0));
      if constexpr (CONV) o = ((xmask[i < XP ? i : 0] >> is_tap) & 1u) ? o : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, LDS_AS3(dma_base + XREG + bx + i * (PR * 128)), 16, o, 0, 0, 0);
    } else {
      const uint32_t su = (HALO ? is_wcol : (uint32_t)(kt * 128)) + (uint32_t)(i - XP) * wstep;
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(o) : "s"(su), "v"(woff0));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_AS3(dma_base + WREG + bw + (i - XP) * (PR * 128)), 16, o, 0, 0, 0);
    }
  };

  // ---- fragment read geometry: per lane one byte offset per k-step and operand (current buffer); blocks are immediates ----
  const int cp = chan_pos(l31);
  uint32_t xo[KPW], wo[KPW];
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    const int ks = kg * KPW + i;
    xo[i] = XREG + (uint32_t)((wr * NMB * 32 + l31) * 128 + (((2 * ks + hh) ^ ((l31 >> 1) & 7)) << 4));
    wo[i] = WREG + (uint32_t)((wc * NCB * 32 + cp) * 128 + (((2 * ks + hh) ^ ((cp >> 1) & 7)) << 4));
  }
  // HALO: the token fragments of K-tile (chunk, tap) are rows pixel + dy (W + 2) + dx of halo buffer chunk & 1: per-lane byte offsets
  // xh[k-step][block], recomputed for every K-tile (the 16-byte XOR swizzle follows the ROW, which moves with the tap)
  uint32_t xh[HALO ? KPW : 1][HALO ? NMB : 1];
  int hb_[HALO ? NMB : 1];  // halo row of the lane's pixel of block mb under tap (0, 0)
  if constexpr (HALO) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const int px = wr * (NMB * 32) + mb * 32 + l31;
      hb_[mb] = (px / p.cv_W) * hW2 + px % p.cv_W;
    }
  }
  int c_tap = 0, c_par = 0;  // compute side: tap and chunk parity of the K-tile whose fragments are read next
  auto halo_offsets = [&]() {
    if constexpr (HALO) {
      const int shift = (c_tap / 3) * hW2 + c_tap % 3;  // wave-uniform
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        const int row = hb_[mb] + shift;
        const uint32_t base = (uint32_t)(c_par * HB) + (uint32_t)row * 128u, swz = (uint32_t)(row >> 1) & 7u;
#pragma unroll
        for (int i = 0; i < KPW; ++i) xh[i][mb] = base + ((((uint32_t)(2 * (kg * KPW + i) + hh)) ^ swz) << 4);
      }
      if (++c_tap == 9) {
        c_tap = 0;
        c_par ^= 1;
      }
    }
  };
  bf16x8 fx[2][NMB], fw[2][NCB];
  auto read_ks = [&](int set, int ks) {
#pragma unroll
    for (int nb = 0; nb < NCB; ++nb) fw[set][nb] = *reinterpret_cast<const bf16x8*>(lds + wo[ks] + nb * 4096);
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      if constexpr (HALO) fx[set][mb] = *reinterpret_cast<const bf16x8*>(lds + xh[ks][mb]);
      else fx[set][mb] = *reinterpret_cast<const bf16x8*>(lds + xo[ks] + mb * 4096);
    }
  };

  f32x16 acc[NCB][NMB];
#pragma unroll
  for (int nb = 0; nb < NCB; ++nb)
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][mb][i] = 0.f;

  const int nk = p.K >> 6;
  // attention epilogues: K-tile t lives in buffer (t + rot) % NBUF, so that the last one lands in buffer 0
  const int rot = ATTN ? (NBUF - ((nk - 1) % NBUF)) % NBUF : 0;
  if constexpr (ATTN) {
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      xo[i] += rot * XSTR;
      wo[i] += rot * WSTR;
    }
  }
#ifdef CD360_WHATIF
  const int abl = p.abl;
#else
  constexpr int abl = 0;  // product build: the what-if branches below fold away
#endif
  // a wave whose channels are all beyond N (last channel tile of N = 640 = 2.5 tiles) moves data and synchronises but does not multiply
  const bool has_ch = n0 + wc * (NCB * 32) < p.N && !(abl & 512) && !mover;  // (what-if bit 512: every wave only moves data)

  // ---- attention epilogues: K and V rows of the tile's WN heads -> LDS by LDS-DMA, no staging registers ----------------------------
  // Per head KROWS rows of K then KROWS rows of V, 128 bytes each (K with the 16-byte XOR swizzle of its fragment reads, V with the 32-byte
  // one of the transposing reads, both applied on the per-lane SOURCE address), starting at the second ring buffer.  Rows >= Nk lie past
  // the end of the buffer descriptor and arrive as zeros.  A piece is 8 rows (1 KiB); the moving waves take pieces round-robin.
  constexpr int NK16 = (EPI == 4 || EPI == 10) ? 6 : EPI == 2 ? 2 : EPI == 3 ? 4 : EPI == 7 ? 5 : EPI == 8 ? 3 : 1;
  constexpr int F8_HEAD = 96 * 64 + 64 * 128;  // fp8: 96 key rows of 64 bytes, then V^T as 64 channel rows of 128 key slots
  constexpr int NKB = (NK16 + 1) / 2, KROWS = NK16 * 16, HEAD_LDS = F8 ? F8_HEAD : 2 * KROWS * 128;
  constexpr bool HALF = (NK16 & 1) != 0;  // the last 32-key block of scores is used in its first 16 keys only
  constexpr uint32_t KVBASE = BB;
  // the tile's slices of bias and wsum (BN floats each) travel with K / V when the LDS has the 2 KiB left: the LayerNorm fold then
  // reads them with two ds_read_b128 per 8 channels instead of four dependent L2 round trips in front of the first attention MFMA
  constexpr uint32_t BWBASE = KVBASE + WN * HEAD_LDS;
  constexpr bool BW_LDS = ATTN && BN * 4 <= 1024 && BWBASE + 2048 <= 160 * 1024;
  static_assert(!ATTN || (NWC * 4096 <= (int)BB), "attention epilogue: the waves' output blocks alias ring buffer 0");
  const int a_bidx = ATTN ? (int)(m0 / p.a_nq) : 0;  // batch element of this token tile
  auto kv_issue = [&](int kvb) {
    if constexpr (F8) {  // the packed image is copied as it lies: 14 pieces per head; heads past the last one are out of range (zeros)
      const int h0 = n0 >> 6, nh = p.a_heads - h0 < WN ? p.a_heads - h0 : WN;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(p.a_kv8 + ((long)kvb * p.a_heads + h0) * F8_HEAD), 0, nh * F8_HEAD, 0x00020000);
      constexpr int NPKV = WN * (F8_HEAD / 1024);
#pragma unroll
      for (int i = 0; i < (NPKV + NWD - 1) / NWD; ++i) {
        const int q = i * NWD + dwave;
        if (q < NPKV) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds3 + (KVBASE + q * 1024), 16, (uint32_t)(q * 1024 + lane * 16), 0, 0, 0);
      }
    } else if constexpr (ATTN) {
      const int kbytes = (int)((((long)p.a_nk - 1) * p.ak_sn + p.N) * 2), vbytes = (int)((((long)p.a_nk - 1) * p.av_sn + p.N) * 2);
      const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ak + (long)kvb * p.ak_sb), 0, kbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.av + (long)kvb * p.av_sb), 0, vbytes, 0x00020000);
      constexpr int PPH = 2 * KROWS / 8, NPKV = WN * PPH;  // pieces per head (K, then V), pieces in all
      const int lrow8 = lane >> 3, lch = lane & 7;
#pragma unroll
      for (int i = 0; i < (NPKV + NWD - 1) / NWD; ++i) {
        const int q = i * NWD + dwave;  // wave-uniform
        if (q < NPKV) {
          const int hl = q / PPH, rem = q - hl * PPH;
          const bool isv = rem >= KROWS / 8;
          const int key = (isv ? rem - KROWS / 8 : rem) * 8 + lrow8;
          const int sch = isv ? (lch ^ (((key >> 1) & 3) << 1)) : (lch ^ ((key >> 1) & 7));
          const int col = n0 + hl * 64;
          uint32_t off = (uint32_t)key * (uint32_t)(isv ? p.av_sn : p.ak_sn) * 2u + (uint32_t)(col * 2 + sch * 16);
          if (col >= p.N) off = 0x80000000u;
          if (isv) __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, lds3 + (KVBASE + q * 1024), 16, off, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, lds3 + (KVBASE + q * 1024), 16, off, 0, 0, 0);
        }
      }
    }
  };
  auto bw_issue = [&]() {  // once per tile, by the first two moving waves
    if constexpr (BW_LDS) {
      const uint32_t off = (lane * 4 < BN && n0 + lane * 4 < p.N) ? (uint32_t)((n0 + lane * 4) * 4) : 0x80000000u;
      if (p.bias && dwave == 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.N * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds3 + BWBASE, 16, off, 0, 0, 0);
      }
      if (p.ln_stats && dwave == (NWD > 1 ? 1 : 0)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wsum, 0, p.N * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds3 + (BWBASE + 1024), 16, off, 0, 0, 0);
      }
    }
  };

  // Linear / convolution epilogues (round 6): the tile's slices of bias and wsum travel by LDS-DMA too -- into 2 .. 4 KB behind the ring and
  // the staging image, requested BEFORE the first operand piece (older than every piece: the counted waits still count pieces only) --
  // and the epilogue reads them with ds_read_b128.  They were one dependent L2 round trip per 8-channel slice of the epilogue, in front
  // of its arithmetic, on a CU with nothing else to run (tools/probe/gemm4w_ab.py ksweep: 24 us of a FF1 launch did not depend on K;
  // the library's 14 are the 63 MB it writes).  The four-wave arrangement's ring is all of the LDS: it fetches them behind the loop.
  constexpr uint32_t RINGK = ASM4 ? 5u * 32768u : (HALO ? 2 * HB + NBUF * WB : NBUF * (XB + WB));
  constexpr uint32_t STAGEK = BM * BN * 2 + (KS == 2 ? BM * BN * 4 : 0);
  constexpr uint32_t LBW = ASM4 ? STAGEK : (RINGK > STAGEK ? RINGK : STAGEK);
  constexpr uint32_t LBW_SLICE = ((BN * 4 + 1023) / 1024) * 1024;
  constexpr bool LIN_BW = !ATTN && LBW + (ASM4 ? 4096u : 2 * LBW_SLICE) <= 160 * 1024;
  auto lin_bw_issue = [&]() {
    if constexpr (LIN_BW) {
#pragma unroll
      for (int q = 0; q < (int)(LBW_SLICE / 1024); ++q) {
        const int c = q * 256 + lane * 4;
        const uint32_t off = (c < BN && n0 + c < p.N) ? (uint32_t)((n0 + c) * 4) : 0x80000000u;
        if (p.bias && dwave == 0) {
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.N * 4, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds3 + (LBW + q * 1024), 16, off, 0, 0, 0);
        }
        if (p.ln_stats && dwave == (NWD > 1 ? 1 : 0)) {
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wsum, 0, p.N * 4, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds3 + (LBW + LBW_SLICE + q * 1024), 16, off, 0, 0, 0);
        }
      }
    }
  };

  // attention epilogues: the rows' LayerNorm constants (q = acc * c1 + (c2 * wsum + bias), c1 = rstd, c2 = -rstd * mean) are requested
  // here, ahead of the first operand pieces (older than every piece: the counted waits below still count pieces only), so that their
  // L2 round trip is over long before the epilogue wants them
  float a_c1[ATTN ? NMB : 1], a_c2[ATTN ? NMB : 1];
  f32x2 a_st[ATTN ? NMB : 1];
  if constexpr (ATTN) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      a_st[mb] = f32x2{0.f, 0.f};
      if (p.ln_stats && !mover) {
        const f32x2* st = reinterpret_cast<const f32x2*>(p.ln_stats) + (m0 + wr * (NMB * 32) + l31 + mb * 32) * p.ln_parts;
        for (int q = 0; q < p.ln_parts; ++q) a_st[mb] += st[q];
      }
    }
  }

  // ---- prologue: tiles 0 .. NBUF-1 in flight (one per buffer), tile 0 landed ----
  // counted wait: everything but the `later` most recently issued tiles has landed (s_waitcnt takes an immediate)
  auto wait_tiles_in_flight = [&](int later) {
    if (abl & 8) return;
    // (HALO: a K-tile carries WP channel pieces and 0 .. 2 halo pieces; counting the guaranteed WP per younger tile waits for at most two
    // pieces more than necessary and never for fewer)
    constexpr int NPG = HALO ? WP : NP;
    if (later <= 0) WAIT_VM0();
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPG) : "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPG) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NPG) : "memory");
  };
  // LayerNorm fold: the rows' mean / rstd from the producer's partial sums -- requested HERE, ahead of the K loop (round 6: they were a
  // dependent L2 round trip at the head of the epilogue), two registers per 32-token block across the loop
  float mu[ATTN ? 1 : NMB], rs[ATTN ? 1 : NMB];
  if constexpr (!ATTN) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) { mu[mb] = 0.f; rs[mb] = 1.f; }
    if (p.ln_stats && !mover) {
      const float inv = 1.f / (float)p.ln_dim;
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        const long m = m0 + wr * (NMB * 32) + l31 + mb * 32;
        float s_ = 0.f, ss = 0.f;
        if (m < p.M) {
          const f32x2* st = reinterpret_cast<const f32x2*>(p.ln_stats) + m * p.ln_parts;
          for (int q = 0; q < p.ln_parts; ++q) {
            const f32x2 v = st[q];
            s_ += v[0];
            ss += v[1];
          }
        }
        const float mean = s_ * inv;
        const float var = fmaxf(ss * inv - mean * mean, 0.f);
        mu[mb] = mean;
        rs[mb] = rsqrtf(var + p.ln_eps);
      }
    }
  }
  static_assert(NBUF >= 2 && NBUF <= 4 && (NBUF - 1) * NP <= 63, "counted waits: up to 3 tiles, 6-bit vmcnt");
  if constexpr (!ASM4) {
  if (!MV || mover) {
    lin_bw_issue();
    if constexpr (HALO) {  // the first chunk's halo image: older than every channel piece, so the first counted wait covers it
      for (int h = 0; h < hPM; ++h) halo_piece(0, h);
    }
#pragma unroll
    for (int b = 0; b < NBUF; ++b)
      if (b < nk) {
        if constexpr (HALO) halo_next();
        else if constexpr (CONV) conv_next();
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(b, i, ((b + rot) % NBUF) * XSTR, ((b + rot) % NBUF) * WSTR);
      }
    if constexpr (ATTN) {
      if (nk == 1) {  // (a one-tile loop never reaches the issue point below; buffers 1 .. are idle from the start)
        kv_issue(a_bidx);
        bw_issue();
      }
    }
    wait_tiles_in_flight((nk < NBUF ? nk : NBUF) - 1);
  }
  BARRIER();
  halo_offsets();
  read_ks(0, 0);
  }  // !ASM4
  if constexpr (ATTN) {
    const float inv = p.ln_stats ? 1.f / (float)p.ln_dim : 0.f;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const float mean = a_st[mb][0] * inv;
      const float rstd = p.ln_stats ? rsqrtf(fmaxf(a_st[mb][1] * inv - mean * mean, 0.f) + p.ln_eps) : 1.f;
      a_c1[mb] = rstd;
      a_c2[mb] = -rstd * mean;
    }
  }

  // the loop exists twice: waves with output channels multiply, the others only move data
  auto k_loop = [&](auto mul_tag, auto move_tag) {
    constexpr bool MUL = decltype(mul_tag)::value, MOVE = decltype(move_tag)::value;
    auto mma_ks = [&](int set) {
      if constexpr (MUL) {
#pragma unroll
        for (int i = 0; i < NMMA; ++i)
          acc[i % NCB][i / NCB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][i % NCB], fx[set][i / NCB], acc[i % NCB][i / NCB], 0, 0, 0);
      }
    };
    // The NP pieces that refill a released buffer are SPREAD over the MFMAs that follow instead of all of them in the last k-step: the
    // CU's one address path takes ~26 cycles per 1-KiB piece (tools/probe/dma_probe.hip: 47-49 B/clk/CU), i.e. 1700 cycles for the 64
    // pieces of a 256 x 256 tile -- queued inside ONE k-step (512 cycles of MFMAs per SIMD) the matrix pipe then idles behind the queue.
    // With three or more buffers the pieces go behind a whole tile's worth of MFMAs (the first share in this tile's last k-step, as soon
    // as the barrier has released the buffer, the rest in the next tile's k-steps 0 .. 2); with TWO buffers (round 4) the refilled tile
    // is already needed at the NEXT barrier, so the shares stop one k-step short of it (last k-step, then k-steps 0 and 1: their pieces
    // have a k-step or two of MFMAs to land in, L2-warm latency 250-400 cycles).  The counted waits are unchanged (every piece of tile
    // t+NBUF-1 is still issued before tile t's wait, none of tile t+NBUF).  -DCD360_NO_SPREAD2 builds the two-buffer loop of round 3 (A/B).
#ifdef CD360_NO_SPREAD2
    constexpr bool spread2 = NBUF >= 3;
#else
    constexpr bool spread2 = true;
#endif
    constexpr bool SPREAD = spread2;
    constexpr int NSLOT = (NBUF >= 3 ? KPW : (KPW > 1 ? KPW - 1 : 1)) * NMMA;
    auto slot_pieces = [&](int j, int i, int tile, uint32_t obx, uint32_t obw) {  // pieces of `tile` that go behind MFMA i of k-step group j
#pragma unroll
      for (int q = 0; q < NP; ++q)
        if ((q * NSLOT / NP) / NMMA == j && (q * NSLOT / NP) % NMMA == i) piece(tile, q, obx, obw);
    };
    uint32_t bx = rot * XSTR, bw = rot * WSTR;  // buffers of tile t
    uint32_t pbx = 0, pbw = 0;  // buffers of tile t-1 (being refilled with tile t-1+NBUF while tile t is multiplied)
    int bnext = (rot + 1) % NBUF;  // index of the buffer of tile t+1
    for (int t = 0; t < nk; ++t) {
      // fences pin the order "reads of k-step ks+1, then the MFMAs of ks": the compiler otherwise sinks the reads to the end of the
      // MFMA run (exposing the LDS latency) or hoists later k-steps' reads (spilling)
      const bool prev_more = MOVE && spread2 && t >= 1 && t - 1 + NBUF < nk && !(abl & 4);
#ifdef CD360_GEMM_STAMP
      const uint64_t stA = __builtin_amdgcn_s_memtime();
      uint64_t stB = stA, stD = stA, stC1 = stA, stC2 = stA;
#endif
#pragma unroll
      for (int i = 0; i + 1 < KPW; ++i) {  // (the wave's last k-step, fragment set 1, runs below with the DMA issue)
        read_ks((i + 1) & 1, i + 1);
#if !(CD360_GEMM_SCHED & 4)
        FENCE();
#endif
#if CD360_GEMM_SCHED & 1
        if constexpr (MUL) __builtin_amdgcn_s_setprio(1);
#endif
        if constexpr (SPREAD) {
#pragma unroll
          for (int m = 0; m < NMMA; ++m) {
            if constexpr (MUL)
              acc[m % NCB][m / NCB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i & 1][m % NCB], fx[i & 1][m / NCB], acc[m % NCB][m / NCB], 0, 0, 0);
            if (prev_more) slot_pieces(i + 1, m, t - 1 + NBUF, pbx, pbw);
          }
        } else {
          mma_ks(i & 1);
        }
#if CD360_GEMM_SCHED & 4
        // (probe: the fragment reads of the next k-step interleaved one by one with this k-step's MFMAs instead of issued ahead of them)
        if constexpr (MUL) {
#pragma unroll
          for (int m = 0; m < NMMA; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (m < NCB + NMB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
#endif
#if CD360_GEMM_SCHED & 1
        if constexpr (MUL) __builtin_amdgcn_s_setprio(0);
#endif
        FENCE();
      }
      if (t + 1 < nk) {
#ifdef CD360_GEMM_STAMP
        stB = __builtin_amdgcn_s_memtime();
#endif
        if (!(abl & 32)) WAIT_LGKM0();  // this wave's reads of the buffer are done ...
#ifdef CD360_GEMM_STAMP
        stC1 = __builtin_amdgcn_s_memtime();
#endif
        // ... and its pieces of tile t+1 have landed (issued NBUF-1 tiles ago; tiles t+2 .. t+NBUF-1 may still be in flight)
        if constexpr (MOVE) wait_tiles_in_flight((nk - 1 < t + NBUF - 1 ? nk - 1 : t + NBUF - 1) - (t + 1));
        FENCE();
#ifdef CD360_GEMM_STAMP
        stC2 = __builtin_amdgcn_s_memtime();
#endif
        if (!(abl & 16)) __builtin_amdgcn_s_barrier();
        FENCE();
#ifdef CD360_GEMM_STAMP
        stD = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t ax = bnext == 0 ? 0u - (NBUF - 1) * XSTR : XSTR, aw = bnext == 0 ? 0u - (NBUF - 1) * WSTR : WSTR;
#pragma unroll
        for (int ks = 0; ks < KPW; ++ks) {
          xo[ks] += ax;
          wo[ks] += aw;
        }
        halo_offsets();  // (every read of tile t has been issued: the offsets move on to tile t + 1)
        read_ks(0, 0);
        FENCE();
      }
      {  // last k-step: its MFMAs with the DMA pieces of tile t+NBUF (into the buffer just released) spread between them.  The MFMAs
         // are unconditional code: accumulators defined in two branch arms make the register allocator copy and spill them.
        const bool more = MOVE && t + NBUF < nk && !(abl & 4);
        if constexpr (HALO) {
          if (more) halo_next();
        } else if constexpr (CONV) {
          if (more) conv_next();
        }
#if CD360_GEMM_SCHED & 1
        if constexpr (MUL) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int i = 0; i < NMMA; ++i) {
          if constexpr (MUL)
            acc[i % NCB][i / NCB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[1][i % NCB], fx[1][i / NCB], acc[i % NCB][i / NCB], 0, 0, 0);
          if (more) {
            if constexpr (SPREAD) {
              slot_pieces(0, i, t + NBUF, bx, bw);
            } else {  // pieces [i NP / NMMA, (i+1) NP / NMMA): all NP of them, evenly spread over the MFMAs
#pragma unroll
              for (int q = i * NP / NMMA; q < (i + 1) * NP / NMMA; ++q) piece(t + NBUF, q, bx, bw);
            }
          }
        }
#if CD360_GEMM_SCHED & 1
        if constexpr (MUL) __builtin_amdgcn_s_setprio(0);
#endif
      }
      if constexpr (ATTN && MOVE) {
        // the barrier of this iteration released every buffer but the last tile's (buffer 0): K / V of the tile's heads travel under
        // the last K-tile's MFMAs (no operand piece is issued any more, the address path is idle)
        if (t == nk - 2) {
          kv_issue(a_bidx);
          bw_issue();
        }
      }
      FENCE();
#ifdef CD360_GEMM_STAMP
      if (CD360_GEMM_STAMP == 1 && p.stamp && t < 64 && wave < NWC) {  // stamps parked in the LDS behind the ring (a global store would count in vmcnt); lane 0 of each wave
        const uint64_t stE = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
          uint32_t* d = reinterpret_cast<uint32_t*>(lds + NBUF * (XB + WB)) + (wave * 64 + t) * 8;
          d[0] = (uint32_t)stA; d[1] = (uint32_t)stB; d[2] = (uint32_t)stD; d[3] = (uint32_t)stE; d[4] = (uint32_t)stC1; d[5] = (uint32_t)stC2;
        }
      }
#endif
      pbx = bx;
      pbw = bw;
      bx = bnext * XSTR;
      bw = bnext * WSTR;
      bnext = bnext + 1 == NBUF ? 0 : bnext + 1;
    }
  };
  // ======================================= epilogue =======================================
  // lane: token row (l31) of each 32-token block; registers: 16 consecutive channels 16*hh + r of each 32-channel block
  const int mrow0 = wr * (NMB * 32) + l31;  // + mb * 32 : row inside the tile
  float rsum[NMB], rsq[NMB];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) { rsum[mb] = 0.f; rsq[mb] = 0.f; }
  // (a lambda called only by the multiplying waves: the accumulators then never merge with the idle waves' zeros)
  // Output staging: the bf16 results go through a wave-private LDS image [token][channel] (16-byte chunks XOR-swizzled by the token)
  // and leave as FULL 128-byte-or-longer row segments: the natural MFMA layout gives every lane 16-byte pieces of 32 different rows
  // per store instruction (8 partial writes per cache line), which measured 13-30 % of the whole kernel on the short-K shapes.
  constexpr int OCH = (GEGLU ? NCB * 16 : ECB * 32);     // output channels of one wave
  const int cbase = wc * (NCB * 32) + kg * (ECB * 32);   // first channel (inside the tile) of the blocks this wave finishes
  constexpr int NCH = OCH / 8, RB = OCH * 2;              // 16-byte chunks / bytes per token row of the wave's image
  constexpr int SWZ = (NCH % 8 == 0) ? 7 : 3;             // chunk index bits that may be XORed without leaving the row
  static_assert(NCH % 4 == 0, "row of at least 4 chunks");
  unsigned char* const stage = lds + wave * (NMB * 32 * RB);
  auto stage_put = [&](int mb, int c, const u32x4& o) {   // chunk c (8 channels) of token mb*32 + l31
    const int tr = mb * 32 + l31;
    *reinterpret_cast<u32x4*>(stage + tr * RB + ((c ^ (tr & SWZ)) << 4)) = o;
  };
  // The epilogue's residual pointer, its row pitch and the statistics pointer as values the compiler cannot re-read from the kernel
  // arguments: with 256 accumulators live it re-loaded each of them in every one of the 32 (slice, block) steps of the four-wave
  // arrangement's epilogue, an s_load + s_waitcnt each (4 of its 6.5 us, tools/probe/gemm4w_stamp.py).
  const uint16_t* e_res = p.res;
  long e_ldr = p.ldr;
  float* e_so = p.stats_out;
  asm volatile("" : "+s"(e_res), "+s"(e_ldr), "+s"(e_so));
  const uint16_t* e_emb = p.emb;  // (the convolution tilings with ten slices per wave re-read these four per slice as well)
  const float* e_bias = p.bias;
  const float* e_ln = p.ln_stats;
  int e_M = p.M, e_N = p.N;
  asm volatile("" : "+s"(e_emb), "+s"(e_bias), "+s"(e_ln), "+s"(e_M), "+s"(e_N));
  auto store_tile = [&]() {
  float mu_[NMB], rs_[NMB];
  if constexpr (!ATTN) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) { mu_[mb] = mu[mb]; rs_[mb] = rs[mb]; }
  } else if (e_ln) {
    const float inv = 1.f / (float)p.ln_dim;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const long m = m0 + mrow0 + mb * 32;
      float s = 0.f, ss = 0.f;
      if (m < e_M) {
        const f32x2* st = reinterpret_cast<const f32x2*>(e_ln) + m * p.ln_parts;
        for (int q = 0; q < p.ln_parts; ++q) {
          const f32x2 v = st[q];
          s += v[0];
          ss += v[1];
        }
      }
      const float mean = s * inv;
      const float var = fmaxf(ss * inv - mean * mean, 0.f);
      mu_[mb] = mean;
      rs_[mb] = rsqrtf(var + p.ln_eps);
    }
  }
  if constexpr (GEGLU) {  // its own instantiation: the erf code next to 256 live accumulators costs the plain epilogue registers
    static_assert(NCB % 2 == 0, "value / gate block pairs");
    {
      // channel blocks 2q / 2q+1 = value / gate columns of the SAME 32 output columns (weight rows interleaved per 64 at pack time)
#pragma unroll
      for (int q = 0; q < NCB / 2; ++q)
#pragma unroll
        for (int c8 = 0; c8 < 2; ++c8) {  // 8 channels at a time keeps the epilogue inside the register budget
          FENCE();
          const int nv = n0 + wc * (NCB * 32) + q * 64 + 16 * hh + 8 * c8;  // packed row of the value block; gate block = + 32
          const int no = ((n0 + wc * (NCB * 32)) >> 1) + q * 32 + 16 * hh + 8 * c8;
          if (nv >= e_N) continue;
          float bv[8], bg[8], sv[8], sg[8];
          if constexpr (LIN_BW) {
            const float* const bl = ASM4 ? reinterpret_cast<const float*>(lds + LBW + wave * 1024) + (nv - n0 - wc * (NCB * 32))
                                         : reinterpret_cast<const float*>(lds + LBW) + (nv - n0);
            const float* const sl = ASM4 ? bl + 128 : reinterpret_cast<const float*>(lds + LBW + LBW_SLICE) + (nv - n0);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 b0 = e_bias ? *reinterpret_cast<const f32x4*>(bl) : z4, b1 = e_bias ? *reinterpret_cast<const f32x4*>(bl + 4) : z4;
            const f32x4 g0 = e_bias ? *reinterpret_cast<const f32x4*>(bl + 32) : z4, g1 = e_bias ? *reinterpret_cast<const f32x4*>(bl + 36) : z4;
            const f32x4 s0 = e_ln ? *reinterpret_cast<const f32x4*>(sl) : z4, s1 = e_ln ? *reinterpret_cast<const f32x4*>(sl + 4) : z4;
            const f32x4 t0 = e_ln ? *reinterpret_cast<const f32x4*>(sl + 32) : z4, t1 = e_ln ? *reinterpret_cast<const f32x4*>(sl + 36) : z4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              bv[r] = b0[r]; bv[4 + r] = b1[r]; bg[r] = g0[r]; bg[4 + r] = g1[r];
              sv[r] = s0[r]; sv[4 + r] = s1[r]; sg[r] = t0[r]; sg[4 + r] = t1[r];
            }
          } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            bv[r] = e_bias ? e_bias[nv + r] : 0.f;
            bg[r] = e_bias ? e_bias[nv + 32 + r] : 0.f;
            sv[r] = e_ln ? p.wsum[nv + r] : 0.f;
            sg[r] = e_ln ? p.wsum[nv + 32 + r] : 0.f;
          }
          }
#pragma unroll
          for (int mb = 0; mb < NMB; ++mb) {
            const long m = m0 + mrow0 + mb * 32;
            if (m >= e_M) continue;
            // LayerNorm fold + bias on pairs of values (v_pk_mul_f32 / v_pk_add_f32), in the operation order this epilogue has always had
            // -- rstd (acc - mu wsum) + bias, every product and sum rounded -- and without a select: no fold means rstd = 1, mu = 0,
            // wsum = 0, for which the same expression is acc + bias exactly
            const f32x2 rs2 = {rs_[mb], rs_[mb]}, mu2 = {mu_[mb], mu_[mb]};
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 xa = {acc[2 * q][mb][8 * c8 + 2 * e], acc[2 * q][mb][8 * c8 + 2 * e + 1]};
              const f32x2 ga = {acc[2 * q + 1][mb][8 * c8 + 2 * e], acc[2 * q + 1][mb][8 * c8 + 2 * e + 1]};
              const f32x2 xv = rs2 * (xa - mu2 * f32x2{sv[2 * e], sv[2 * e + 1]}) + f32x2{bv[2 * e], bv[2 * e + 1]};
              const f32x2 gv = rs2 * (ga - mu2 * f32x2{sg[2 * e], sg[2 * e + 1]}) + f32x2{bg[2 * e], bg[2 * e + 1]};
              const f32x2 v2 = xv * gelu_erf2(gv);
              o[e] = pack_bf16x2(v2[0], v2[1]);
            }
            stage_put(mb, q * 4 + 2 * hh + c8, o);
          }
        }
    }
    return;
  }

  long embrow[CONV ? NMB : 1];  // convolution: element offset of the row's image in the per-image addend
  if constexpr (CONV) {
    const int hw = p.cv_H * p.cv_W;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) embrow[mb] = ((m0 + mrow0 + mb * 32) / hw) * p.emb_stride;
  }
#pragma unroll
  for (int nb = 0; nb < ECB; ++nb) {
#pragma unroll
    for (int c8 = 0; c8 < 2; ++c8) {
      FENCE();  // bound the scheduling region: hoisting every block's loads next to 256 live accumulators spills
      const int n = n0 + cbase + nb * 32 + 16 * hh + 8 * c8;
      if (n >= e_N) continue;  // N % 16 == 0
      float bv[8], sv[8];
      if constexpr (LIN_BW) {
        const float* const bl = ASM4 ? reinterpret_cast<const float*>(lds + LBW + wave * 1024) + (n - n0 - wc * (NCB * 32))
                                     : reinterpret_cast<const float*>(lds + LBW) + (n - n0);
        const float* const sl = ASM4 ? bl + 128 : reinterpret_cast<const float*>(lds + LBW + LBW_SLICE) + (n - n0);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 b0 = e_bias ? *reinterpret_cast<const f32x4*>(bl) : z4, b1 = e_bias ? *reinterpret_cast<const f32x4*>(bl + 4) : z4;
        const f32x4 s0 = e_ln ? *reinterpret_cast<const f32x4*>(sl) : z4, s1 = e_ln ? *reinterpret_cast<const f32x4*>(sl + 4) : z4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bv[r] = b0[r]; bv[4 + r] = b1[r];
          sv[r] = s0[r]; sv[4 + r] = s1[r];
        }
      } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        bv[r] = e_bias ? e_bias[n + r] : 0.f;
        sv[r] = e_ln ? p.wsum[n + r] : 0.f;
      }
      }
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) {
        const long m = m0 + mrow0 + mb * 32;
        if (m >= e_M) continue;
        float v[8];
        {  // (the fold as in the GEGLU epilogue above: packed, in the order it has always had, no select)
          const f32x2 rs2 = {rs_[mb], rs_[mb]}, mu2 = {mu_[mb], mu_[mb]};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 a2 = {acc[nb][mb][8 * c8 + 2 * e], acc[nb][mb][8 * c8 + 2 * e + 1]};
            const f32x2 t2 = rs2 * (a2 - mu2 * f32x2{sv[2 * e], sv[2 * e + 1]}) + f32x2{bv[2 * e], bv[2 * e + 1]};
            v[2 * e] = t2[0];
            v[2 * e + 1] = t2[1];
          }
        }
        if constexpr (CONV) {
          if (e_emb) {
            const u32x4 e0 = *reinterpret_cast<const u32x4*>(e_emb + embrow[mb] + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] += bf16lo_to_f32(e0[e]);
              v[2 * e + 1] += bf16hi_to_f32(e0[e]);
            }
          }
        }
        if (e_res) {
          const u32x4 e0 = *reinterpret_cast<const u32x4*>(e_res + m * e_ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] += bf16lo_to_f32(e0[e]);
            v[2 * e + 1] += bf16hi_to_f32(e0[e]);
          }
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        stage_put(mb, nb * 4 + 2 * hh + c8, o);
        if (!CONV && e_so) {  // statistics of the values as stored (bf16-rounded): what the consumer's LayerNorm fold sees
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a0 = bf16lo_to_f32(o[e]), a1 = bf16hi_to_f32(o[e]);
            rsum[mb] += a0 + a1;
            rsq[mb] = fmaf(a0, a0, fmaf(a1, a1, rsq[mb]));
          }
        }
      }
    }
  }

  };

  // ---- attention epilogues: softmax(q k^T * scale) v per head on the projected tile (sgm/modules/attention.py:368-372,406-408 for a
  // 77-token context: the text cross-attention attn2 of every block and the pose-token attention :578-588) -----------------------------
  // A wave's NMB x 32 tokens x 64 channels accumulator IS the query block of one head (WN waves = WN heads per tile).  The LayerNorm
  // fold and the bias are applied in registers (two FMAs per value) and the rows are packed to bf16 MFMA B fragments without moving: the
  // contraction over the 64 channels runs in the order the accumulator holds them (k-step j <-> channels 32 (j >> 1) + 16 hh + 8 (j & 1)
  // + 0..7), which the K fragments follow by reading chunk 4 (j >> 1) + 2 hh + (j & 1) of their rows.  S^T for all keys at once (keys
  // >= Nk masked through the accumulator's initial value; of a half-used last block only registers 0..7 = its first 16 keys are looked
  // at), one max / exp2 / sum per token, P from the accumulator registers straight into the P V MFMAs (one per 16 keys and 32 channels)
  // with V^T taken from the row-major V rows by ds_read_b64_tr_b16 (attn_fwd.hip's small-Nk kernel, minus its Q round trip through HBM).
  // K / V are already in the LDS when the loop ends (kv_issue); the second key / value set of a de-duplicated tile is fetched between
  // the two passes.
  auto attn_tile = [&]() {
    if constexpr (ATTN) {
      static_assert(!ATTN || NCB == 2, "one head (64 channels) per wave; any number of 32-token blocks and of heads per tile");
      unsigned char* const Ks = lds + KVBASE + wc * HEAD_LDS;
      unsigned char* const Vs = Ks + (F8 ? 96 * 64 : KROWS * 128);
      unsigned char* const Os = lds + wave * (32 * 128);  // this wave's 32-token output block (ring buffer 0 is idle now)
      typedef __attribute__((ext_vector_type(8))) int i32x8;
      // fp8 form: the LayerNorm-folded q stays fp32 in the accumulator registers until every channel of the token has been seen (the
      // e4m3 scale is per token: max |q| over its 64 channels / 448), then becomes ONE B operand of the K = 64 MFMA per 32-token block
      i32x8 q8[F8 ? NMB : 1];
      float qs[F8 ? NMB : 1], qmax[F8 ? NMB : 1];
      if constexpr (F8) {
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) qmax[mb] = 0.f;
      }
      // LayerNorm fold + bias, pack to B fragments (has_ch waves only; the others hold zeros and are skipped below):
      // q = rstd (acc - mu wsum) + bias = acc * rstd + (bias - rstd mu wsum)
      bf16x8 qf[NMB][4];
      if (has_ch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          FENCE();
          const int nb = j >> 1, c8 = j & 1;
          const int n = n0 + wc * 64 + nb * 32 + 16 * hh + 8 * c8;
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          f32x4 b0 = z4, b1 = z4, s0 = z4, s1 = z4;
          if constexpr (BW_LDS) {  // (slices that were not fetched -- no bias, no fold -- are never read)
            const uint32_t bo = BWBASE + (uint32_t)(n - n0) * 4;
            if (p.bias) {
              b0 = *reinterpret_cast<const f32x4*>(lds + bo);
              b1 = *reinterpret_cast<const f32x4*>(lds + bo + 16);
            }
            if (p.ln_stats) {
              s0 = *reinterpret_cast<const f32x4*>(lds + bo + 1024);
              s1 = *reinterpret_cast<const f32x4*>(lds + bo + 1040);
            }
          } else {
            if (p.bias) {
              b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
              b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
            }
            if (p.ln_stats) {
              s0 = *reinterpret_cast<const f32x4*>(p.wsum + n);
              s1 = *reinterpret_cast<const f32x4*>(p.wsum + n + 4);
            }
          }
          const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          const float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
          for (int mb = 0; mb < NMB; ++mb) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t0 = fmaf(acc[nb][mb][8 * c8 + 2 * e], a_c1[mb], fmaf(a_c2[mb], sv[2 * e], bv[2 * e]));
              const float t1 = fmaf(acc[nb][mb][8 * c8 + 2 * e + 1], a_c1[mb], fmaf(a_c2[mb], sv[2 * e + 1], bv[2 * e + 1]));
              if constexpr (F8) {
                acc[nb][mb][8 * c8 + 2 * e] = t0;
                acc[nb][mb][8 * c8 + 2 * e + 1] = t1;
                qmax[mb] = fmaxf(qmax[mb], fmaxf(fabsf(t0), fabsf(t1)));
              } else {
                o[e] = pack_bf16x2(t0, t1);
              }
            }
            if constexpr (!F8) qf[mb][j] = __builtin_bit_cast(bf16x8, o);
          }
        }
        if constexpr (F8) {
          // B operand of v_mfma_scale_f32_32x32x64_f8f6f4: lane (token, hh) supplies k-slots 32 hh + e, e = 0 .. 31 -- here channel
          // 16 hh + e (e < 16) or 32 + 16 hh + (e - 16): the order the accumulator holds them; cd360_kv_pack_fp8 lays K out the same way
#pragma unroll
          for (int mb = 0; mb < NMB; ++mb) {
            float am = fmaxf(qmax[mb], __shfl_xor(qmax[mb], 32));
            am = fmaxf(am, 1e-20f);
            qs[mb] = am * (1.f / 448.f);
            const float inv = 448.f * __builtin_amdgcn_rcpf(am);
#pragma unroll
            for (int jd = 0; jd < 8; ++jd) {
              const int nb = jd >> 2, r0 = 4 * (jd & 3);
              int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(acc[nb][mb][r0] * inv, acc[nb][mb][r0 + 1] * inv, 0, false);
              w8 = __builtin_amdgcn_cvt_pk_fp8_f32(acc[nb][mb][r0 + 2] * inv, acc[nb][mb][r0 + 3] * inv, w8, true);
              q8[mb][jd] = w8;
            }
          }
        }
      }
      const int nrep = (p.a_dup > 0 && a_bidx >= p.a_dup_from) ? 2 : 1;
      for (int rep = 0; rep < nrep; ++rep) {
      const long orow = (long)rep * p.a_dup * p.a_nq;
      if (rep) {  // the first pass has consumed its K / V: fetch the second set into the same rows
        __syncthreads();
        if (!MV || mover) kv_issue(a_bidx + p.a_dup);
        WAIT_VM0();
        __syncthreads();
      }
      if (!has_ch) continue;
      f32x16 init_last;  // accumulator start of the last key block: 0 for real keys, -1e30 for padding
#pragma unroll
      for (int r = 0; r < 16; ++r) init_last[r] = ((NKB - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh < p.a_nk) ? 0.f : -1e30f;
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float c = p.a_scale_log2e;
      float ksc = 1.f, vsc = 1.f;  // fp8: the head's K and V tensor scales (amax / 448, cd360_kv_pack_fp8)
      int koff8[2], voff8[2][2];
      if constexpr (F8) {
        const float* sc = p.a_kvs + ((long)(a_bidx + rep * p.a_dup) * p.a_heads + (n0 >> 6) + wc) * 2;
        ksc = sc[0];
        vsc = sc[1];
        // K row = 64 bytes (four 16-byte chunks, chunk ^ ((row >> 2) & 3)): the lane's 32 bytes are chunks 2 hh, 2 hh + 1;
        // V^T row = 128 bytes (eight chunks, chunk ^ ((row >> 1) & 7)): chunks 4 m + 2 hh, + 1 for the m-th 64-key MFMA
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          koff8[u] = l31 * 64 + (((2 * hh + u) ^ ((l31 >> 2) & 3)) << 4);
#pragma unroll
          for (int m = 0; m < 2; ++m) voff8[m][u] = l31 * 128 + (((4 * m + 2 * hh + u) ^ ((l31 >> 1) & 7)) << 4);
        }
      }
      // per-lane offsets: K fragment of k-step j = chunk 4 (j >> 1) + 2 hh + (j & 1) of row kb * 32 + l31 (16-B XOR swizzle; the rows
      // 16 .. 31 of a half-used last block are the head's first V rows: finite values whose scores nobody reads);
      // V^T fragments as attn_fwd.hip's v_frag_offset / v_frag (32-B swizzle, transposing reads)
      int koff[4], voff[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) koff[j] = l31 * 128 + (((4 * (j >> 1) + 2 * hh + (j & 1)) ^ ((l31 >> 1) & 7)) << 4);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int i = l31 & 15;
        voff[db] = (4 * hh + (i >> 2)) * 128 + (((2 * db + (l31 >> 4)) ^ ((2 * hh + (i >> 3)) & 3)) << 5) + 8 * (i & 3);
      }
      typedef __attribute__((ext_vector_type(4))) short s16x4;
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
      const int lrow = lane >> 3, lchunk = lane & 7;
      const int ocol = n0 + wc * 64 + lchunk * 8;
#pragma unroll  // (a runtime mb would index qf[] dynamically: scratch)
      for (int mb = 0; mb < NMB; ++mb) {
        FENCE();
        f32x16 sT[NKB];
        f32x16 oT[2];
        float inv;
        if constexpr (F8) {
          static_assert(!F8 || NKB == 3, "fp8 epilogue: 96 key slots");
#pragma unroll
          for (int kb = 0; kb < 3; ++kb) {
            const u32x4 lo = *reinterpret_cast<const u32x4*>(Ks + kb * 2048 + koff8[0]), hi = *reinterpret_cast<const u32x4*>(Ks + kb * 2048 + koff8[1]);
            const i32x8 kf = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            sT[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, q8[mb], kb == 2 ? init_last : zero, 0, 0, 0, 0x7f, 0, 0x7f);
          }
          float mx = sT[0][0];
#pragma unroll
          for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sT[kb][r]);
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          // scores = raw * (q scale of the token * K scale of the head); P leaves as p * 2^8 (e4m3 keeps three mantissa bits down to 2^-6:
          // the shift puts the probabilities that matter there) -- the row sum carries the same factor, so O / sum is unchanged
          const float cs = c * qs[mb] * ksc;
          const float mc = fmaf(-mx, cs, 8.f);
          float rsum_p = 0.f;
          i32x8 pb[2];
#pragma unroll
          for (int jd = 0; jd < 12; ++jd) {
            const int kb = jd >> 2, r0 = 4 * (jd & 3);
            const float p0 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r0], cs, mc)), p1 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r0 + 1], cs, mc));
            const float p2 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r0 + 2], cs, mc)), p3 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r0 + 3], cs, mc));
            rsum_p += (p0 + p1) + (p2 + p3);
            int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, 0, false);
            w8 = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, w8, true);
            pb[jd >> 3][jd & 7] = w8;
          }
#pragma unroll
          for (int jd = 4; jd < 8; ++jd) pb[1][jd] = 0;  // key slots 96 .. 127 of the second 64-key MFMA
          rsum_p += __shfl_xor(rsum_p, 32);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
              const u32x4 lo = *reinterpret_cast<const u32x4*>(Vs + db * 4096 + voff8[m][0]), hi = *reinterpret_cast<const u32x4*>(Vs + db * 4096 + voff8[m][1]);
              const i32x8 vf = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
              oT[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pb[m], m == 0 ? zero : oT[db], 0, 0, 0, 0x7f, 0, 0x7f);
            }
          inv = __builtin_amdgcn_rcpf(rsum_p) * vsc;
        } else {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bf16x8 kfr = *reinterpret_cast<const bf16x8*>(Ks + kb * 32 * 128 + koff[j]);
            sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[mb][j], j == 0 ? (kb == NKB - 1 ? init_last : zero) : sT[kb], 0, 0, 0);
          }
        float mx = sT[0][0];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < ((HALF && kb == NKB - 1) ? 8 : 16); ++r) mx = fmaxf(mx, sT[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mc = -mx * c;
        float rsum_p = 0.f;
        uint32_t pk[NK16 * 4];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < ((HALF && kb == NKB - 1) ? 8 : 16); r += 2) {
            const float p0 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r], c, mc)), p1 = __builtin_amdgcn_exp2f(fmaf(sT[kb][r + 1], c, mc));
            rsum_p += p0 + p1;
            pk[kb * 8 + (r >> 1)] = pack_bf16x2(p0, p1);
          }
        rsum_p += __shfl_xor(rsum_p, 32);
#pragma unroll
        for (int kk = 0; kk < NK16; ++kk) {
          const u32x4 pw = {pk[kk * 4 + 0], pk[kk * 4 + 1], pk[kk * 4 + 2], pk[kk * 4 + 3]};
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Vs + voff[db] + kk * 16 * 128));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Vs + voff[db] + kk * 16 * 128 + 8 * 128));
            const s16x8 vfr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vfr), __builtin_bit_cast(bf16x8, pw), kk == 0 ? zero : oT[db], 0, 0, 0);
          }
        }
        inv = __builtin_amdgcn_rcpf(rsum_p);
        }
        // O^T registers (lane = token, 4 consecutive channels per group) -> this wave's LDS block -> full 128-byte rows
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int dbyte = (db * 32 + 8 * g + 4 * hh) * 2;
            const u32x2 wv = {pack_bf16x2(oT[db][4 * g + 0] * inv, oT[db][4 * g + 1] * inv), pack_bf16x2(oT[db][4 * g + 2] * inv, oT[db][4 * g + 3] * inv)};
            *reinterpret_cast<u32x2*>(Os + l31 * 128 + (((dbyte >> 4) ^ ((l31 >> 1) & 7)) << 4) + (dbyte & 8)) = wv;
          }
        // (token tiles never straddle the end of the matrix: M = B Nq with Nq a multiple of the tile height)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * i + lrow;
          const long m = m0 + wr * (NMB * 32) + mb * 32 + row;
          const u32x4 v = *reinterpret_cast<const u32x4*>(Os + row * 128 + ((lchunk ^ ((row >> 1) & 7)) << 4));
          if (p.wt) __builtin_amdgcn_raw_buffer_store_b128(v, __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7fffffff, 0x00020000),
                                                           (int)(((m + orow) * p.ldo + ocol) * 2), 0, 16);
          else *reinterpret_cast<u32x4*>(p.out + (m + orow) * p.ldo + ocol) = v;
        }
      }
      }  // rep
    }
  };
#if CD360_GEMM_SCHED & 2
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);  // (probe: static priority for the second-dispatched half of the workgroup)
#endif
  // ASM4: the ring is all of the LDS while the loop runs, so this wave's 128 channels of bias and wsum wait in four registers
  // (lane l: channels l and 64 + l), requested here, and are parked in the LDS behind the loop
  float bw4[ASM4 ? 4 : 1];
  if constexpr (ASM4) {
    const int c = n0 + wc * (NCB * 32) + lane;
    bw4[0] = (p.bias && c < p.N) ? p.bias[c] : 0.f;
    bw4[1] = (p.bias && c + 64 < p.N) ? p.bias[c + 64] : 0.f;
    bw4[2] = (p.ln_stats && c < p.N) ? p.wsum[c] : 0.f;
    bw4[3] = (p.ln_stats && c + 64 < p.N) ? p.wsum[c + 64] : 0.f;
  }
  G4STAMP(1);
  if constexpr (ASM4) {
    // descriptors as plain words (the asm takes them in SGPRs): base, base_hi (stride 0), bytes, flags -- the ranges of xrsrc / wrsrc
    const uint64_t pa = (uint64_t)p.a, pw = (uint64_t)p.w;
    u32x4 xd = {(uint32_t)pa, (uint32_t)(pa >> 32) & 0xffffu, (uint32_t)(((long)p.M - 1) * p.lda * 2 + (long)p.K * 2), 0x00020000u};
    u32x4 wd = {(uint32_t)pw, (uint32_t)(pw >> 32) & 0xffffu, (uint32_t)(((long)p.N - 1) * p.ldw * 2 + (long)p.K * 2), 0x00020000u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xd[i] = __builtin_amdgcn_readfirstlane(xd[i]);
      wd[i] = __builtin_amdgcn_readfirstlane(wd[i]);
    }
    const uint32_t s_lds = __builtin_amdgcn_readfirstlane((uint32_t)(dwave * 1024));
    const uint32_t xo0 = xo[0] - XREG, wo0 = wo[0] - WREG;  // offsets inside a 32 KB ring slot
    const uint32_t s_xstep = __builtin_amdgcn_readfirstlane(xstep), s_wstep = __builtin_amdgcn_readfirstlane(wstep);
    const uint32_t s_nk = __builtin_amdgcn_readfirstlane((uint32_t)nk);
#ifdef CD360_WHATIF
#define CD360_G4_ASM(TXT)                                                                                                                     \
    asm volatile(TXT : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]),     \
                 "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]),         \
                 "+a"(acc[3][2]), "+a"(acc[3][3])                                                                                                \
                 : "v"(xoff0), "v"(woff0), "s"(s_lds), "v"(xo0), "v"(wo0), "s"(xd), "s"(wd), "s"(s_xstep), "s"(s_wstep), "s"(s_nk)                \
                 : CD360_GEMM4W_CLOBBERS)
    if (abl & 0x80000) CD360_G4_ASM(CD360_GEMM4W_LOOP_V128);
    else if (abl & 0x40000) CD360_G4_ASM(CD360_GEMM4W_LOOP_V64);
    else if (abl & 0x20000) CD360_G4_ASM(CD360_GEMM4W_LOOP_V32);
    else if (abl & 0x10000) CD360_G4_ASM(CD360_GEMM4W_LOOP_V16);
    else if (abl & 0x8000) CD360_G4_ASM(CD360_GEMM4W_LOOP_V8);
    else if (abl & 0x4000) CD360_G4_ASM(CD360_GEMM4W_LOOP_V4);
    else if (abl & 0x2000) CD360_G4_ASM(CD360_GEMM4W_LOOP_V3);
    else if (abl & 0x1000) CD360_G4_ASM(CD360_GEMM4W_LOOP_V1);
    else CD360_G4_ASM(CD360_GEMM4W_LOOP);
#undef CD360_G4_ASM
#else
    asm volatile(CD360_GEMM4W_LOOP
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]),
                   "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]),
                   "+a"(acc[3][2]), "+a"(acc[3][3])
                 : "v"(xoff0), "v"(woff0), "s"(s_lds), "v"(xo0), "v"(wo0), "s"(xd), "s"(wd), "s"(s_xstep), "s"(s_wstep), "s"(s_nk)
                 : CD360_GEMM4W_CLOBBERS);
#endif
  } else if constexpr (MV > 0) {
    if (mover) k_loop(std::false_type{}, std::true_type{});
    else if (has_ch) k_loop(std::true_type{}, std::false_type{});
    else k_loop(std::false_type{}, std::false_type{});
  } else {
    if (has_ch) k_loop(std::true_type{}, std::true_type{});
    else k_loop(std::false_type{}, std::true_type{});
  }
  G4STAMP(2);
  if constexpr (ATTN) WAIT_VM0();  // this wave's K / V pieces have landed (visible to the others behind the barrier)
  __syncthreads();  // every wave is past its last fragment read: the K-loop buffers become the output staging area
  G4STAMP(3);
  if constexpr (ASM4) {  // (bw4: requested ahead of the loop, see there) -> this wave's kilobyte behind the staging image
    float* const d = reinterpret_cast<float*>(lds + LBW + wave * 1024) + lane;
    d[0] = bw4[0];
    d[64] = bw4[1];
    d[128] = bw4[2];
    d[192] = bw4[3];
  }
#ifdef CD360_GEMM_STAMP
  if (CD360_GEMM_STAMP == 1 && p.stamp) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(lds + NBUF * (XB + WB));
    for (int i = tid; i < NWC * 64 * 8; i += 64 * NW) p.stamp[(long)blockIdx.x * (NWC * 64 * 8) + i] = src[i];
    __syncthreads();
  }
#endif
  if constexpr (ATTN) {
    static_assert(!ATTN || KS == 1, "the attention epilogues own their tile");
    attn_tile();
    return;
  }
  if constexpr (KS == 2) {
    // The two k-step groups' partial sums meet: group kg keeps channel blocks [kg ECB, (kg + 1) ECB) and receives the other group's
    // partials of them (fp32, lane-linear 16-byte pieces behind the staging image); its result ends up in acc[0 .. ECB-1].
    constexpr int XF = ECB * NMB * 16 * 64;  // floats one wave sends
    float* const xbase = reinterpret_cast<float*>(lds + BM * BN * 2) + lane * 4;
    if (has_ch) {
      float* const snd = xbase + wave * XF;
#pragma unroll
      for (int nb = 0; nb < ECB; ++nb)
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
            if (kg == 0) v = f32x4{acc[ECB + nb][mb][4 * q], acc[ECB + nb][mb][4 * q + 1], acc[ECB + nb][mb][4 * q + 2], acc[ECB + nb][mb][4 * q + 3]};
            else v = f32x4{acc[nb][mb][4 * q], acc[nb][mb][4 * q + 1], acc[nb][mb][4 * q + 2], acc[nb][mb][4 * q + 3]};
            *reinterpret_cast<f32x4*>(snd + ((nb * NMB + mb) * 4 + q) * 256) = v;
          }
    }
    __syncthreads();
    if (has_ch) {
      const float* const rcv = xbase + (wv + (1 - kg) * NWT) * XF;
#pragma unroll
      for (int nb = 0; nb < ECB; ++nb)
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) {
          if (kg == 1) acc[nb][mb] = acc[ECB + nb][mb];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rcv + ((nb * NMB + mb) * 4 + q) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nb][mb][4 * q + e] += v[e];
          }
        }
    }
  }
  if (has_ch) {
    store_tile();
    G4STAMP(4);
    // (wave-private image: the compiler's lgkmcnt wait orders the ds_writes before the ds_reads, no barrier)
    const int ocol0 = GEGLU ? (n0 >> 1) + wc * OCH : n0 + cbase;
    const int nout = GEGLU ? (p.N >> 1) : p.N;
    // lane -> (row it * RPI + lane / NCH, chunk lane % NCH): a lane keeps its 8 channels over all rows (the per-channel sums of the
    // convolution epilogue accumulate in registers); chunk counts that do not divide 64 leave the last lanes idle
    constexpr int RPI = 64 / NCH;
    const int j = lane % NCH, rl = lane / NCH;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7fffffff, 0x00020000);
    float cs[CSTATS ? 8 : 1], cq[CSTATS ? 8 : 1];
    if constexpr (CSTATS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs[e] = 0.f; cq[e] = 0.f; }
    }
    // The image is read back in batches of eight row groups -- eight ds_read_b128 in flight, then their stores: one read, one wait, one
    // store per row group left the LDS latency in the open 16 .. 32 times per wave (round 6).  Rows / lanes without an output row read a
    // clamped address of the wave's own image and store nothing.
    constexpr int NIT = (NMB * 32 + RPI - 1) / RPI, RBATCH = 8;
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += RBATCH) {
      u32x4 ob[RBATCH];
#pragma unroll
      for (int b = 0; b < RBATCH; ++b) {
        if (it0 + b < NIT) {
          const int rr = (it0 + b) * RPI + rl, rc = rr < NMB * 32 ? rr : NMB * 32 - 1;
          ob[b] = *reinterpret_cast<const u32x4*>(stage + rc * RB + (((j < NCH ? j : 0) ^ (rc & SWZ)) << 4));
        }
      }
#pragma unroll
      for (int b = 0; b < RBATCH; ++b) {
      if (it0 + b >= NIT) continue;
      const int it = it0 + b;
      const int r = it * RPI + rl;
      const long m = m0 + wr * (NMB * 32) + r;
      if (rl < RPI && r < NMB * 32 && m < p.M && ocol0 + j * 8 < nout && !(abl & 64)) {
        const u32x4 o = ob[b];
        long orow_ = m;
        if constexpr (CONV) {
          if (p.cv_up) {  // source pixel (n, i, j) of phase (a, b) -> pixel (n, 2 i + a, 2 j + b) of the 2 H x 2 W output
            const int hw = p.cv_H * p.cv_W;
            const long n_ = m / hw;
            const int rem = (int)(m - n_ * hw), i_ = rem / p.cv_W, j_ = rem - i_ * p.cv_W;
            orow_ = (n_ * (2 * p.cv_H) + 2 * i_ + up_a) * (2L * p.cv_W) + 2 * j_ + up_b;
          }
        }
        if (p.wt) __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, (int)((orow_ * p.ldo + ocol0 + j * 8) * 2), 0, 16);  // aux 16 = sc1
        else *reinterpret_cast<u32x4*>(p.out + orow_ * p.ldo + ocol0 + j * 8) = o;
        if constexpr (CSTATS) {
          if (p.cstats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a0 = bf16lo_to_f32(o[e]), a1 = bf16hi_to_f32(o[e]);
              cs[2 * e] += a0;
              cs[2 * e + 1] += a1;
              cq[2 * e] = fmaf(a0, a0, cq[2 * e]);
              cq[2 * e + 1] = fmaf(a1, a1, cq[2 * e + 1]);
            }
          }
        }
      }
      }  // b
    }
    if constexpr (CSTATS) {
      if (p.cstats) {  // fold the RPI row groups (lanes j, j + NCH, ...) in a fixed order, lane j writes its 8 channels
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if constexpr ((RPI & (RPI - 1)) == 0) {  // NCH divides 64: xor tree over the lane bits above the chunk index
#pragma unroll
            for (int o = 32; o >= NCH; o >>= 1) {
              cs[e] += __shfl_xor(cs[e], o);
              cq[e] += __shfl_xor(cq[e], o);
            }
          } else {
            float s_ = cs[e], q_ = cq[e];
#pragma unroll
            for (int k = 1; k < RPI; ++k) {
              s_ += __shfl(cs[e], lane + k * NCH);
              q_ += __shfl(cq[e], lane + k * NCH);
            }
            cs[e] = s_;
            cq[e] = q_;
          }
        }
        if (lane < NCH && ocol0 + j * 8 < nout && m0 + wr * (NMB * 32) < p.M) {  // (a wave whose rows are all past M has no slab)
          const long slab = (m0 + wr * (NMB * 32)) / (NMB * 32);
          float* d = p.cstats + (slab * p.N + ocol0 + j * 8) * 2;
#pragma unroll
          for (int e = 0; e < 8; e += 2) *reinterpret_cast<f32x4*>(d + 2 * e) = f32x4{cs[e], cq[e], cs[e + 1], cq[e + 1]};
        }
      }
    }
  }

  G4STAMP(5);
  if (!CONV && p.stats_out) {  // kernel-uniform: per-row sums over this N tile = both lane halves, all channel blocks, the WN waves of the row
    float* red = reinterpret_cast<float*>(lds);  // [wc][BM][2]; the K-loop buffers are idle once every wave is past its last read
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const float s = rsum[mb] + __shfl_xor(rsum[mb], 32);
      const float q = rsq[mb] + __shfl_xor(rsq[mb], 32);
      if (hh == 0 && !mover) {
        float* d = red + (((wc * KS + kg) * BM) + mrow0 + mb * 32) * 2;
        d[0] = s;
        d[1] = q;
      }
    }
    __syncthreads();
    if (tid < BM) {
      const long m = m0 + tid;
      if (m < p.M) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int c = 0; c < WN * KS; ++c) {
          s += red[(c * BM + tid) * 2];
          q += red[(c * BM + tid) * 2 + 1];
        }
        float* d = p.stats_out + (m * p.tiles_n + tn) * 2;
        d[0] = s;
        d[1] = q;
      }
    }
  }
}

template <int WM, int WN, int NCB, int NMB, int NBUF, int KS, int MV, int EPI>
int launch_mv(const GemmParams& p0, hipStream_t stream) {
  GemmParams p = p0;
  constexpr int BM = WM * NMB * 32, BN = WN * NCB * 32;
  // K-loop buffers, reused as the output staging image (+ the fp32 partial tile of the second k-step group)
  // (the attention epilogues keep the K / V rows of the tile's WN heads, 2 x NK16 x 16 rows of 128 bytes each, behind ring buffer 0; the
  // waves' 4-KB output blocks alias buffer 0 and the projected tile never leaves the registers)
  constexpr bool ATTN = (EPI >= 2 && EPI <= 4) || (EPI >= 7 && EPI <= 10);
  constexpr int NK16 = (EPI == 4 || EPI == 10) ? 6 : EPI == 2 ? 2 : EPI == 3 ? 4 : EPI == 7 ? 5 : EPI == 8 ? 3 : 1;
  constexpr int ATTN_KV_END = (BM + BN) * 128 + WN * (EPI == 10 ? 96 * 64 + 64 * 128 : 2 * NK16 * 16 * 128);  // (EPI 10: the packed fp8 image)
  constexpr int ATTN_BYTES = ATTN ? ATTN_KV_END + ((BN * 4 <= 1024 && ATTN_KV_END + 2048 <= 160 * 1024) ? 2048 : 0) : 0;  // + bias / wsum slices (BW_LDS)
  constexpr int PRL = 8 * (MV ? MV : WM * WN * KS);  // rows per DMA piece: the ring's buffers are whole pieces (see XB / WB in the kernel)
  constexpr bool ASM4 = WM == 2 && WN == 2 && NCB == 4 && NMB == 4 && NBUF == 2 && KS == 1 && MV == 0 && (EPI == 0 || EPI == 1);  // five 32 KB slots
  constexpr int RING_BYTES = ASM4 ? 5 * 32768 : EPI == 11 ? 2 * 9 * PRL * 128 + NBUF * (((BN + PRL - 1) / PRL) * PRL) * 128  // two halo buffers + the channel ring
                                       : NBUF * (((BM + PRL - 1) / PRL) * PRL + ((BN + PRL - 1) / PRL) * PRL) * 128,
                STAGE_BYTES0 = ATTN ? 0 : BM * BN * 2 + (KS == 2 ? BM * BN * 4 : 0);
  // linear / convolution epilogues: + the bias / wsum slices behind ring and staging image when they fit (LIN_BW in the kernel)
  constexpr int LBW0 = ASM4 ? STAGE_BYTES0 : (RING_BYTES > STAGE_BYTES0 ? RING_BYTES : STAGE_BYTES0), LBW_SL = ((BN * 4 + 1023) / 1024) * 1024;
  constexpr int LIN_BW_END = (!ATTN && LBW0 + (ASM4 ? 4096 : 2 * LBW_SL) <= 160 * 1024) ? LBW0 + (ASM4 ? 4096 : 2 * LBW_SL) : 0;
  constexpr int STAGE_BYTES1 = STAGE_BYTES0 > ATTN_BYTES ? STAGE_BYTES0 : ATTN_BYTES;
  constexpr int STAGE_BYTES = STAGE_BYTES1 > LIN_BW_END ? STAGE_BYTES1 : LIN_BW_END;
#ifdef CD360_GEMM_STAMP
  constexpr int BASE_BYTES = RING_BYTES > STAGE_BYTES ? RING_BYTES : STAGE_BYTES;
  constexpr int STAMP_BYTES = BASE_BYTES + WM * WN * KS * 64 * 32 <= 160 * 1024 ? WM * WN * KS * 64 * 32 : 0;
  constexpr int LDS_BYTES = BASE_BYTES + STAMP_BYTES;
#else
  constexpr int LDS_BYTES = RING_BYTES > STAGE_BYTES ? RING_BYTES : STAGE_BYTES;
#endif
  static_assert(LDS_BYTES <= 160 * 1024, "LDS of one CU");
  static_assert(!ATTN || (BM % PRL == 0 && BN % PRL == 0), "attention epilogues: exact pieces (ATTN_KV_END assumes unpadded buffers)");
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const cd360_tuning& tune = cd360_tune();
  p.group_m = tune.gemm_group_m > 0 ? tune.gemm_group_m : 6;  // token tiles per group (swept 1 .. 48 on the block's shapes: 6 best or tied)
  // Write-through output stores.  The eight XCDs' L2s are not coherent with each other, so a kernel boundary has to write back every line
  // the kernel left dirty before its consumer may start (MI355X_MICROARCH.md, price row "boundary": + B / 6 TB/s behind B dirty bytes --
  // 1.3 us behind the 7.9 MB of a C -> C projection, 5 us behind FF1's 31 MB).  Stores with agent scope (sc1) go through to memory as
  // they are issued, under the other workgroups' K loops and epilogues, and the boundary finds nothing to write back.
  // MEASURED (round 5, profiles/r05_store_wt_ab.txt): chains of one shape -0.4 ... -3.8 % per launch, outputs bit-identical -- and the
  // captured denoise step unchanged (27.69 / 27.75 ms plain, 27.71 / 28.25 ms write-through, alternating on one box): inside the step the
  // write-back already overlaps the next launch's prologue.  Off by default; cd360_tuning.store_wt = 1 turns it on.
  {
    const long out_rows = ((EPI == 5 && p.cv_up) ? 4L : ATTN ? 2L : 1L) * p.M;  // (the de-duplicated attention writes rows of batch i + a_dup too)
    p.wt = (tune.store_wt > 0 && (out_rows - 1) * p.ldo * 2 + (long)p.N * 2 < 0x7fffffffL) ? 1 : 0;
  }
  p.abl = 0;
#ifdef CD360_WHATIF
  if (tune.whatif > 0) p.abl = tune.whatif;
#endif
#ifdef CD360_GEMM_STAMP
  p.stamp = nullptr;
  if ((STAMP_BYTES || CD360_GEMM_STAMP == 2) && !(tune.reserved[0] == -1 && tune.reserved[1] == -1))  // probe build: device pointer of the stamp buffer in reserved[0..1]
    p.stamp = reinterpret_cast<uint32_t*>(((uint64_t)(uint32_t)tune.reserved[1] << 32) | (uint64_t)(uint32_t)tune.reserved[0]);
#endif
  const long nwg = (long)p.tiles_m * p.tiles_n * ((EPI == 5 && p.cv_up) ? 4 : 1);
  if (nwg > 0x7fffffffL) return CD360_ERR_SHAPE;
  // weight prefetcher (prefetch.hip; armed only while a step is being captured): a small kernel on the forked side stream touches THIS
  // launch's weights as soon as the launch `lag` positions earlier has finished, i.e. while its predecessors compute
  cd360_prefetch_before_launch(stream, p.w, ((long)p.N * ((EPI == 5 && p.cv_up) ? 4 : 1) - 1) * p.ldw * 2 + (long)p.K * 2);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_kernel<WM, WN, NCB, NMB, NBUF, KS, MV, EPI>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (attr != hipSuccess) return CD360_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_mfma_kernel<WM, WN, NCB, NMB, NBUF, KS, MV, EPI>), dim3((unsigned)nwg), dim3(64 * (WM * WN * KS + MV)), LDS_BYTES, stream, p);
  CD360_LAUNCH_CHECK();
  cd360_prefetch_after_launch(stream);
  return CD360_OK;
}

// Mover waves (MV = 4: one per SIMD) where the multiplying waves leave the registers for a third wave per SIMD (<= 168) and the epilogue
// is a linear / convolution one.  cd360_tuning.gemm_movers = 0 turns them off, 4 on for every arrangement that can take them; -1 = the
// arrangements they were measured to help (tools/bench_gemm.py movers).
template <int WM, int WN, int NCB, int NMB, int NBUF, int KS, int EPI>
int launch_ks(const GemmParams& p, hipStream_t stream) {
  constexpr bool CAN = NCB * NMB <= 6 && WM * WN * KS + 4 <= 16;
  if constexpr (EPI == 11) {  // the halo convolution exists with mover waves only
    return launch_mv<WM, WN, NCB, NMB, NBUF, KS, 4, EPI>(p, stream);
  } else {
  if constexpr (CAN) {
    // measured (hipGraph-timed, interleaved): the four- and three-buffer arrangements -5 ... -10 % (C -> C 18.7 -> 17.8 us, FF2 50.1 -> 47.0,
    // 3 x 3 convolutions at 32^2 / 64^2 108 -> 98 / 176 -> 161), 256 x 192 -2.6 %, the two-buffer 128 x 128 with two workgroups per CU +-0
    constexpr bool DEFAULT_ON = NBUF >= 3 || NCB == 3;
    const int mv = cd360_tune().gemm_movers;
    if (mv >= 0 ? mv == 4 : DEFAULT_ON) return launch_mv<WM, WN, NCB, NMB, NBUF, KS, 4, EPI>(p, stream);
  }
  return launch_mv<WM, WN, NCB, NMB, NBUF, KS, 0, EPI>(p, stream);
  }
}

template <int WM, int WN, int NCB, int NMB, int NBUF, int EPI>
int launch_epi(const GemmParams& p, hipStream_t stream) {
  return launch_ks<WM, WN, NCB, NMB, NBUF, 1, EPI>(p, stream);
}

// The 128 x 128 tiling with four LDS buffers (launches of at most one workgroup per CU), as EPI 0 / 5 / 6, in its wave arrangements:
// eight waves of 64 x 32, every wave on all four k-steps of a K-tile (mode 0), or eight waves of 64 x 64 in two k-step groups (mode 1:
// a third fewer fragment reads per K-tile against one exchange of partial sums through the LDS in front of the epilogue).  Measured on
// the 1280-level shapes, interleaved on one box (tools/bench_gemm.py ksplit, us, mode 0 / mode 1): K = 1280 19.9 / 20.7, K = 2560
// 30.0 / 29.7, K = 5120 51.9 / 49.3, the 3 x 3 convolutions at 32^2 (K = 11520 .. 23040) 112.7 / 108.4, 166.5 / 156.5, 220.0 / 206.0 --
// so the split serves K >= 3072.  (Mode 2, four waves of 64 x 64 with one wave per SIMD, is 10 % slower everywhere: kept for A/B only.)
// cd360_tuning.gemm_ksplit = 0 | 1 | 2 forces a mode.
template <int EPI>
int launch_128x4(const GemmParams& p, hipStream_t stream) {
  const int forced = cd360_tune().gemm_ksplit;
  const int mode = forced >= 0 ? forced : (p.K >= 3072 ? 1 : 0);
  if (mode == 1) return launch_ks<2, 2, 2, 2, 4, 2, EPI>(p, stream);
  if (mode == 2) return launch_ks<2, 2, 2, 2, 4, 1, EPI>(p, stream);
  return launch_ks<2, 4, 1, 2, 4, 1, EPI>(p, stream);
}

// 64 x 128 tiles for launches whose 128 x 128 tiling would leave more than half of the 256 CUs without a workgroup (the 1280-wide
// projections of a 1024-token batch: 80 tiles -> 160): eight waves of 32 x 32 on all four k-steps, or -- K >= 3072 -- eight waves of
// 32 x 64 in two k-step groups; four LDS buffers + mover waves as the 128 x 128 arrangement.
template <int EPI>
int launch_64x4(const GemmParams& p, hipStream_t stream) {
  const int forced = cd360_tune().gemm_ksplit;
  const int mode = forced >= 0 ? forced : (p.K >= 3072 ? 1 : 0);
  if (mode == 1) return launch_ks<2, 2, 2, 1, 4, 2, EPI>(p, stream);
  return launch_ks<2, 4, 1, 1, 4, 1, EPI>(p, stream);
}

template <int WM, int WN, int NCB, int NMB, int NBUF>
int launch(const GemmParams& p, hipStream_t stream) {
  return p.geglu ? launch_epi<WM, WN, NCB, NMB, NBUF, 1>(p, stream) : launch_epi<WM, WN, NCB, NMB, NBUF, 0>(p, stream);
}

// Tilings (tokens x channels, waves, LDS buffers): 1 = 128 x 128, 4 waves of 64 x 64, 2 buffers (two workgroups per CU);
// 2 = 128 x 128, 8 waves of 64 x 32, 2 buffers; 3 = 256 x 256, 8 waves of 128 x 64, 2 buffers; 4 = as 2 with 4 buffers (3 tiles in
// flight: long K loops of launches with one workgroup per CU); 5 = 256 x 128, 8 waves of 64 x 64, 3 buffers; 6 = 256 x 192, 8 waves of
// 64 x 96, 2 buffers; 7 = 256 x 256, SIXTEEN waves of 64 x 64 (four per SIMD), 2 buffers; 8 = 64 x 128, 8 waves, 4 buffers (launch_64x4)
constexpr int NCFG = 9;  // 9 = 256 x 256 as FOUR waves of 128 x 128 (one per SIMD): A/B
constexpr int CFG_BM[NCFG + 1] = {0, 128, 128, 256, 128, 256, 256, 256, 64, 256}, CFG_BN[NCFG + 1] = {0, 128, 128, 256, 128, 128, 192, 256, 128, 256};
int pick_cfg(int64_t M, int N, bool geglu) {
  const int cfg = cd360_tune().gemm_cfg;  // tuning / A-B override
  if (cfg >= 1 && cfg <= NCFG && !(geglu && (cfg == 2 || cfg == 4 || cfg == 6 || cfg == 8))) return cfg;
  if (geglu) return 7;  // FF1 + GEGLU: sixteen waves of 64 x 64 on the 256 x 256 tile, -3 % against eight of 128 x 64 (bit-identical results)
  // Measured on the SDXL shapes (tools/bench_gemm.py, profiles/r02_gemm_shapes.txt).  Narrow outputs (the C -> C projections and the
  // feed-forward's second Linear): 128 x 128 tiles -- with four LDS buffers when the launch has at most one workgroup per CU (the
  // 1280-wide level: 240 tiles), two otherwise (two workgroups per CU overlap each other's prologue / epilogue) -- except for very
  // tall problems (the FeatureNeRF pose tokens) where 256 x 256 tiles amortise the weights better.
  // Small batches (the fine-tune step's target stream, context / embedding projections; tools/bench_gemm.py small_m, mid_m): with at most
  // 128 tiles of 128 x 128, 64 x 128 tiles double the busy CUs (M = 1024, N = 1280: 14.0 -> 11.7 us at K = 1280, 63.5 -> 50.8 at K = 10240;
  // M = 320, N = 2560: 16.5 -> 14.1 where the wide rule below took 32.9)
  const long nwg = ((M + 127) / 128) * ((N + 127) / 128);
  const bool small = cd360_tune().gemm_small != 0;
  if (small && nwg <= 128) return 8;
  if (N <= 1536) {
    if (M >= 65536) return 3;
    return nwg <= 256 ? 4 : 2;
  }
  // wide outputs that 256-wide tiles cannot spread over the chip: 128 x 128 up to one workgroup per CU (M = 1024, N = 3840: 15.2 us
  // against 25.9 on 256 x 192), 256 x 128 with three buffers up to two (M = 1024, N = 5120: 20.5 against 27.8; M = 4096, N = 1920: 16.2 / 19.7)
  if (small && nwg <= 256) return 4;
  if (small && nwg <= 512) return 5;
  // Wide outputs: 256 x 256 tiles unless 256 x 192 fills the 256 CUs better (q|k|v: 12 x 15 = 180 tiles against 12 x 20 = 240)
  auto eff = [&](int bm, int bn) {
    const long tm = (M + bm - 1) / bm, tn = (N + bn - 1) / bn, nwg = tm * tn, rounds = (nwg + 255) / 256;
    return (double)nwg / (double)(rounds * 256) * (double)N / (double)(tn * bn);
  };
  return eff(256, 192) * 0.95 > eff(256, 256) ? 6 : 3;
}

}  // namespace

// Number of output columns per (sum, sumsq) partial that cd360_gemm_bf16(..., stats_out) writes for an [M, N] output: the consumer
// passes ceil(N / that) back as `ln_parts`.  Depends on the tiling the launch will choose -- ask with the same M, N.
extern "C" int cd360_gemm_tile_n(int64_t M, int N) { return CFG_BN[pick_cfg(M, N, false)]; }

// Rows per slab of the channel statistics cd360_gemm_cstats_bf16 writes for an [M, N] output (the launch's wave tiling decides)
extern "C" int cd360_gemm_cstats_rows(int64_t M, int N) {
  const int cfg = pick_cfg(M, N, false);
  return cfg == 2 || cfg == 4 ? 64 : (cfg == 8 ? 32 : 0);
}

// out[M, N] (bf16, row stride ldo) = epilogue(A[M, K] @ W[N, K]^T); A, W bf16 with row strides lda, ldw (elements, multiples of 8),
// K % 64 == 0, N % 16 == 0, all base pointers 16-byte aligned.
//   bias     fp32 [N] | NULL
//   res      bf16 [M, N] row stride ldr | NULL : added last
//   ln_stats fp32 [M, ln_parts, 2] | NULL : LayerNorm over the ln_dim (= K) channels of each A row folded in front of the GEMM --
//            W must be the gamma-scaled weight, `wsum` fp32 [N] its row sums, `bias` must already contain beta W^T
//   stats_out fp32 [M, ceil(N / cd360_gemm_tile_n(M, N)), 2] | NULL : per-row partial (sum, sumsq) of the stored bf16 outputs
//   flags bit 0: GEGLU -- W rows are packed per 64-row group as [32 value rows | 32 gate rows] of the same 32 output columns
//            (cd360.ops.geglu_row_order); out is [M, N / 2]
extern "C" int cd360_gemm_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                               const void* bias, const void* res, int64_t ldr, const void* ln_stats, int ln_parts, int ln_dim, float ln_eps,
                               const void* wsum, void* stats_out, int flags, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!a || !w || !out || M <= 0 || N <= 0 || K <= 0) return CD360_ERR_ARG;
  if (K % 64 || N % 16 || lda % 8 || ldw % 8 || ldo % 8 || (res && ldr % 8) || lda < K || ldw < K) return CD360_ERR_SHAPE;
  if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)out | (uintptr_t)res) % 16) return CD360_ERR_ARG;
  if (((uintptr_t)bias | (uintptr_t)ln_stats | (uintptr_t)wsum | (uintptr_t)stats_out) % 8) return CD360_ERR_ARG;
  if (M > 0x7fffffffL || (M + 256) * lda * 2 >= (1L << 32) || ((long)N + 256) * ldw * 2 >= (1L << 32)) return CD360_ERR_SHAPE;  // 32-bit buffer offsets
  if (ln_stats && (!wsum || ln_parts <= 0 || ln_dim <= 0)) return CD360_ERR_ARG;
  const bool geglu = flags & 1;
  if (geglu && (N % 64 || stats_out || res)) return CD360_ERR_SHAPE;
  GemmParams p;
  p.a = (const uint16_t*)a; p.w = (const uint16_t*)w; p.out = (uint16_t*)out; p.bias = (const float*)bias; p.res = (const uint16_t*)res;
  p.ln_stats = (const float*)ln_stats; p.wsum = (const float*)wsum; p.stats_out = (float*)stats_out;
  p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = res ? ldr : 0;
  p.M = (int)M; p.N = N; p.K = K; p.ln_parts = ln_parts; p.ln_dim = ln_dim; p.ln_eps = ln_eps; p.geglu = geglu ? 1 : 0;
  p.tiles_m = p.tiles_n = p.group_m = 0;
  p.ak = p.av = nullptr; p.ak_sb = p.ak_sn = p.av_sb = p.av_sn = 0; p.a_nq = p.a_nk = 0; p.a_scale_log2e = 0.f; p.a_dup = p.a_dup_from = 0; p.a_kv8 = nullptr; p.a_kvs = nullptr; p.a_heads = 0;
  p.cv_H = p.cv_W = p.cv_kg = p.cv_up = 0; p.emb = nullptr; p.emb_stride = 0; p.cstats = nullptr;
  int cfg = pick_cfg(M, N, geglu);
  // narrow outputs of 257 .. 512 tiles with a long K loop: 256 x 128 tiles with three buffers instead of two 128 x 128 workgroups per CU
  // (same 128-column statistics partials; tools/bench_gemm.py mid_m "narrow": 4096 x 1280 x 5120 64.7 -> 58.5 us, 12288 x 640 x 2560 44.8 -> 40.9,
  // K = 1280 22.1 -> 21.2, K = 640 equal).  Decided here, not in pick_cfg: the tile width the host sizes buffers from does not change.
  if (cfg == 2 && K >= 1280 && cd360_tune().gemm_cfg < 1 && cd360_tune().gemm_small != 0 && ((M + 127) / 128) * ((N + 127) / 128) <= 512) cfg = 5;
  // 256 x 256 tiles as four waves of 128 x 128 on the generated instruction stream (gemm4w_loop.inc).  Measured against the sixteen- /
  // eight-wave arrangements on one box, interleaved (tools/probe/gemm4w_ab.py, us): plain 8192^3 771 / 875, 4096^3 106 / 116; FF1 + GEGLU
  // 3072 x 10240 x 1280 84.0 / 86.0 as a chain of itself -- but inside the captured denoise step (cold weights, a neighbour's tail)
  // the sixty FF1 launches come out 0.2 % SLOWER on it (32.2 against 32.3 steps/s, three alternations on one box).  So by default it
  // serves long plain K loops only; cd360_tuning.gemm_asm4 = 1 puts every 256 x 256 launch on it, 0 none.
  if ((cfg == 7 || cfg == 3) && cd360_tune().gemm_cfg < 1) {
    const int a4 = cd360_tune().gemm_asm4;
    if (a4 > 0 || (a4 < 0 && !geglu && K >= 4096 && M >= 2048)) cfg = 9;
  }
  switch (cfg) {
    case 1: return launch<2, 2, 2, 2, 2>(p, (hipStream_t)stream);
    case 2: return geglu ? CD360_ERR_SHAPE : launch_epi<2, 4, 1, 2, 2, 0>(p, (hipStream_t)stream);
    case 4: return geglu ? CD360_ERR_SHAPE : launch_128x4<0>(p, (hipStream_t)stream);
    case 8: return geglu ? CD360_ERR_SHAPE : launch_64x4<0>(p, (hipStream_t)stream);
    case 5: return launch<4, 2, 2, 2, 3>(p, (hipStream_t)stream);
    case 7: return launch<4, 4, 2, 2, 2>(p, (hipStream_t)stream);  // 256 x 256 as sixteen waves of 64 x 64 (four per SIMD): A/B only
    case 6: return geglu ? CD360_ERR_SHAPE : launch_epi<4, 2, 3, 2, 2, 0>(p, (hipStream_t)stream);
    case 9: return launch<2, 2, 4, 4, 2>(p, (hipStream_t)stream);
    default: return launch<2, 4, 2, 4, 2>(p, (hipStream_t)stream);
  }
}

// cd360_gemm_bf16(a, w, out, ..., bias, res) for an output that a GroupNorm reads next (SpatialTransformer.proj_out + its residual,
// attention.py:880-886, followed by the next ResBlock's in_layers): additionally writes cstats fp32 [M / cd360_gemm_cstats_rows(M, N), N, 2]
// = per slab of cd360_gemm_cstats_rows(M, N) = 64 | 32 rows and channel, (sum, sum of squares) of the stored bf16 outputs -- the `tile_stats`
// of cd360_gn_silu_bf16, like the convolution epilogue's.  CD360_ERR_SHAPE when the tiling chosen for (M, N) writes none (rows == 0).
extern "C" int cd360_gemm_cstats_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                                      const void* bias, const void* res, int64_t ldr, void* cstats, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!a || !w || !out || !cstats || M <= 0 || N <= 0 || K <= 0) return CD360_ERR_ARG;
  if (K % 64 || N % 16 || lda % 8 || ldw % 8 || ldo % 8 || (res && ldr % 8) || lda < K || ldw < K || M % 64) return CD360_ERR_SHAPE;
  if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)out | (uintptr_t)res | (uintptr_t)cstats) % 16 || (uintptr_t)bias % 8) return CD360_ERR_ARG;
  if (M > 0x7fffffffL || (M + 256) * lda * 2 >= (1L << 32) || ((long)N + 256) * ldw * 2 >= (1L << 32)) return CD360_ERR_SHAPE;
  const int cfg = pick_cfg(M, N, false);
  if (cfg != 2 && cfg != 4 && cfg != 8) return CD360_ERR_SHAPE;
  GemmParams p;
  p.a = (const uint16_t*)a; p.w = (const uint16_t*)w; p.out = (uint16_t*)out; p.bias = (const float*)bias; p.res = (const uint16_t*)res;
  p.ln_stats = nullptr; p.wsum = nullptr; p.stats_out = nullptr;
  p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = res ? ldr : 0;
  p.M = (int)M; p.N = N; p.K = K; p.ln_parts = 0; p.ln_dim = 0; p.ln_eps = 0.f; p.geglu = 0;
  p.tiles_m = p.tiles_n = p.group_m = 0;
  p.ak = p.av = nullptr; p.ak_sb = p.ak_sn = p.av_sb = p.av_sn = 0; p.a_nq = p.a_nk = 0; p.a_scale_log2e = 0.f; p.a_dup = p.a_dup_from = 0; p.a_kv8 = nullptr; p.a_kvs = nullptr; p.a_heads = 0;
  p.cv_H = p.cv_W = p.cv_kg = p.cv_up = 0; p.emb = nullptr; p.emb_stride = 0; p.cstats = (float*)cstats;
  if (cfg == 8) return launch_64x4<6>(p, (hipStream_t)stream);
  return cfg == 2 ? launch_epi<2, 4, 1, 2, 2, 6>(p, (hipStream_t)stream) : launch_128x4<6>(p, (hipStream_t)stream);
}

namespace {
// tile dispatch of the fused query projection + attention (qcfg as cd360_tuning.qattn_cfg); the key count picks the epilogue: 2 / 4 / 6
// groups of 16 keys (EPI 2 / 3 / 4), or five (EPI 7: 65 .. 80 keys -- SDXL's 77 text tokens) unless cd360_tuning.qattn_keys16 = 0;
// a packed fp8 K / V image (p.a_kv8, cd360_qproj_attn_fp8_bf16) selects the fp8-MFMA epilogue (EPI 10, 96 key slots)
int qattn_launch(const GemmParams& p, int qcfg, int Nk, bool keys16, hipStream_t stream) {
  if (qcfg == 3) {  // 128 x 128, four waves of 64 x 64, two buffers: two workgroups per CU (one's attention epilogue under the other's K loop)
    if (p.a_kv8) return launch_epi<2, 2, 2, 2, 2, 10>(p, stream);
    if (Nk <= 32) return launch_epi<2, 2, 2, 2, 2, 2>(p, stream);
    if (Nk <= 64) return launch_epi<2, 2, 2, 2, 2, 3>(p, stream);
    if (Nk <= 80 && keys16) return launch_epi<2, 2, 2, 2, 2, 7>(p, stream);
    return launch_epi<2, 2, 2, 2, 2, 4>(p, stream);
  }
  if (qcfg == 4) {  // 256 x 128, eight waves of 64 x 64, two buffers (96 KB: one workgroup per CU, half the K-loop operand bytes per flop of 128 x 128)
    if (p.a_kv8) return launch_epi<4, 2, 2, 2, 2, 10>(p, stream);
    if (Nk <= 32) return launch_epi<4, 2, 2, 2, 2, 2>(p, stream);
    if (Nk <= 64) return launch_epi<4, 2, 2, 2, 2, 3>(p, stream);
    if (Nk <= 80 && keys16) return launch_epi<4, 2, 2, 2, 2, 7>(p, stream);
    return launch_epi<4, 2, 2, 2, 2, 4>(p, stream);
  }
  if (qcfg == 2) {
    if (p.a_kv8) return launch_epi<4, 2, 2, 1, 4, 10>(p, stream);
    if (Nk <= 32) return launch_epi<4, 2, 2, 1, 4, 2>(p, stream);
    if (Nk <= 64) return launch_epi<4, 2, 2, 1, 4, 3>(p, stream);
    if (Nk <= 80 && keys16) return launch_epi<4, 2, 2, 1, 4, 7>(p, stream);
    return launch_epi<4, 2, 2, 1, 4, 4>(p, stream);
  }
  if (p.a_kv8) return launch_epi<2, 4, 2, 4, 2, 10>(p, stream);
  if (Nk <= 32) return launch_epi<2, 4, 2, 4, 2, 2>(p, stream);
  if (Nk <= 64) return launch_epi<2, 4, 2, 4, 2, 3>(p, stream);
  if (Nk <= 80 && keys16) return launch_epi<2, 4, 2, 4, 2, 7>(p, stream);
  return launch_epi<2, 4, 2, 4, 2, 4>(p, stream);
}
}  // namespace

// out[M, N] = softmax_keys((A W^T [LayerNorm-folded] + bias) K_h^T * scale) V_h per head h (N = heads * 64): the query projection of a
// cross-attention over Nk <= 96 keys fused with the attention itself -- Q never exists in memory.  A, W, bias, ln_stats, wsum as in
// cd360_gemm_bf16; k, v bf16 [B, >= Nk, N] (element strides k_sb / k_sn, v_sb / v_sn: batch, key; head h at columns 64 h .. 64 h + 63),
// M = B * Nq with Nq % 128 == 0 (a token tile never straddles two batch elements; 256-token tiles need Nq % 256 == 0).
// `dup` > 0: the last `dup` of the B = M / Nq query batch elements each attend to TWO key / value sets -- k, v then hold B + dup batch
// elements and out (B + dup) Nq rows: query element i >= B - dup writes batch i (keys of batch i) and batch i + dup (keys of batch
// i + dup).  The 3-way CFG batch of sample.py has identical pose tokens in its two image-conditional thirds: q is projected once.
extern "C" int cd360_qproj_attn_dedup_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                                           const void* bias, const void* ln_stats, int ln_parts, int ln_dim, float ln_eps, const void* wsum,
                                           const void* k, const void* v, int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn, int Nq, int Nk,
                                           float scale, int dup, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!a || !w || !out || !k || !v || M <= 0 || N <= 0 || K <= 0 || Nq <= 0 || Nk <= 0 || dup < 0 || (int64_t)dup * Nq > M) return CD360_ERR_ARG;
  if (K % 64 || N % 64 || lda % 8 || ldw % 8 || ldo % 8 || lda < K || ldw < K || Nk > 96 || Nq % 128 || M % Nq) return CD360_ERR_SHAPE;
  // tile (cd360_tuning.qattn_cfg): 1 = 256 tokens x 256 channels (four heads, eight waves of 128 x 64: the FeatureNeRF pose tokens, 10^5
  // rows), 2 = 128 x 128 (two heads, eight waves of 32 x 64 + four mover waves, four LDS buffers), 4 = 256 x 128 (eight waves of 64 x 64) --
  // the text cross-attention of every block, M = 3072 ... 12288, where 256 x 256 tiles would leave most of the 256 CUs idle; 3 = 128 x 128
  // as four waves of 64 x 64 with two buffers (two workgroups per CU: A/B only)
  // measured (tools/bench_gemm.py qattn, b = 3): the largest tile that still gives every CU a workgroup wins -- pose tokens 256 x 256
  // (479 / 297 us at the 640 / 1280 level; 128 x 128 with two workgroups per CU 474 / 345: one's attention epilogue under the other's K
  // loop buys nothing), text tokens of the 640 level 256 x 128 (22 us against 28 on 128 x 128), of the 1280 level 128 x 128 (20 against 31)
  int qcfg = cd360_tune().qattn_cfg;
  if (qcfg < 1 || qcfg > 4) qcfg = (M / 256) * ((N + 255) / 256) >= 200 ? 1 : ((M / 256) * ((N + 127) / 128) >= 200 ? 4 : 2);
  if (Nq % 256 && (qcfg == 1 || qcfg == 4)) qcfg = 2;
  if (k_sb % 8 || k_sn % 8 || v_sb % 8 || v_sn % 8 || k_sn < N || v_sn < N) return CD360_ERR_SHAPE;
  if ((96 * k_sn + N) * 2 >= (1L << 31) || (96 * v_sn + N) * 2 >= (1L << 31)) return CD360_ERR_SHAPE;  // 32-bit offsets inside one batch element's K / V
  if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)out | (uintptr_t)k | (uintptr_t)v) % 16) return CD360_ERR_ARG;
  if (((uintptr_t)bias | (uintptr_t)ln_stats | (uintptr_t)wsum) % 8) return CD360_ERR_ARG;
  if (M > 0x7fffffffL || (M + 256) * lda * 2 >= (1L << 32) || ((long)N + 256) * ldw * 2 >= (1L << 32)) return CD360_ERR_SHAPE;
  if (ln_stats && (!wsum || ln_parts <= 0 || ln_dim <= 0)) return CD360_ERR_ARG;
  GemmParams p;
  p.a = (const uint16_t*)a; p.w = (const uint16_t*)w; p.out = (uint16_t*)out; p.bias = (const float*)bias; p.res = nullptr;
  p.ln_stats = (const float*)ln_stats; p.wsum = (const float*)wsum; p.stats_out = nullptr;
  p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = 0;
  p.M = (int)M; p.N = N; p.K = K; p.ln_parts = ln_parts; p.ln_dim = ln_dim; p.ln_eps = ln_eps; p.geglu = 0;
  p.tiles_m = p.tiles_n = p.group_m = 0;
  p.ak = (const uint16_t*)k; p.av = (const uint16_t*)v; p.ak_sb = k_sb; p.ak_sn = k_sn; p.av_sb = v_sb; p.av_sn = v_sn;
  p.a_nq = Nq; p.a_nk = Nk; p.a_scale_log2e = scale * 1.4426950408889634f;
  p.a_dup = dup; p.a_dup_from = (int)(M / Nq) - dup; p.a_kv8 = nullptr; p.a_kvs = nullptr; p.a_heads = N / 64;
  p.cv_H = p.cv_W = p.cv_kg = p.cv_up = 0; p.emb = nullptr; p.emb_stride = 0; p.cstats = nullptr;
  const bool keys16 = cd360_tune().qattn_keys16 != 0;
  // A width that is 128 short of a multiple of 256 (SDXL's 640 = 2.5 tiles) leaves half of the last 256-column tile's waves without
  // channels while the workgroup still holds its CU.  Sending the last 128 columns (two heads) to the 256 x 128 tile in a second launch
  // was measured and LOSES (pose tokens of the 640 level, b = 3: 428 us in one launch, 450-464 split; de-duplicated 350 / 356): two
  // launches have two ragged last rounds and the narrow tile is slower per column.  cd360_tuning.qattn_split = 1 keeps the A/B.
  if (qcfg == 1 && N > 256 && N % 256 == 128 && cd360_tune().qattn_split > 0) {
    GemmParams p1 = p;
    p1.N = N - 128;
    const int r1 = qattn_launch(p1, 1, Nk, keys16, (hipStream_t)stream);
    if (r1 != CD360_OK) return r1;
    GemmParams p2 = p;
    const long c0 = N - 128;
    p2.N = 128;
    p2.w += c0 * ldw; p2.out += c0; p2.ak += c0; p2.av += c0;
    if (p2.bias) p2.bias += c0;
    if (p2.wsum) p2.wsum += c0;
    return qattn_launch(p2, 4, Nk, keys16, (hipStream_t)stream);
  }
  return qattn_launch(p, qcfg, Nk, keys16, (hipStream_t)stream);
}

// ---- BASELINE configs[4]: the fused cross-attention with its two contractions (q K^T and P V) on fp8 MFMA --------------------------
// K and V of a cross-attention over <= 96 keys are constant for a whole trajectory (the text context), so they are quantised ONCE per
// image by cd360_kv_pack_fp8 into the image the kernel's LDS wants (OCP e4m3, one scale per tensor and head = amax / 448):
//   per (batch, head) 14336 bytes = K as 96 key rows of 64 bytes (rows >= Nk zero; byte 32 hh + e of a row = channel 16 hh + e for
//   e < 16, 32 + 16 hh + (e - 16) otherwise -- the order in which a lane of the projection's accumulator holds its 32 channels, i.e. the
//   k-slot order of the B operand; 16-byte chunks XOR (row >> 2) & 3) followed by V^T as 64 channel rows of 128 key slots (slot
//   64 m + 32 hh + e of row d = V[key][d] with key = 32 (2 m + (e >> 4)) + (e & 3) + 8 ((e & 15) >> 2) + 4 hh, the key a lane's score
//   register holds; keys >= Nk zero; chunks XOR (d >> 1) & 7): both are the A operands of v_mfma_scale_f32_32x32x64_f8f6f4 read with
//   two conflict-free ds_read_b128 per fragment, and the LDS-DMA copies the image as it lies.
// In the kernel (EPI 10) the projected q is scaled per TOKEN (max |q| over the 64 channels of the head / 448), P is written as
// p * 2^8, and the scales are applied to the fp32 scores / outputs -- see attn_tile.  Same envelope as cd360_qproj_attn_dedup_bf16.
namespace {
__global__ __launch_bounds__(256) void kv_pack_fp8_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, unsigned char* __restrict__ kv8,
                                                          float* __restrict__ scales, int H, int Nk, long k_sb, long k_sn, long v_sb, long v_sn) {
  const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  const uint16_t* kb = k + b * k_sb + h * 64;
  const uint16_t* vb = v + b * v_sb + h * 64;
  __shared__ float red[2][4];
  float ak = 0.f, av = 0.f;
  for (int i = tid; i < Nk * 64; i += 256) {
    const int key = i >> 6, c = i & 63;
    ak = fmaxf(ak, fabsf(bf16_to_f32(kb[key * k_sn + c])));
    av = fmaxf(av, fabsf(bf16_to_f32(vb[key * v_sn + c])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ak = fmaxf(ak, __shfl_xor(ak, o));
    av = fmaxf(av, __shfl_xor(av, o));
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = ak;
    red[1][tid >> 6] = av;
  }
  __syncthreads();
  ak = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  av = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  const float sk = ak > 0.f ? ak / 448.f : 1.f, sv = av > 0.f ? av / 448.f : 1.f;
  const float ik = 1.f / sk, iv = 1.f / sv;
  if (tid == 0) {
    scales[(long)blockIdx.x * 2] = sk;
    scales[(long)blockIdx.x * 2 + 1] = sv;
  }
  uint32_t* out = reinterpret_cast<uint32_t*>(kv8 + (long)blockIdx.x * (96 * 64 + 64 * 128));
  for (int dw = tid; dw < 96 * 16; dw += 256) {  // K: physical dword dw of the image
    const int row = dw >> 4, pd = dw & 15, lc = (pd >> 2) ^ ((row >> 2) & 3);
    float x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pos = lc * 16 + (pd & 3) * 4 + t, hh = pos >> 5, e = pos & 31;
      const int c = e < 16 ? 16 * hh + e : 32 + 16 * hh + (e - 16);
      x[t] = row < Nk ? bf16_to_f32(kb[row * k_sn + c]) * ik : 0.f;
    }
    int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
    out[dw] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w8, true);
  }
  for (int dw = tid; dw < 64 * 32; dw += 256) {  // V^T
    const int d = dw >> 5, pd = dw & 31, lc = (pd >> 2) ^ ((d >> 1) & 7);
    float x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pos = lc * 16 + (pd & 3) * 4 + t, m = pos >> 6, hh = (pos >> 5) & 1, e = pos & 31;
      const int key = 32 * (2 * m + (e >> 4)) + (e & 3) + 8 * ((e & 15) >> 2) + 4 * hh;
      x[t] = key < Nk ? bf16_to_f32(vb[key * v_sn + d]) * iv : 0.f;
    }
    int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
    out[96 * 16 + dw] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w8, true);
  }
}
}  // namespace

extern "C" int64_t cd360_kv_fp8_bytes(int B, int H) { return (int64_t)B * H * (96 * 64 + 64 * 128); }

// k, v bf16 [B, >= Nk, H * 64] (element strides k_sb / k_sn, v_sb / v_sn) -> kv8 (cd360_kv_fp8_bytes(B, H) bytes, 16-byte aligned),
// scales fp32 [B, H, 2] = (K scale, V scale) per head; Nk <= 96
extern "C" int cd360_kv_pack_fp8(const void* k, const void* v, void* kv8, void* scales, int B, int H, int Nk, int64_t k_sb, int64_t k_sn,
                                 int64_t v_sb, int64_t v_sn, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!k || !v || !kv8 || !scales || B <= 0 || H <= 0 || Nk <= 0) return CD360_ERR_ARG;
  if (Nk > 96 || (uintptr_t)kv8 % 16 || (uintptr_t)scales % 4) return CD360_ERR_SHAPE;
  hipLaunchKernelGGL(kv_pack_fp8_kernel, dim3((unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)k, (const uint16_t*)v,
                     (unsigned char*)kv8, (float*)scales, H, Nk, (long)k_sb, (long)k_sn, (long)v_sb, (long)v_sn);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// cd360_qproj_attn_dedup_bf16 with K / V given as the packed fp8 image of cd360_kv_pack_fp8 (kv8, scales: B + dup batch elements,
// N / 64 heads): out[M, N] = softmax_keys(q K_h^T * scale) V_h with both contractions on fp8 MFMA, fp32 accumulation, bf16 output.
extern "C" int cd360_qproj_attn_fp8_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                                         const void* bias, const void* ln_stats, int ln_parts, int ln_dim, float ln_eps, const void* wsum,
                                         const void* kv8, const void* scales, int Nq, int Nk, float scale, int dup, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!a || !w || !out || !kv8 || !scales || M <= 0 || N <= 0 || K <= 0 || Nq <= 0 || Nk <= 0 || dup < 0 || (int64_t)dup * Nq > M) return CD360_ERR_ARG;
  if (K % 64 || N % 64 || lda % 8 || ldw % 8 || ldo % 8 || lda < K || ldw < K || Nk > 96 || Nq % 128 || M % Nq) return CD360_ERR_SHAPE;
  if (Nk <= 64) return CD360_ERR_SHAPE;  // the fp8 epilogue masks the padding keys of its LAST 32-key block only: 65 .. 96 keys (SDXL: 77)
  int qcfg = cd360_tune().qattn_cfg;
  if (qcfg < 1 || qcfg > 4) qcfg = (M / 256) * ((N + 255) / 256) >= 200 ? 1 : ((M / 256) * ((N + 127) / 128) >= 200 ? 4 : 2);
  if (Nq % 256 && (qcfg == 1 || qcfg == 4)) qcfg = 2;
  if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)out | (uintptr_t)kv8) % 16 || (uintptr_t)scales % 4) return CD360_ERR_ARG;
  if (((uintptr_t)bias | (uintptr_t)ln_stats | (uintptr_t)wsum) % 8) return CD360_ERR_ARG;
  if (M > 0x7fffffffL || (M + 256) * lda * 2 >= (1L << 32) || ((long)N + 256) * ldw * 2 >= (1L << 32)) return CD360_ERR_SHAPE;
  if (ln_stats && (!wsum || ln_parts <= 0 || ln_dim <= 0)) return CD360_ERR_ARG;
  GemmParams p;
  p.a = (const uint16_t*)a; p.w = (const uint16_t*)w; p.out = (uint16_t*)out; p.bias = (const float*)bias; p.res = nullptr;
  p.ln_stats = (const float*)ln_stats; p.wsum = (const float*)wsum; p.stats_out = nullptr;
  p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = 0;
  p.M = (int)M; p.N = N; p.K = K; p.ln_parts = ln_parts; p.ln_dim = ln_dim; p.ln_eps = ln_eps; p.geglu = 0;
  p.tiles_m = p.tiles_n = p.group_m = 0;
  p.ak = p.av = nullptr; p.ak_sb = p.ak_sn = p.av_sb = p.av_sn = 0;
  p.a_nq = Nq; p.a_nk = Nk; p.a_scale_log2e = scale * 1.4426950408889634f;
  p.a_dup = dup; p.a_dup_from = (int)(M / Nq) - dup; p.a_kv8 = (const unsigned char*)kv8; p.a_kvs = (const float*)scales; p.a_heads = N / 64;
  p.cv_H = p.cv_W = p.cv_kg = p.cv_up = 0; p.emb = nullptr; p.emb_stride = 0; p.cstats = nullptr;
  return qattn_launch(p, qcfg, Nk, true, (hipStream_t)stream);
}

extern "C" int cd360_qproj_attn_bf16(const void* a, const void* w, void* out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldo,
                                     const void* bias, const void* ln_stats, int ln_parts, int ln_dim, float ln_eps, const void* wsum,
                                     const void* k, const void* v, int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn, int Nq, int Nk,
                                     float scale, void* stream) {
  CD360_TUNE_SCOPE(stream);
  return cd360_qproj_attn_dedup_bf16(a, w, out, M, N, K, lda, ldw, ldo, bias, ln_stats, ln_parts, ln_dim, ln_eps, wsum, k, v, k_sb, k_sn, v_sb, v_sn,
                                     Nq, Nk, scale, 0, stream);
}

// ---- 3 x 3 / stride 1 / pad 1 convolution on the same core (EPI 5): the im2col matrix is never built -- K-tile kt of the A operand is
// the 64-channel chunk of the input pixel shifted by the tile's tap, fetched by the same LDS-DMA pieces with a wave-uniform offset added
// and padding neighbours redirected past the end of the buffer descriptor (zeros).  Replaces conv_igemm_kernel (register-staged
// 128 x 128 tiles) for the ResBlock / Upsample convolutions (openaimodel.py:161-164, 352-376).
namespace {
// tilings (pixels x channels): 1 = 256 x 320 (Cout % 320 == 0: the 128^2 level, one channel tile), 2 = 256 x 128 / 3 buffers,
// 3 = 256 x 256, 4 = 128 x 128 / 4 buffers (the 32^2 level: 240 tiles)
// 5 = 192 x 320 (round 5; six waves of 64 x 160): the 320-channel convolutions at 128^2, M = 49152 = 256 tiles of 192 -- one per CU, where
// tiling 1 leaves a quarter of the CUs without a workgroup (192 tiles)
// (six waves put two on two SIMDs and one on the others); 6 = the same tile as twelve waves of 32 x 160 (three per SIMD)
constexpr int NCONV = 6;
constexpr int CONV_BM[NCONV + 1] = {0, 256, 256, 256, 128, 192, 192}, CONV_BN[NCONV + 1] = {0, 320, 128, 256, 128, 320, 320},
              CONV_SLAB[NCONV + 1] = {0, 64, 64, 128, 64, 64, 32};
int pick_conv_cfg(long M, int Cout) {
  {
    const int c = cd360_tune().conv_cfg;
    if (c >= 1 && c <= NCONV && !((c == 1 || c >= 5) && Cout % 320)) return c;
  }
  // relative speed of the tilings on full tiles (tools/bench_kernels.py conv / conv_tilings).  Round 5, same box, interleaved, us (tiling 1 /
  // 5 / 6): 320 -> 320 at 128^2 95.5 / 94.9 / 91.6, 640 -> 320 175.6 / 174.5 / 169.0, 960 -> 320 256.6 / 257.6 / 247.7, 640 -> 640 351.6 /
  // 337.2 / 331.4, outputs bit-identical: six waves (5) put two waves on two of the SIMDs and gain nothing from the 256 busy CUs; twelve
  // (6) finish a 192-row tile in 0.96 of a 256-row tile's time = 0.78 of tiling 1 per row, which beats its 0.75 CU fill.  5 stays A/B only.
  static const double weight[NCONV + 1] = {0, 1.0, 0.93, 1.0, 0.85, 0.0, 0.78};  // relative speed of the tilings on full tiles (tools/bench_kernels.py conv)
  int best = 4;
  double best_eff = -1.0;
  for (int c = 1; c <= NCONV; ++c) {
    if ((c == 1 || c >= 5) && Cout % 320) continue;
    const long tm = (M + CONV_BM[c] - 1) / CONV_BM[c], tn = (Cout + CONV_BN[c] - 1) / CONV_BN[c], nwg = tm * tn, rounds = (nwg + 255) / 256;
    const double eff = (double)nwg / (double)(rounds * 256) * (double)Cout / (double)(tn * CONV_BN[c]) * weight[c];
    if (eff > best_eff) { best_eff = eff; best = c; }
  }
  return best;
}
// The halo form of the 3 x 3 convolution (gemm_mfma_kernel EPI 11: 128 x 128 tiles of whole image rows, the tile's input pixels fetched once
// per 64-channel chunk): image rows that divide the 128-pixel tile, images that are whole tiles, a halo image of at most nine 32-row
// pieces -- W = 8 ... 64, i.e. the 32^2 and 64^2 levels of the SDXL UNet.  cd360_tuning.conv_halo: 0 never, 1 wherever it fits, -1 = the
// measured default (launches whose tiling is 128 x 128: the 32^2 level).
bool conv_halo_ok(int N, int H, int W, int Cin, int Cout, int cfg) {
  const int mode = cd360_tune().conv_halo;
  if (mode == 0) return false;
  if (W < 8 || 128 % W || ((long)H * W) % 128 || (128 / W + 2) * (W + 2) > 9 * 32 || (128 / W + 2) * (W + 2) >= 1024) return false;
  if (Cin < 64 || 9 * Cin < 64 * 4) return false;
  return mode == 1 || cfg == 4;  // (measured, tools/probe/conv_halo_ab.py: 32^2 level -3 ... -7 %; the 64^2 level's 256 x 128 tiling beats 128 x 128 halo tiles by 15 %)
}
bool conv_dma_ok(int N, int H, int W, int Cin, int Cout, int taps, int stride) {
  if (cd360_tune().conv_dma == 0) return false;
  if (taps != 9 || stride != 1 || Cin % 64 || Cout % 16 || N <= 0 || H <= 0 || W <= 0) return false;
  const long M = (long)N * H * W;
  return M * Cin * 2 < (1L << 31) && ((long)Cout + 320) * 9 * Cin * 2 < (1L << 32);
}
}  // namespace

extern "C" int cd360_conv_k_order(int Cin, int taps);

// Pixels per slab of the `tile_stats` output of cd360_conv_igemm_bf16 for this convolution (the launch's wave tiling decides), or 0
// when the call is served by the register-staged kernel (then 128 / cd360_conv_stats_slabs(Cout)).
extern "C" int cd360_conv_dma_slab_rows(int N, int H, int W, int Cin, int Cout, int taps, int stride) {
  if (!conv_dma_ok(N, H, W, Cin, Cout, taps, stride)) return 0;
  const int rows = CONV_SLAB[pick_conv_cfg((long)N * H * W, Cout)];
  return ((long)H * W) % rows ? 0 : rows;
}

// Same contract as cd360_conv_igemm_bf16 for taps = 9, stride = 1; tile_stats fp32 [N H W / cd360_conv_dma_slab_rows(...), Cout, 2].
// CD360_ERR_SHAPE when the shape is outside the envelope (the caller then uses the register-staged kernel).
extern "C" int cd360_conv3x3_dma_bf16(const void* x, const void* w_packed, const void* bias, const void* emb, int64_t emb_stride, const void* res,
                                      void* out, int N, int H, int W, int Cin, int Cout, void* tile_stats, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!x || !w_packed || !out) return CD360_ERR_ARG;
  if (!conv_dma_ok(N, H, W, Cin, Cout, 9, 1)) return CD360_ERR_SHAPE;
  if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)out | (uintptr_t)emb | (uintptr_t)res | (uintptr_t)tile_stats) % 16 || (uintptr_t)bias % 8) return CD360_ERR_ARG;
  if (emb && (emb_stride < Cout || emb_stride % 8)) return CD360_ERR_SHAPE;
  const long M = (long)N * H * W;
  const int cfg = pick_conv_cfg(M, Cout);
  if (tile_stats && ((long)H * W) % CONV_SLAB[cfg]) return CD360_ERR_SHAPE;
  GemmParams p;
  p.a = (const uint16_t*)x; p.w = (const uint16_t*)w_packed; p.out = (uint16_t*)out; p.bias = (const float*)bias; p.res = (const uint16_t*)res;
  p.ln_stats = nullptr; p.wsum = nullptr; p.stats_out = nullptr;
  p.lda = Cin; p.ldw = 9L * Cin; p.ldo = Cout; p.ldr = res ? Cout : 0;
  p.M = (int)M; p.N = Cout; p.K = 9 * Cin; p.ln_parts = 0; p.ln_dim = 0; p.ln_eps = 0.f; p.geglu = 0;
  p.tiles_m = p.tiles_n = p.group_m = 0;
  p.ak = p.av = nullptr; p.ak_sb = p.ak_sn = p.av_sb = p.av_sn = 0; p.a_nq = p.a_nk = 0; p.a_scale_log2e = 0.f; p.a_dup = p.a_dup_from = 0; p.a_kv8 = nullptr; p.a_kvs = nullptr; p.a_heads = 0;
  p.cv_H = H; p.cv_W = W; p.cv_kg = cd360_conv_k_order(Cin, 9); p.cv_up = 0;
  p.emb = (const uint16_t*)emb; p.emb_stride = emb ? emb_stride : 0; p.cstats = (float*)tile_stats;
  if (conv_halo_ok(N, H, W, Cin, Cout, cfg)) return launch_128x4<11>(p, (hipStream_t)stream);
  switch (cfg) {
    case 1: return launch_epi<4, 2, 5, 2, 2, 5>(p, (hipStream_t)stream);
    case 2: return launch_epi<4, 2, 2, 2, 3, 5>(p, (hipStream_t)stream);
    case 3: return launch_epi<2, 4, 2, 4, 2, 5>(p, (hipStream_t)stream);
    case 5: return launch_epi<3, 2, 5, 2, 2, 5>(p, (hipStream_t)stream);
    case 6: return launch_epi<6, 2, 5, 1, 2, 5>(p, (hipStream_t)stream);
    default: return launch_128x4<5>(p, (hipStream_t)stream);
  }
}

// Upsample.forward (openaimodel.py:114-181): nearest-neighbour 2x interpolation followed by conv3x3 / pad 1, as ONE launch that never
// builds the upsampled image.  Output pixel (2 i + a, 2 j + b) only sees the 2 x 2 source pixels {i + a - 1, i + a} x {j + b - 1, j + b}:
// three of the nine taps of each row / column pair read the same source pixel, so phase (a, b) is a 2 x 2-tap convolution of the SOURCE
// image with the coinciding taps' weights summed -- 4 / 9 of the multiply-adds and no 4x-sized intermediate.  x [N, H, W, Cin] bf16
// channels-last; w_phases [4, Cout, 4 Cin] bf16: phase 2 a + b, K order of cd360_conv_k_order(Cin, 9) with tap slot t = 2 ty + tx
// (cd360.ops.pack_upsample_conv_weight); bias fp32 [Cout] | NULL; out [N, 2H, 2W, Cout] bf16.  Cin % 64 == 0, Cout % 16 == 0.
extern "C" int cd360_conv_up2x_bf16(const void* x, const void* w_phases, const void* bias, void* out, int N, int H, int W, int Cin, int Cout,
                                    void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!x || !w_phases || !out || N <= 0 || H <= 0 || W <= 0) return CD360_ERR_ARG;
  if (Cin % 64 || Cout % 16) return CD360_ERR_SHAPE;
  if (((uintptr_t)x | (uintptr_t)w_phases | (uintptr_t)out) % 16 || (uintptr_t)bias % 8) return CD360_ERR_ARG;
  const long M = (long)N * H * W;
  if (M * Cin * 2 >= (1L << 31) || ((long)4 * Cout + 320) * 4 * Cin * 2 >= (1L << 32) || 4 * M * Cout * 2 >= (1L << 40)) return CD360_ERR_SHAPE;
  GemmParams p;
  p.a = (const uint16_t*)x; p.w = (const uint16_t*)w_phases; p.out = (uint16_t*)out; p.bias = (const float*)bias; p.res = nullptr;
  p.ln_stats = nullptr; p.wsum = nullptr; p.stats_out = nullptr;
  p.lda = Cin; p.ldw = 4L * Cin; p.ldo = Cout; p.ldr = 0;
  p.M = (int)M; p.N = Cout; p.K = 4 * Cin; p.ln_parts = 0; p.ln_dim = 0; p.ln_eps = 0.f; p.geglu = 0;
  p.tiles_m = p.tiles_n = p.group_m = 0;
  p.ak = p.av = nullptr; p.ak_sb = p.ak_sn = p.av_sb = p.av_sn = 0; p.a_nq = p.a_nk = 0; p.a_scale_log2e = 0.f; p.a_dup = p.a_dup_from = 0; p.a_kv8 = nullptr; p.a_kvs = nullptr; p.a_heads = 0;
  p.cv_H = H; p.cv_W = W; p.cv_kg = cd360_conv_k_order(Cin, 9); p.cv_up = 1;
  p.emb = nullptr; p.emb_stride = 0; p.cstats = nullptr;
  switch (pick_conv_cfg(4 * M, Cout)) {  // the four phases share the launch: tile count of the full-resolution output
    case 1: return launch_epi<4, 2, 5, 2, 2, 5>(p, (hipStream_t)stream);
    case 2: return launch_epi<4, 2, 2, 2, 3, 5>(p, (hipStream_t)stream);
    case 3: return launch_epi<2, 4, 2, 4, 2, 5>(p, (hipStream_t)stream);
    case 5: return launch_epi<3, 2, 5, 2, 2, 5>(p, (hipStream_t)stream);
    case 6: return launch_epi<6, 2, 5, 1, 2, 5>(p, (hipStream_t)stream);
    default: return launch_128x4<5>(p, (hipStream_t)stream);
  }
}

// Per-row (sum, sumsq) of a bf16 [rows, C] matrix (row stride ld) as ONE partial per row: the `ln_stats` input of cd360_gemm_bf16 for a
// tensor that did not come out of a stats-writing GEMM (the first block of a SpatialTransformer, the FeatureNeRF pose tokens).
namespace {
__global__ __launch_bounds__(256) void row_stats_kernel(const uint16_t* __restrict__ x, float* __restrict__ st, long rows, int C, long ld) {
  const int lane = threadIdx.x & 63;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long row = wave0; row < rows; row += nwaves) {
    const uint16_t* src = x + row * ld;
    float s = 0.f, q = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a0 = bf16lo_to_f32(v[e]), a1 = bf16hi_to_f32(v[e]);
        s += a0 + a1;
        q = fmaf(a0, a0, fmaf(a1, a1, q));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      q += __shfl_xor(q, o);
    }
    if (lane == 0) {
      st[row * 2] = s;
      st[row * 2 + 1] = q;
    }
  }
}
}  // namespace

extern "C" int cd360_row_stats_bf16(const void* x, void* stats, int64_t rows, int C, int64_t ld, void* stream) {
  CD360_TUNE_SCOPE(stream);
  if (!x || !stats || rows <= 0 || C <= 0) return CD360_ERR_ARG;
  if (C % 8 || ld % 8 || ld < C || (uintptr_t)x % 16) return CD360_ERR_SHAPE;
  const long blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                     (float*)stats, (long)rows, C, (long)ld);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}
