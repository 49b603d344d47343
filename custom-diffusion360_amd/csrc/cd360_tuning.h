// Tuning / A-B state of libcd360_hip.so (declared publicly in include/cd360_hip.h: cd360_tuning, cd360_set_tuning, cd360_get_tuning,
// cd360_set_stream_tuning).  The launch functions read it through cd360_tune(): the override of the stream they were called with, else
// the process-wide default; nothing in a launch path calls getenv.  Every field is
// -1 by default = "the measured best for the shape"; the Python binding fills the struct once from the CD360_* environment variables
// when it loads the library (cd360/_lib.py) and tools / tests change it explicitly (cd360.ops.tuning(...)).
#pragma once
#include <stdint.h>

extern "C" {
typedef struct cd360_tuning {
  int32_t size;             // sizeof(cd360_tuning): ABI check of cd360_set_tuning
  int32_t gemm_cfg;         // 1..8: tiling of cd360_gemm_bf16 (gemm8p.hip: pick_cfg)
  int32_t gemm_group_m;     // > 0: token tiles per tile group
  int32_t gemm_movers;      // 0 | 4: mover waves off / on wherever the arrangement can take them
  int32_t gemm_ksplit;      // 0 | 1 | 2: wave arrangement of the 128 x 128 four-buffer tiling
  int32_t conv_cfg;         // 1..6: tiling of cd360_conv3x3_dma_bf16
  int32_t conv_dma;         // 0: 3 x 3 / 1 x 1 convolutions on the register-staged kernel
  int32_t conv_kgroup;      // > 0: K-order group size (set BEFORE weights are packed; process-wide: cd360_set_stream_tuning rejects another value)
  int32_t conv_wide;        // 0: register-staged kernel never uses its 160-channel tiles
  int32_t conv_wmajor;      // 0 | 1: register-staged kernel's tile order
  int32_t conv_split;       // 1 | 2: register-staged kernel's in-workgroup split-K
  int32_t attn_smallk;      // 0: <= 96-key attention on the tiled kernel
  int32_t attn_smallk_wgs;  // > 0: workgroup target of the register-resident kernel
  int32_t attn_self;        // 0 | 1 | 2 | 3: self-attention kernel generation / tiling
  int32_t attn_fast;        // 0: guarded (masked) path of the first-generation kernel
  int32_t nerf_kernel;      // 0 | 1: render kernel with register gathers / full-line gathers through the wave's LDS block
  int32_t qattn_cfg;        // 1..4: tile of cd360_qproj_attn_bf16 (256 x 256 / 128 x 128 + movers / 128 x 128 x 2 WGs / 256 x 128)
  int32_t whatif;           // what-if timing bits of the GEMM core: honoured by -DCD360_WHATIF builds only (results are wrong when set)
  int32_t gemm_small;       // 0: pick_cfg without its small-batch rules (64 x 128 tiles; 128-wide tiles for wide outputs of <= 512 tiles)
  int32_t qattn_keys16;     // 0: cd360_qproj_attn_bf16 pads 65 .. 80 keys to three full 32-key blocks (96) instead of five groups of 16
  int32_t qattn_split;      // 1: second launch for the last 128 columns of a width that is 128 short of a multiple of 256 (A/B: slower)
  int32_t store_wt;         // 0 | 1: output tiles of the GEMM family leave by plain stores / by write-through (sc1) stores (gemm8p.hip: store_tile)
  int32_t conv_halo;        // 0 | 1: the halo form of the 3 x 3 convolution never / wherever it fits (-1: the launches it was measured to help)
  int32_t gemm_asm4;        // 0 | 1: the four-wave 256 x 256 arrangement on the generated instruction stream (gemm4w_loop.inc) never / for every 256 x 256 launch (-1: where measured: FF1 + GEGLU with K >= 1024)
  int32_t reserved[2];
} cd360_tuning;

int cd360_set_tuning(const cd360_tuning* t);
int cd360_get_tuning(cd360_tuning* t);
int cd360_set_stream_tuning(void* stream, const cd360_tuning* t);
int cd360_get_stream_tuning(void* stream, cd360_tuning* t);
int cd360_query_stream(void* stream);
}

// The tuning a launch function reads: the override of the stream the running entry point was called with (CD360_TUNE_SCOPE), else the
// process-wide default (tuning.hip).
const cd360_tuning& cd360_tune();
// The process-wide default regardless of stream: for pack-time choices (conv_kgroup), which cannot differ per stream.
const cd360_tuning& cd360_tune_default();

// First statement of every entry point that takes a stream and reads the tuning: for the duration of the call cd360_tune() answers
// with that stream's override, if one was set (cd360_set_stream_tuning).  Thread-local, nothing shared is written; with no per-stream
// override registered anywhere (the normal case) the constructor is one relaxed atomic load.
class Cd360TuneScope {
 public:
  explicit Cd360TuneScope(void* stream);
  ~Cd360TuneScope();
  Cd360TuneScope(const Cd360TuneScope&) = delete;
  Cd360TuneScope& operator=(const Cd360TuneScope&) = delete;

 private:
  const cd360_tuning* prev_;
  bool set_;
  cd360_tuning local_;
};
#define CD360_TUNE_SCOPE(stream) Cd360TuneScope cd360_tune_scope_((void*)(stream))
