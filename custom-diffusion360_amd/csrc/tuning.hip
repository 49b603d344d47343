// The tuning / A-B struct (cd360_tuning.h): a process-wide DEFAULT (cd360_set_tuning) and per-stream overrides (cd360_set_stream_tuning)
// that the launch functions read through cd360_tune() for the stream they were called with.  Defaults (-1 everywhere) select the measured-best kernel for every shape; nothing here changes
// results except `whatif`, which only a -DCD360_WHATIF probe build honours.
#include "cd360_common.h"
#include "cd360_tuning.h"
#include <string.h>

#include <atomic>
#include <mutex>

namespace {
cd360_tuning make_default() {
  cd360_tuning t;
  memset(&t, 0xff, sizeof(t));  // every int32 field = -1
  t.size = (int32_t)sizeof(cd360_tuning);
  return t;
}
cd360_tuning g_tuning = make_default();  // the process-wide default

// per-stream overrides: a small table behind a mutex, consulted only while it is non-empty
struct StreamTune {
  void* stream;
  cd360_tuning t;
};
constexpr int MAX_STREAMS = 16;
StreamTune g_streams[MAX_STREAMS];
std::atomic<int> g_nstreams{0};
std::mutex g_mu;
thread_local const cd360_tuning* tl_tune = nullptr;   // the override in force on this thread (inside an entry point, or after cd360_query_stream)
thread_local cd360_tuning tl_query;                    // copy behind tl_tune between cd360_query_stream calls

bool lookup(void* stream, cd360_tuning* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int n = g_nstreams.load(std::memory_order_relaxed);
  for (int i = 0; i < n; ++i)
    if (g_streams[i].stream == stream) {
      *out = g_streams[i].t;
      return true;
    }
  return false;
}

int sanitize(const cd360_tuning* t, cd360_tuning* out) {
  if (t->size != (int32_t)sizeof(cd360_tuning)) return CD360_ERR_ARG;
  *out = *t;
#ifndef CD360_WHATIF
  out->whatif = -1;
#endif
  return CD360_OK;
}
}  // namespace

// (a query context left on some thread by cd360_query_stream is honoured only while overrides exist at all)
// Pack-time choices (conv_kgroup: the K order of a packed convolution weight) are process-wide by nature: packing and every later
// launch must agree whatever stream they run on, so they are read from the default only
const cd360_tuning& cd360_tune_default() { return g_tuning; }

const cd360_tuning& cd360_tune() { return (tl_tune && g_nstreams.load(std::memory_order_relaxed) > 0) ? *tl_tune : g_tuning; }

Cd360TuneScope::Cd360TuneScope(void* stream) : prev_(tl_tune), set_(false) {
  if (g_nstreams.load(std::memory_order_relaxed) == 0) return;
  if (lookup(stream, &local_)) {
    tl_tune = &local_;
    set_ = true;
  } else if (tl_tune == &tl_query) {  // a query context of another stream must not leak into this launch
    tl_tune = nullptr;
    set_ = true;
  }
}
Cd360TuneScope::~Cd360TuneScope() {
  if (set_) tl_tune = prev_;
}

// The process-wide DEFAULT.  t == NULL restores the built-in defaults.  Streams with an override of their own are not affected.
extern "C" int cd360_set_tuning(const cd360_tuning* t) {
  if (!t) {
    g_tuning = make_default();
    return CD360_OK;
  }
  return sanitize(t, &g_tuning);
}

extern "C" int cd360_get_tuning(cd360_tuning* t) {
  if (!t) return CD360_ERR_ARG;
  *t = g_tuning;
  return CD360_OK;
}

// Per-stream override: every launch issued ON `stream` reads *t instead of the default (two samplers / two captures in one process can
// hold different tilings); t == NULL removes the override.  At most 16 streams.  Thread-safe.
extern "C" int cd360_set_stream_tuning(void* stream, const cd360_tuning* t) {
  cd360_tuning clean;
  if (t) {
    const int rc = sanitize(t, &clean);
    if (rc != CD360_OK) return rc;
    // conv_kgroup decides the K order weights are PACKED in (cd360_conv_k_order, read by pack_conv_weight on no stream in particular):
    // a stream that multiplied them in another order would be silently wrong, so the field cannot differ per stream
    if (clean.conv_kgroup > 0 && clean.conv_kgroup != g_tuning.conv_kgroup) return CD360_ERR_ARG;
    clean.conv_kgroup = g_tuning.conv_kgroup;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  int n = g_nstreams.load(std::memory_order_relaxed);
  for (int i = 0; i < n; ++i)
    if (g_streams[i].stream == stream) {
      if (t) {
        g_streams[i].t = clean;
      } else {
        g_streams[i] = g_streams[n - 1];
        g_nstreams.store(n - 1, std::memory_order_relaxed);
      }
      return CD360_OK;
    }
  if (!t) return CD360_OK;
  if (n == MAX_STREAMS) return CD360_ERR_SHAPE;
  g_streams[n].stream = stream;
  g_streams[n].t = clean;
  g_nstreams.store(n + 1, std::memory_order_relaxed);
  return CD360_OK;
}

// the tuning launches on `stream` read: its override, else the default
extern "C" int cd360_get_stream_tuning(void* stream, cd360_tuning* t) {
  if (!t) return CD360_ERR_ARG;
  if (g_nstreams.load(std::memory_order_relaxed) == 0 || !lookup(stream, t)) *t = g_tuning;
  return CD360_OK;
}

// The shape queries (cd360_gemm_tile_n, cd360_gemm_cstats_rows, cd360_conv_stats_slabs, cd360_conv_stats_rows, cd360_conv_dma_slab_rows,
// cd360_conv_k_order) size buffers for a launch that follows and take no stream: on the CALLING THREAD they answer for the stream named
// by the last cd360_query_stream (NULL, or a stream without an override: the default).  Thread-local; nothing shared is written.
extern "C" int cd360_query_stream(void* stream) {
  if (stream && g_nstreams.load(std::memory_order_relaxed) != 0 && lookup(stream, &tl_query)) tl_tune = &tl_query;
  else tl_tune = nullptr;
  return CD360_OK;
}

// 1 when the library was built with -DCD360_WHATIF (what-if timing bits and probe kernels compiled in), else 0
extern "C" int cd360_whatif_build(void) {
#ifdef CD360_WHATIF
  return 1;
#else
  return 0;
#endif
}
