// Weight prefetcher for a captured step (round 4).  One denoise step streams 5.3 GB of weights through a 256 MB Infinity Cache, so every
// GEMM / convolution launch finds its weights in HBM (tools/probe/cold_weights.py: +3 ... 12 % per launch against warm weights).  The
// launches of a captured step are a fixed sequence: while the step is captured (and only then: cd360_prefetch_arm ... _disarm) every launch
// of the GEMM family also enqueues, on a side stream forked into the same capture, a small kernel that touches one dword per 128-byte line
// of THIS launch's weights and depends only on the completion of the launch `lag` positions earlier -- in the replayed graph the touches
// of launch i run beside launches i - lag + 1 ... i - 1, and launch i finds its weights in the Infinity Cache.  The main chain never waits
// for the side branch (one join at the end of the capture); nothing spins; no extra HBM traffic (every line is read from HBM once, by
// whoever touches it first).  (SURVEY.md section 8 has no counterpart: the reference leaves residency to the caches.)
#include "cd360_common.h"
#include "cd360_prefetch.h"

#include <atomic>
#include <mutex>

namespace {
constexpr int RING = 8, MAX_ARMS = 8;
// One arm per capturing stream (cd360_prefetch_arm_on): two captures on two streams -- two samplers in one process -- prefetch independently.
// main == nullptr is the wildcard arm of the legacy entry point (cd360_prefetch_arm): it serves launches on ANY stream without an arm
// of its own.  The table is consulted only while it is non-empty (one relaxed atomic load per launch otherwise).
struct Arm {
  bool used = false;
  hipStream_t main = nullptr, side = nullptr;
  int lag = 2, wgs = 32;
  long index = 0, min_bytes = 1 << 20;
  uint32_t* sink = nullptr;
  hipEvent_t ev[RING];
  bool ev_made = false;
};
Arm g_arms[MAX_ARMS];
std::atomic<int> g_narmed{0};
std::mutex g_mu;

__global__ __launch_bounds__(256) void weight_touch_kernel(const unsigned char* __restrict__ base, long bytes, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const long stride = (long)gridDim.x * 256 * 128;
  for (long off = ((long)blockIdx.x * 256 + threadIdx.x) * 128; off < bytes; off += stride) acc ^= *reinterpret_cast<const uint32_t*>(base + off);
  if (acc == 0x9e3779b9u) sink[0] = acc;  // keeps the loads alive (and is harmless should a weight pattern ever produce the value)
}

Arm* find(hipStream_t stream) {  // g_mu held
  Arm* wild = nullptr;
  for (Arm& a : g_arms) {
    if (!a.used) continue;
    if (a.main == stream) return &a;
    if (a.main == nullptr) wild = &a;
  }
  return wild;
}
}  // namespace

void cd360_prefetch_before_launch(hipStream_t stream, const void* w, long bytes) {
  if (g_narmed.load(std::memory_order_relaxed) == 0 || !w) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Arm* a = find(stream);
  if (!a || bytes < a->min_bytes) return;
  if (a->index >= a->lag) hipStreamWaitEvent(a->side, a->ev[(a->index - a->lag) % RING], 0);
  hipLaunchKernelGGL(weight_touch_kernel, dim3((unsigned)a->wgs), dim3(256), 0, a->side, (const unsigned char*)w, bytes, a->sink);
}
void cd360_prefetch_after_launch(hipStream_t stream) {
  if (g_narmed.load(std::memory_order_relaxed) == 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Arm* a = find(stream);
  if (!a) return;
  hipEventRecord(a->ev[a->index % RING], stream);
  ++a->index;
}

// Arm inside a stream capture of `main_stream`, after `side_stream` has been forked into it: from now on every launch of the GEMM family
// ON main_stream is followed by an event record on it and preceded by a touch kernel of its weights on the side stream (`wgs` workgroups of
// 256 threads; weights below `min_bytes` are left alone) that waits for the event of the launch `lag` (1 .. 7) positions earlier.  sink: 4
// bytes of device scratch.  cd360_prefetch_disarm_on(main_stream) before the capture ends (the caller then joins the side stream).  Launches
// on other streams are not touched; up to 8 streams can be armed at once.  main_stream == NULL arms the wildcard (any stream without an arm).
extern "C" int cd360_prefetch_arm_on(void* main_stream, void* side_stream, int lag, int wgs, int64_t min_bytes, void* sink) {
  if (!side_stream || !sink || lag < 1 || lag >= RING || wgs <= 0) return CD360_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  Arm* a = nullptr;
  for (Arm& c : g_arms)
    if (c.used && c.main == (hipStream_t)main_stream) a = &c;
  if (!a)
    for (Arm& c : g_arms)
      if (!c.used) {
        a = &c;
        break;
      }
  if (!a) return CD360_ERR_SHAPE;
  if (!a->ev_made) {
    for (int i = 0; i < RING; ++i)
      if (hipEventCreateWithFlags(&a->ev[i], hipEventDisableTiming) != hipSuccess) return CD360_ERR_LAUNCH;
    a->ev_made = true;
  }
  if (!a->used) g_narmed.fetch_add(1, std::memory_order_relaxed);
  a->used = true;
  a->main = (hipStream_t)main_stream;
  a->side = (hipStream_t)side_stream;
  a->lag = lag; a->wgs = wgs; a->min_bytes = min_bytes; a->sink = (uint32_t*)sink; a->index = 0;
  return CD360_OK;
}
extern "C" int cd360_prefetch_disarm_on(void* main_stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (Arm& c : g_arms)
    if (c.used && c.main == (hipStream_t)main_stream) {
      c.used = false;
      g_narmed.fetch_sub(1, std::memory_order_relaxed);
    }
  return CD360_OK;
}
// the round-4 entry points: the wildcard arm (one capture at a time, whatever stream it runs on)
extern "C" int cd360_prefetch_arm(void* side_stream, int lag, int wgs, int64_t min_bytes, void* sink) {
  return cd360_prefetch_arm_on(nullptr, side_stream, lag, wgs, min_bytes, sink);
}
extern "C" int cd360_prefetch_disarm(void) { return cd360_prefetch_disarm_on(nullptr); }
