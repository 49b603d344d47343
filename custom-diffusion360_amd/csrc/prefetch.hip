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

namespace {
constexpr int RING = 8;
bool g_armed = false;
hipStream_t g_side = nullptr;
int g_lag = 2, g_wgs = 32;
long g_index = 0, g_min_bytes = 1 << 20;
hipEvent_t g_ev[RING];
bool g_ev_made = false;

__global__ __launch_bounds__(256) void weight_touch_kernel(const unsigned char* __restrict__ base, long bytes, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const long stride = (long)gridDim.x * 256 * 128;
  for (long off = ((long)blockIdx.x * 256 + threadIdx.x) * 128; off < bytes; off += stride) acc ^= *reinterpret_cast<const uint32_t*>(base + off);
  if (acc == 0x9e3779b9u) sink[0] = acc;  // keeps the loads alive (and is harmless should a weight pattern ever produce the value)
}
uint32_t* g_sink = nullptr;
}  // namespace

void cd360_prefetch_before_launch(const void* w, long bytes) {
  if (!g_armed || !w || bytes < g_min_bytes) return;
  if (g_index >= g_lag) hipStreamWaitEvent(g_side, g_ev[(g_index - g_lag) % RING], 0);
  hipLaunchKernelGGL(weight_touch_kernel, dim3((unsigned)g_wgs), dim3(256), 0, g_side, (const unsigned char*)w, bytes, g_sink);
}
void cd360_prefetch_after_launch(hipStream_t stream) {
  if (!g_armed) return;
  hipEventRecord(g_ev[g_index % RING], stream);
  ++g_index;
}

// Arm inside a stream capture, after `side_stream` has been forked into it: from now on every launch of the GEMM family is followed by an
// event record on its stream and preceded by a touch kernel of its weights on the side stream (`wgs` workgroups of 256 threads; weights
// below `min_bytes` are left alone) that waits for the event of the launch `lag` (1 .. 7) positions earlier.  sink: 4 bytes of device
// scratch.  cd360_prefetch_disarm() before the capture ends (the caller then joins the side stream).  Process-wide: one capture at a time.
extern "C" int cd360_prefetch_arm(void* side_stream, int lag, int wgs, int64_t min_bytes, void* sink) {
  if (!side_stream || !sink || lag < 1 || lag >= RING || wgs <= 0) return CD360_ERR_ARG;
  if (!g_ev_made) {
    for (int i = 0; i < RING; ++i)
      if (hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming) != hipSuccess) return CD360_ERR_LAUNCH;
    g_ev_made = true;
  }
  g_side = (hipStream_t)side_stream;
  g_lag = lag; g_wgs = wgs; g_min_bytes = min_bytes; g_sink = (uint32_t*)sink; g_index = 0;
  g_armed = true;
  return CD360_OK;
}
extern "C" int cd360_prefetch_disarm(void) {
  g_armed = false;
  return CD360_OK;
}
