// Weight prefetcher of the GEMM family (prefetch.hip).  Internal to libcd360_hip.so; the public entry points are in include/cd360_hip.h.
#pragma once
#include <hip/hip_runtime.h>
void cd360_prefetch_before_launch(hipStream_t stream, const void* w, long bytes);  // no-op unless `stream` (or the wildcard) is armed
void cd360_prefetch_after_launch(hipStream_t stream);           // no-op unless armed
