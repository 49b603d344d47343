// One-thread kernel that records the shader clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime): two stamps around
// a stretch of work on the same stream give the average shader clock the work ran at.  Built on the GPU box by tools/probe/clock_under_load.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void stamp_kernel(uint64_t* out) {
  out[0] = clock64();
  out[1] = wall_clock64();
}
extern "C" void cd360_probe_stamp(void* out, void* stream) { hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint64_t*)out); }
