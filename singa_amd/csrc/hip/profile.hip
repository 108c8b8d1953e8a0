// Time stamps taken by the device itself.  A hipGraph cannot contain timing
// events, so per-node profiling of a captured training step uses one-thread
// kernels that read the constant-rate wall clock: they are ordinary kernel
// nodes, run in stream order on whichever stream the profiled node uses, and
// accumulate on the device, so profiling a replayed graph needs no host
// synchronisation per iteration.
//
// Plays the role of the cudaEventRecord pairs of the reference's
// CudaGPU::TimeProfilingDoExec (src/core/device/cuda_gpu.cc:128-169).
#include "common.cuh"
#include "kernels.h"

namespace sa {

__global__ void stamp_open_kernel(unsigned long long *slots, int i) {
  slots[i] = wall_clock64();
}

__global__ void stamp_close_kernel(const unsigned long long *slots, int open,
                                   float *acc, int node, const int *gate,
                                   float sec_per_tick) {
  const unsigned long long t = wall_clock64();
  if (gate[0] != 0) acc[node] += (float)(t - slots[open]) * sec_per_tick;
}

}  // namespace sa

int sa_stamp_open(unsigned long long *slots, int i, hipStream_t s) {
  hipLaunchKernelGGL(sa::stamp_open_kernel, dim3(1), dim3(1), 0, s, slots, i);
  return 0;
}

int sa_stamp_close(const unsigned long long *slots, int open, float *acc,
                   int node, const int *gate, float sec_per_tick,
                   hipStream_t s) {
  hipLaunchKernelGGL(sa::stamp_close_kernel, dim3(1), dim3(1), 0, s, slots,
                     open, acc, node, gate, sec_per_tick);
  return 0;
}

double sa_wall_clock_hz() {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1e8;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) !=
          hipSuccess || khz <= 0)
    return 1e8;  // 100 MHz reference clock
  return 1e3 * (double)khz;
}
