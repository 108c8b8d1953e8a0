// Shared device helpers for the gfx950 kernels: dtype load/store through
// float, wave64 reductions, launch geometry.
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sa {

// dtype codes shared with the Python side (core/dtypes.py)
enum DType : int { kF32 = 0, kF16 = 1, kI32 = 2, kBF16 = 7 };

constexpr int kWave = 64;      // CDNA wavefront width
constexpr int kMaxBlocks = 2048;  // 256 CUs x 8 resident blocks

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float to_f<int>(int v) { return static_cast<float>(v); }

template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ int from_f<int>(float v) { return static_cast<int>(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}

// Sums each of 16 per-lane values over the 32 lanes of the lane's half of the
// wave (the layout of a 32 x 32 MFMA accumulator block: 32 columns across the
// lanes of a half, 16 rows in registers).  The halves exchange registers while
// they add, so the whole reduction costs 16 shuffles instead of 80: on return
// a[0] of lane l holds the total of value number (l >> 1) & 15.
__device__ __forceinline__ void half_wave_sums16(float (&a)[16], int lane) {
#pragma unroll
  for (int width = 8; width >= 1; width >>= 1) {
    const int mask = 2 * width;  // 16, 8, 4, 2
    const bool upper = (lane & mask) != 0;
#pragma unroll
    for (int k = 0; k < width; ++k) {
      const float send = upper ? a[k] : a[k + width];
      const float keep = upper ? a[k + width] : a[k];
      a[k] = keep + __shfl_xor(send, mask, 64);
    }
  }
  a[0] += __shfl_xor(a[0], 1, 64);
}

// Block-wide sum for blockDim.x <= 1024 (16 waves); result valid in all lanes.
__device__ __forceinline__ float block_sum(float v, float *smem /*>=17 floats*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = (lane < nw) ? smem[lane] : 0.f;
  r = wave_sum(r);
  return r;
}
__device__ __forceinline__ float block_max(float v, float *smem) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = (lane < nw) ? smem[lane] : -3.4e38f;
  r = wave_max(r);
  return r;
}

inline int64_t i64min(int64_t a, int64_t b) { return a < b ? a : b; }
inline int64_t i64max(int64_t a, int64_t b) { return a > b ? a : b; }

inline int grid_for(int64_t n, int block, int per_thread = 1) {
  int64_t b = (n + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return static_cast<int>(b);
}

// Zero-fill and device-to-device copy as ordinary kernels.  Inside a captured
// hipGraph a memset / memcpy node is not reliably ordered against the kernel
// nodes around it (a zero-fill was seen to land after the atomics it was meant
// to precede once the device was idle at launch); a kernel node is.
__global__ static void zero_words_kernel(uint32_t *__restrict__ p, int64_t words,
                                         uint8_t *__restrict__ tail,
                                         int tail_bytes) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0)
    for (int b = 0; b < tail_bytes; ++b) tail[b] = 0;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    uint4 *p4 = reinterpret_cast<uint4 *>(p);
    const int64_t n4 = words >> 2;
    for (int64_t j = i; j < n4; j += step) p4[j] = make_uint4(0, 0, 0, 0);
    for (int64_t j = (n4 << 2) + i; j < words; j += step) p[j] = 0;
  } else {
    for (; i < words; i += step) p[i] = 0;
  }
}

// The whole grid clears `words` 32-bit words at p as a side job of a kernel
// whose own work is unrelated; every workgroup takes a strided share.
__device__ __forceinline__ void grid_zero_words(uint32_t *__restrict__ p,
                                                int64_t words) {
  const int64_t blocks = (int64_t)gridDim.x * gridDim.y * gridDim.z;
  const int64_t block =
      blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z);
  const int threads = blockDim.x * blockDim.y * blockDim.z;
  const int t = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  const int64_t step = blocks * threads;
  const int64_t first = block * threads + t;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    uint4 *p4 = reinterpret_cast<uint4 *>(p);
    const int64_t n4 = words >> 2;
    for (int64_t j = first; j < n4; j += step) p4[j] = make_uint4(0, 0, 0, 0);
    for (int64_t j = (n4 << 2) + first; j < words; j += step) p[j] = 0;
  } else {
    for (int64_t j = first; j < words; j += step) p[j] = 0;
  }
}

__global__ static void copy_words_kernel(uint32_t *__restrict__ d,
                                         const uint32_t *__restrict__ s,
                                         int64_t words,
                                         uint8_t *__restrict__ dtail,
                                         const uint8_t *__restrict__ stail,
                                         int tail_bytes) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0)
    for (int b = 0; b < tail_bytes; ++b) dtail[b] = stail[b];
  if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) &
       15) == 0) {
    uint4 *d4 = reinterpret_cast<uint4 *>(d);
    const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
    const int64_t n4 = words >> 2;
    for (int64_t j = i; j < n4; j += step) d4[j] = s4[j];
    for (int64_t j = (n4 << 2) + i; j < words; j += step) d[j] = s[j];
  } else {
    for (; i < words; i += step) d[i] = s[i];
  }
}

// p must be 4-byte aligned (every tensor here is)
// Process-wide switch (sa_set_deterministic): reductions that are split over
// workgroups combine their partial sums in a fixed order instead of with
// float atomics, so that two runs on the same inputs agree bit for bit.
int &deterministic_flag();
inline bool deterministic() { return deterministic_flag() != 0; }

inline void zero_async(void *p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return;
  const int64_t words = (int64_t)(bytes >> 2);
  hipLaunchKernelGGL(zero_words_kernel, dim3(grid_for(words, 256, 16)),
                     dim3(256), 0, s, (uint32_t *)p, words,
                     (uint8_t *)p + (words << 2), (int)(bytes & 3));
}

inline void copy_async(void *dst, const void *src, size_t bytes,
                       hipStream_t s) {
  if (bytes == 0) return;
  const int64_t words = (int64_t)(bytes >> 2);
  hipLaunchKernelGGL(copy_words_kernel, dim3(grid_for(words, 256, 16)),
                     dim3(256), 0, s, (uint32_t *)dst, (const uint32_t *)src,
                     words, (uint8_t *)dst + (words << 2),
                     (const uint8_t *)src + (words << 2), (int)(bytes & 3));
}

}  // namespace sa

#define SA_DISPATCH_FLOAT(dtype, T, ...)                                   \
  switch (dtype) {                                                         \
    case sa::kF32: { using T = float; __VA_ARGS__; break; }                 \
    case sa::kF16: { using T = __half; __VA_ARGS__; break; }                \
    case sa::kBF16: { using T = __hip_bfloat16; __VA_ARGS__; break; }       \
    default: return -1;                                                    \
  }

#define SA_DISPATCH_ALL(dtype, T, ...)                                     \
  switch (dtype) {                                                         \
    case sa::kF32: { using T = float; __VA_ARGS__; break; }                 \
    case sa::kF16: { using T = __half; __VA_ARGS__; break; }                \
    case sa::kBF16: { using T = __hip_bfloat16; __VA_ARGS__; break; }       \
    case sa::kI32: { using T = int; __VA_ARGS__; break; }                   \
    default: return -1;                                                    \
  }
