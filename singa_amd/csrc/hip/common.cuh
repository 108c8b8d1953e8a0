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
