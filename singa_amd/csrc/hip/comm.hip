// Kernels around the collectives of data-parallel training.
//
// pack / unpack move a whole fusion bucket (dozens of gradient tensors) to and
// from one flat buffer in a single launch, converting on the fly between the
// parameter dtype and the wire dtype (fp32 <-> fp16 / bf16 compression) and,
// on the way back, applying the 1 / world_size scaling.  The reference issues
// one cudaMemcpy per tensor plus separate conversion kernels
// (src/io/communicator.cc:225-414).  The sparsification helpers implement
// threshold selection with stream compaction into fixed-capacity buffers and
// the scatter-add of gathered (index, value) pairs, all without host syncs.
#include "common.cuh"
#include "kernels.h"

namespace sa {

// grid.y = tensor index, grid.x strides over that tensor's elements.  The
// (pointer, offset, count) entries travel in the kernel argument block: no
// device-side table whose lifetime would have to span the side streams, no
// host-to-device copy, and a captured launch carries them with it.
template <typename TS, typename TD>
__global__ void pack_kernel(PackBatch batch, TD *__restrict__ bucket) {
  const int t = blockIdx.y;
  const TS *src = (const TS *)batch.ptr[t];
  TD *dst = bucket + batch.offset[t];
  const int64_t n = batch.count[t];
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    dst[i] = from_f<TD>(to_f<TS>(src[i]));
}

template <typename TS, typename TD>
__global__ void unpack_kernel(PackBatch batch, const TS *__restrict__ bucket,
                              float scale) {
  const int t = blockIdx.y;
  TD *dst = (TD *)batch.ptr[t];
  const TS *src = bucket + batch.offset[t];
  const int64_t n = batch.count[t];
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    dst[i] = from_f<TD>(to_f<TS>(src[i]) * scale);
}

// keeps |x| >= thr: appends (index + 1, value) pairs, count is the cursor
__global__ void sparsify_threshold_kernel(const float *__restrict__ x,
                                          float thr, float *__restrict__ vals,
                                          int *__restrict__ idx,
                                          int *__restrict__ count, int64_t n,
                                          int64_t cap) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step) {
    const float v = x[i];
    if (fabsf(v) >= thr) {
      const int slot = atomicAdd(count, 1);
      if (slot < cap) {
        vals[slot] = v;
        idx[slot] = (int)i;
      }
    }
  }
}

// ---- top-k by magnitude: most-significant-digit radix select -------------------
// The magnitude of a float, read as an unsigned integer, orders like the
// magnitude itself.  Four passes over the data, one per byte from the top,
// histogram the current byte of the elements that share the prefix found so
// far; a one-thread scan of the 256 bins picks the bin holding the k-th
// largest key and how many entries of that bin are still wanted.  After the
// last pass the prefix is the k-th largest key T: a final pass emits every
// element above T plus the first `ties` elements equal to T.  Everything stays
// on the device (k is known to the host, the threshold never is).
//
// work: 4 * 256 histogram bins, then {prefix, remaining, n_above, n_ties}.
constexpr int kTopkState = 4 * 256;

__device__ __forceinline__ unsigned mag_key(float v) {
  return __float_as_uint(v) & 0x7fffffffu;
}

__global__ void topk_hist_kernel(const float *__restrict__ x, int64_t n,
                                 int pass, unsigned *__restrict__ work) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned prefix = work[kTopkState];
  const int shift = 24 - 8 * pass;
  const unsigned hi_mask = pass == 0 ? 0u : 0xffffffffu << (shift + 8);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step) {
    const unsigned key = mag_key(x[i]);
    if ((key & hi_mask) == prefix) atomicAdd(&h[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&work[pass * 256 + threadIdx.x], h[threadIdx.x]);
}

__global__ void topk_scan_kernel(int pass, unsigned k,
                                 unsigned *__restrict__ work) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned *h = work + pass * 256;
  unsigned rem = pass == 0 ? k : work[kTopkState + 1];
  int d = 255;
  for (; d > 0; --d) {
    if (h[d] >= rem) break;
    rem -= h[d];
  }
  work[kTopkState] |= (unsigned)d << (24 - 8 * pass);
  work[kTopkState + 1] = rem;
}

__global__ void topk_select_kernel(const float *__restrict__ x, int64_t n,
                                   unsigned k, float *__restrict__ vals,
                                   int *__restrict__ idx,
                                   unsigned *__restrict__ work) {
  const unsigned T = work[kTopkState], ties = work[kTopkState + 1];
  const unsigned above = k - ties;  // slots [0, above) hold the keys > T
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step) {
    const float v = x[i];
    const unsigned key = mag_key(v);
    if (key > T) {
      const unsigned slot = atomicAdd(&work[kTopkState + 2], 1u);
      vals[slot] = v;
      idx[slot] = (int)i;
    } else if (key == T) {
      const unsigned t = atomicAdd(&work[kTopkState + 3], 1u);
      if (t < ties) {
        vals[above + t] = v;
        idx[above + t] = (int)i;
      }
    }
  }
}

__global__ void scatter_add_sparse_kernel(const float *__restrict__ vals,
                                          const int *__restrict__ idx,
                                          const int *__restrict__ count,
                                          int64_t max_count,
                                          float *__restrict__ dense,
                                          int64_t n) {
  int64_t c = count ? (int64_t)count[0] : max_count;
  if (c > max_count) c = max_count;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < c;
       i += step) {
    const int j = idx[i];
    if (j >= 0 && j < n) atomicAdd(dense + j, vals[i]);
  }
}

// Sum of `nseg` gathered sparse payloads into a dense vector.  Every payload
// is `1 + 2 * cap` 32-bit words: the number of selected entries (it may exceed
// cap when the selection overflowed its capacity; only the first cap were
// stored), cap indices, cap float values.  One launch covers all segments.
__global__ void scatter_add_segments_kernel(const int *__restrict__ payloads,
                                            int nseg, int64_t cap, float sign,
                                            float *__restrict__ dense,
                                            int64_t n) {
  const int64_t stride = 1 + 2 * cap;
  const int64_t total = (int64_t)nseg * cap;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    const int64_t seg = i / cap, slot = i - seg * cap;
    const int *p = payloads + seg * stride;
    int64_t c = p[0];
    if (c > cap) c = cap;
    if (slot >= c) continue;
    const int j = p[1 + slot];
    const float v = __int_as_float(p[1 + cap + slot]);
    if (j >= 0 && j < n) atomicAdd(dense + j, sign * v);
  }
}

__global__ void set_word_kernel(int *p, int v) { p[0] = v; }

// one launch per kPackMax tensors; offsets are the running sum of the counts
template <typename TS, typename TD, bool PACK>
int launch_packing(int nt, void *const *ptrs, const int64_t *counts,
                   void *bucket, float scale, hipStream_t s) {
  int64_t off = 0;
  for (int first = 0; first < nt; first += kPackMax) {
    PackBatch b;
    const int n = nt - first < kPackMax ? nt - first : kPackMax;
    int64_t most = 1;
    for (int i = 0; i < n; ++i) {
      b.ptr[i] = ptrs[first + i];
      b.offset[i] = off;
      b.count[i] = counts[first + i];
      off += counts[first + i];
      most = i64max(most, counts[first + i]);
    }
    const int gx = (int)i64min((most + 1023) / 1024, 64);
    if (PACK)
      hipLaunchKernelGGL((pack_kernel<TS, TD>), dim3(gx, n), dim3(256), 0, s,
                         b, (TD *)bucket);
    else
      hipLaunchKernelGGL((unpack_kernel<TS, TD>), dim3(gx, n), dim3(256), 0, s,
                         b, (const TS *)bucket, scale);
  }
  return 0;
}

template <typename TS, typename TD>
int launch_pack(int nt, void *const *ptrs, const int64_t *counts, void *bucket,
                float scale, hipStream_t s) {
  return launch_packing<TS, TD, true>(nt, ptrs, counts, bucket, scale, s);
}
template <typename TS, typename TD>
int launch_unpack(int nt, void *const *ptrs, const int64_t *counts,
                  void *bucket, float scale, hipStream_t s) {
  return launch_packing<TS, TD, false>(nt, ptrs, counts, bucket, scale, s);
}

}  // namespace sa

using namespace sa;

#define PAIR_DISPATCH(FN, ...)                                              \
  if (src_dtype == kF32 && dst_dtype == kF32) return FN<float, float>(__VA_ARGS__); \
  if (src_dtype == kF32 && dst_dtype == kF16) return FN<float, __half>(__VA_ARGS__); \
  if (src_dtype == kF16 && dst_dtype == kF32) return FN<__half, float>(__VA_ARGS__); \
  if (src_dtype == kF32 && dst_dtype == kBF16) return FN<float, __hip_bfloat16>(__VA_ARGS__); \
  if (src_dtype == kBF16 && dst_dtype == kF32) return FN<__hip_bfloat16, float>(__VA_ARGS__); \
  if (src_dtype == kF16 && dst_dtype == kF16) return FN<__half, __half>(__VA_ARGS__); \
  if (src_dtype == kBF16 && dst_dtype == kBF16) return FN<__hip_bfloat16, __hip_bfloat16>(__VA_ARGS__); \
  return -1;

int sa_pack(int src_dtype, int dst_dtype, int n_tensors, void *const *ptrs,
            const int64_t *counts, void *bucket, hipStream_t s) {
  if (n_tensors <= 0) return 0;
  PAIR_DISPATCH(launch_pack, n_tensors, ptrs, counts, bucket, 1.f, s)
}

int sa_unpack(int src_dtype, int dst_dtype, int n_tensors, void *const *ptrs,
              const int64_t *counts, const void *bucket, float scale,
              hipStream_t s) {
  if (n_tensors <= 0) return 0;
  PAIR_DISPATCH(launch_unpack, n_tensors, ptrs, counts,
                const_cast<void *>(bucket), scale, s)
}

int sa_sparsify_threshold(const float *x, float thr, float *vals, int *idx,
                          int *count, int64_t n, int64_t cap, hipStream_t s) {
  zero_async(count, sizeof(int), s);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sparsify_threshold_kernel, dim3(grid_for(n, 256, 4)),
                     dim3(256), 0, s, x, thr, vals, idx, count, n, cap);
  return 0;
}

int64_t sa_topk_abs_workspace() { return kTopkState + 4; }

int sa_topk_abs(const float *x, int64_t n, int64_t k, float *vals, int *idx,
                unsigned *work, hipStream_t s) {
  if (k <= 0) return 0;
  if (k > n || n >= ((int64_t)1 << 31)) return -2;
  zero_async(work, (kTopkState + 4) * sizeof(unsigned), s);
  const dim3 grid(grid_for(n, 256, 8)), block(256);
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(topk_hist_kernel, grid, block, 0, s, x, n, pass, work);
    hipLaunchKernelGGL(topk_scan_kernel, dim3(1), dim3(64), 0, s, pass,
                       (unsigned)k, work);
  }
  hipLaunchKernelGGL(topk_select_kernel, grid, block, 0, s, x, n, (unsigned)k,
                     vals, idx, work);
  return 0;
}

int sa_sparse_select(const float *x, int64_t n, float thr, int topk, int64_t k,
                     int *payload, int64_t cap, unsigned *work,
                     hipStream_t s) {
  if (cap <= 0 || n <= 0) return -2;
  int *idx = payload + 1;
  float *vals = reinterpret_cast<float *>(payload + 1 + cap);
  if (topk) {
    if (k > cap) return -2;
    const int rc = sa_topk_abs(x, n, k, vals, idx, work, s);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, s, payload,
                       (int)k);
    return 0;
  }
  return sa_sparsify_threshold(x, thr, vals, idx, payload, n, cap, s);
}

int sa_scatter_add_segments(const int *payloads, int nseg, int64_t cap,
                            float sign, float *dense, int64_t n,
                            hipStream_t s) {
  if (nseg <= 0 || cap <= 0) return 0;
  hipLaunchKernelGGL(scatter_add_segments_kernel,
                     dim3(grid_for((int64_t)nseg * cap, 256)), dim3(256), 0, s,
                     payloads, nseg, cap, sign, dense, n);
  return 0;
}

int sa_scatter_add_sparse(const float *vals, const int *idx, const int *count,
                          int64_t max_count, float *dense, int64_t n,
                          hipStream_t s) {
  if (max_count <= 0) return 0;
  hipLaunchKernelGGL(scatter_add_sparse_kernel,
                     dim3(grid_for(max_count, 256)), dim3(256), 0, s, vals,
                     idx, count, max_count, dense, n);
  return 0;
}
