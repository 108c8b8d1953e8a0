// Index kernels: gather slices along one axis (embedding lookup, ONNX Gather,
// reflect / edge padding, flips) and the matching scatter-add for gradients.
// The tensor is viewed as (outer, n, inner); one launch handles the whole
// index list, replacing the per-index SliceOn / ConcatOn copy chains the
// reference builds in Python (python/singa/autograd.py:4448-4526, 5665-5712).
#include "common.cuh"
#include "kernels.h"

namespace sa {

template <typename T>
__global__ void index_select_kernel(const T *__restrict__ x,
                                    const int *__restrict__ idx,
                                    T *__restrict__ out, int64_t outer,
                                    int64_t n_in, int64_t n_out,
                                    int64_t inner) {
  const int64_t total = outer * n_out * inner;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    const int64_t in = i % inner;
    const int64_t j = (i / inner) % n_out;
    const int64_t o = i / (inner * n_out);
    int64_t src = idx[j];
    if (src < 0) src += n_in;
    out[i] = x[(o * n_in + src) * inner + in];
  }
}

template <typename T>
__global__ void index_add_kernel(const T *__restrict__ dy,
                                 const int *__restrict__ idx,
                                 float *__restrict__ out, int64_t outer,
                                 int64_t n_in, int64_t n_out, int64_t inner) {
  const int64_t total = outer * n_out * inner;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    const int64_t in = i % inner;
    const int64_t j = (i / inner) % n_out;
    const int64_t o = i / (inner * n_out);
    int64_t dst = idx[j];
    if (dst < 0) dst += n_in;
    atomicAdd(out + (o * n_in + dst) * inner + in, to_f<T>(dy[i]));
  }
}

}  // namespace sa

using namespace sa;

int sa_index_select(int dtype, const void *x, const int *idx, void *out,
                    int64_t outer, int64_t n_in, int64_t n_out, int64_t inner,
                    hipStream_t s) {
  const int64_t total = outer * n_out * inner;
  if (total <= 0) return 0;
  SA_DISPATCH_ALL(dtype, T, {
    hipLaunchKernelGGL((index_select_kernel<T>), dim3(grid_for(total, 256)),
                       dim3(256), 0, s, (const T *)x, idx, (T *)out, outer,
                       n_in, n_out, inner);
  });
  return 0;
}

int sa_index_add(int dtype, const void *dy, const int *idx, void *out,
                 int64_t outer, int64_t n_in, int64_t n_out, int64_t inner,
                 hipStream_t s) {
  const int64_t total = outer * n_out * inner;
  if (dtype != kF32) return -3;
  zero_async(out, (size_t)(outer * n_in * inner) * sizeof(float), s);
  if (total <= 0) return 0;
  hipLaunchKernelGGL((index_add_kernel<float>), dim3(grid_for(total, 256)),
                     dim3(256), 0, s, (const float *)dy, idx, (float *)out,
                     outer, n_in, n_out, inner);
  return 0;
}
