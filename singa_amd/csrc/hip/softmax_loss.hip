// Soft-max forward / backward over the rows of a matrix, cross-entropy,
// and the fused soft-max-cross-entropy pair used by training.
//
// One wave handles one row when cols <= 2048 (the row is kept in registers,
// max and sum come from wave shuffles, a single read and a single write per
// element); longer rows take one 256-thread block and three passes.  The
// fused loss kernel produces the probabilities and the batch-mean loss in one
// launch (wave-per-row, one float atomic per row into the loss), the fused
// backward writes (p - onehot) / batch directly.
//
// Replaces cudnnSoftmaxForward/Backward and the CrossEntropy kernels of the
// reference (tensor_math_cuda.h:1274-1425, math_kernel.cu:501-611), and the
// 4 + 2 kernel chain of autograd.SoftMaxCrossEntropy.
#include "common.cuh"
#include "kernels.h"

namespace sa {

constexpr int kRegCols = 32;  // per-lane register budget: 64 * 32 = 2048 cols

template <typename T>
__global__ void softmax_fwd_wave_kernel(const T *__restrict__ x,
                                        T *__restrict__ y, int64_t rows,
                                        int64_t cols) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const T *row = x + r * cols;
    float v[kRegCols];
    float m = -3.4e38f;
#pragma unroll
    for (int k = 0; k < kRegCols; ++k) {
      const int64_t c = lane + 64 * k;
      v[k] = c < cols ? to_f<T>(row[c]) : -3.4e38f;
      m = fmaxf(m, v[k]);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < kRegCols; ++k) {
      const int64_t c = lane + 64 * k;
      v[k] = c < cols ? __expf(v[k] - m) : 0.f;
      sum += v[k];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    T *orow = y + r * cols;
#pragma unroll
    for (int k = 0; k < kRegCols; ++k) {
      const int64_t c = lane + 64 * k;
      if (c < cols) orow[c] = from_f<T>(v[k] * inv);
    }
  }
}

template <typename T>
__global__ void softmax_fwd_block_kernel(const T *__restrict__ x,
                                         T *__restrict__ y, int64_t rows,
                                         int64_t cols) {
  __shared__ float smem[32];
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const T *row = x + r * cols;
    T *orow = y + r * cols;
    float m = -3.4e38f;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x)
      m = fmaxf(m, to_f<T>(row[c]));
    m = block_max(m, smem);
    float sum = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x)
      sum += __expf(to_f<T>(row[c]) - m);
    sum = block_sum(sum, smem);
    const float inv = 1.f / sum;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x)
      orow[c] = from_f<T>(__expf(to_f<T>(row[c]) - m) * inv);
  }
}

// dx = (dy - sum(dy * y)) * y
template <typename T>
__global__ void softmax_bwd_kernel(const T *__restrict__ dy,
                                   const T *__restrict__ y, T *__restrict__ dx,
                                   int64_t rows, int64_t cols) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const T *dyr = dy + r * cols;
    const T *yr = y + r * cols;
    float dot = 0.f;
    for (int64_t c = lane; c < cols; c += 64)
      dot += to_f<T>(dyr[c]) * to_f<T>(yr[c]);
    dot = wave_sum(dot);
    T *dxr = dx + r * cols;
    for (int64_t c = lane; c < cols; c += 64)
      dxr[c] = from_f<T>((to_f<T>(dyr[c]) - dot) * to_f<T>(yr[c]));
  }
}

template <typename T, typename TT>
__device__ __forceinline__ float ce_row(const T *p, const TT *t, bool t_is_int,
                                        int64_t r, int64_t cols, int lane) {
  const float tiny = 1.17549435e-38f;
  if (t_is_int) {
    float v = 0.f;
    if (lane == 0) {
      const int label = (int)to_f<TT>(t[r]);
      v = -__logf(fmaxf(to_f<T>(p[r * cols + label]), tiny));
    }
    return v;  // only lane 0 carries the value
  }
  const TT *tr = t + r * cols;
  float tsum = 0.f, acc = 0.f;
  for (int64_t c = lane; c < cols; c += 64) {
    const float tv = to_f<TT>(tr[c]);
    tsum += tv;
    acc += tv * __logf(fmaxf(to_f<T>(p[r * cols + c]), tiny));
  }
  tsum = wave_sum(tsum);
  acc = wave_sum(acc);
  return -acc / tsum;
}

template <typename T, typename TT>
__global__ void cross_entropy_kernel(const T *__restrict__ p,
                                     const TT *__restrict__ t, int t_is_int,
                                     T *__restrict__ loss, int64_t rows,
                                     int64_t cols) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float v = ce_row<T, TT>(p, t, t_is_int != 0, r, cols, lane);
    if (lane == 0) loss[r] = from_f<T>(v);
  }
}

// grad (in place on p, or into dx): (p - target) * scale
template <typename T, typename TT>
__global__ void softmax_ce_bwd_kernel(const T *p,  // may alias dx
                                      const TT *__restrict__ t, int t_is_int,
                                      T *dx, int64_t rows,
                                      int64_t cols, float scale) {
  const int64_t n = rows * cols;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step) {
    const int64_t r = i / cols, c = i - r * cols;
    float tv;
    if (t_is_int)
      tv = ((int)to_f<TT>(t[r]) == (int)c) ? 1.f : 0.f;
    else
      tv = to_f<TT>(t[i]);
    dx[i] = from_f<T>((to_f<T>(p[i]) - tv) * scale);
  }
}

// fused: probabilities + batch-mean loss.  loss must be zeroed beforehand.
template <typename T, typename TT>
__global__ void softmax_ce_fused_fwd_kernel(const T *__restrict__ x,
                                            const TT *__restrict__ t,
                                            int t_is_int, T *__restrict__ p,
                                            float *__restrict__ loss_acc,
                                            int64_t rows, int64_t cols) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float tiny = 1.17549435e-38f;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const T *row = x + r * cols;
    T *prow = p + r * cols;
    float m = -3.4e38f;
    for (int64_t c = lane; c < cols; c += 64) m = fmaxf(m, to_f<T>(row[c]));
    m = wave_max(m);
    float sum = 0.f;
    for (int64_t c = lane; c < cols; c += 64) sum += __expf(to_f<T>(row[c]) - m);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float tsum = 0.f, acc = 0.f;
    int label = -1;
    if (t_is_int) label = (int)to_f<TT>(t[r]);
    for (int64_t c = lane; c < cols; c += 64) {
      const float pv = __expf(to_f<T>(row[c]) - m) * inv;
      prow[c] = from_f<T>(pv);
      if (t_is_int) {
        if ((int)c == label) acc += __logf(fmaxf(pv, tiny));
      } else {
        const float tv = to_f<TT>(t[r * cols + c]);
        tsum += tv;
        acc += tv * __logf(fmaxf(pv, tiny));
      }
    }
    acc = wave_sum(acc);
    if (!t_is_int) {
      tsum = wave_sum(tsum);
      acc /= tsum;
    }
    if (lane == 0) atomicAdd(loss_acc, -acc / (float)rows);
  }
}

template <typename T>
__global__ void finish_loss_kernel(const float *acc, T *out) {
  *out = from_f<T>(*acc);
}

}  // namespace sa

using namespace sa;

static inline int wave_grid(int64_t rows) {
  int64_t b = (rows + 3) / 4;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

int sa_softmax_fwd(int dtype, const void *x, void *y, int64_t rows,
                   int64_t cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    if (cols <= 64 * kRegCols)
      hipLaunchKernelGGL((softmax_fwd_wave_kernel<T>), dim3(wave_grid(rows)),
                         dim3(256), 0, s, (const T *)x, (T *)y, rows, cols);
    else
      hipLaunchKernelGGL((softmax_fwd_block_kernel<T>),
                         dim3((int)i64min(rows, kMaxBlocks)), dim3(256),
                         0, s, (const T *)x, (T *)y, rows, cols);
  });
  return 0;
}

int sa_softmax_bwd(int dtype, const void *dy, const void *y, void *dx,
                   int64_t rows, int64_t cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((softmax_bwd_kernel<T>), dim3(wave_grid(rows)),
                       dim3(256), 0, s, (const T *)dy, (const T *)y, (T *)dx,
                       rows, cols);
  });
  return 0;
}

#define TARGET_SWITCH(TT, ...)                                              \
  if (t_is_int || t_dtype == kI32) {                                        \
    if (t_dtype == kI32) { using TT = int; __VA_ARGS__; }                   \
    else if (t_dtype == kF32) { using TT = float; __VA_ARGS__; }            \
    else return -1;                                                         \
  } else {                                                                  \
    if (t_dtype == kF32) { using TT = float; __VA_ARGS__; }                 \
    else if (t_dtype == kF16) { using TT = __half; __VA_ARGS__; }           \
    else if (t_dtype == kBF16) { using TT = __hip_bfloat16; __VA_ARGS__; }  \
    else return -1;                                                         \
  }

int sa_cross_entropy_fwd(int dtype, const void *p, const void *t, int t_is_int,
                         int t_dtype, void *loss, int64_t rows, int64_t cols,
                         hipStream_t s) {
  if (rows <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    TARGET_SWITCH(TT, {
      hipLaunchKernelGGL((cross_entropy_kernel<T, TT>), dim3(wave_grid(rows)),
                         dim3(256), 0, s, (const T *)p, (const TT *)t,
                         t_is_int, (T *)loss, rows, cols);
    });
  });
  return 0;
}

int sa_softmax_ce_bwd(int dtype, void *p, const void *t, int t_is_int,
                      int t_dtype, int64_t rows, int64_t cols, hipStream_t s) {
  if (rows <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    TARGET_SWITCH(TT, {
      hipLaunchKernelGGL((softmax_ce_bwd_kernel<T, TT>),
                         dim3(grid_for(rows * cols, 256)), dim3(256), 0, s,
                         (const T *)p, (const TT *)t, t_is_int, (T *)p, rows,
                         cols, 1.0f);
    });
  });
  return 0;
}

int sa_softmax_ce_fused_fwd(int dtype, const void *x, const void *t,
                            int t_is_int, int t_dtype, void *p, void *loss,
                            int64_t rows, int64_t cols, hipStream_t s) {
  if (rows <= 0) return 0;
  if (dtype != kF32) return -3;  // the loss word doubles as float accumulator
  zero_async(loss, sizeof(float), s);
  TARGET_SWITCH(TT, {
    // deterministic mode: one wave adds the rows' losses in row order
    hipLaunchKernelGGL((softmax_ce_fused_fwd_kernel<float, TT>),
                       dim3(deterministic() ? 1 : wave_grid(rows)),
                       dim3(deterministic() ? 64 : 256), 0, s,
                       (const float *)x, (const TT *)t, t_is_int, (float *)p,
                       (float *)loss, rows, cols);
  });
  return 0;
}

int sa_softmax_ce_fused_bwd(int dtype, const void *p, const void *t,
                            int t_is_int, int t_dtype, void *dx, int64_t rows,
                            int64_t cols, hipStream_t s) {
  if (rows <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    TARGET_SWITCH(TT, {
      hipLaunchKernelGGL((softmax_ce_bwd_kernel<T, TT>),
                         dim3(grid_for(rows * cols, 256)), dim3(256), 0, s,
                         (const T *)p, (const TT *)t, t_is_int, (T *)dx, rows,
                         cols, 1.0f / (float)rows);
    });
  });
  return 0;
}
