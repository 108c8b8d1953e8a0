// Batch normalisation over (N, H, W) per channel for NCHW tensors (and over N
// for (N, C) matrices, HW = 1).
//
// Training forward and backward are two passes each.  Pass 1 reduces the
// per-channel statistics: grid = (C, S) blocks, block (c, s) walks the images
// n = s, s + S, ... of channel c (HW contiguous values each, coalesced) and
// writes its partial sums into a small workspace; sums are taken on values
// shifted by the channel's first element, which keeps E[x^2] - E[x]^2 free of
// cancellation.  Pass 2 re-reads the tensor, every block first folds the S
// partials of its channel (a few hundred bytes, L2 resident) and then applies
// the normalisation; block (c, 0) also updates the running statistics and the
// saved mean / inverse standard deviation.  Both passes are HBM bound: the
// tensor is read twice and written once, the minimum without a grid barrier.
//
// Semantics follow the reference's cuDNN calls (src/model/operation/
// batchnorm.cc:290-406): saved tensors are the batch mean and 1/sqrt(var+eps),
// running = (1 - factor) * running + factor * batch with the unbiased batch
// variance.
#include "common.cuh"
#include "kernels.h"

namespace sa {

struct Div32 {
  uint32_t d, mul, shift;
  Div32() : d(1), mul(0), shift(0) {}
  explicit Div32(uint32_t div) : d(div) {
    shift = 0;
    while ((1u << shift) < div) ++shift;
    uint64_t one = 1;
    mul = (uint32_t)(((one << 32) * ((one << shift) - div)) / div + 1);
  }
  __device__ __forceinline__ void divmod(uint32_t n, uint32_t &q,
                                         uint32_t &r) const {
    q = (__umulhi(n, mul) + n) >> shift;
    r = n - q * d;
  }
};

static inline int bn_splits(int N, int C) {
  int s = 2048 / (C > 0 ? C : 1);
  if (s < 1) s = 1;
  if (s > N) s = N;
  return s;
}

// work layout: [C][S][2]
template <typename T>
__global__ void bn_stats_kernel(const T *__restrict__ x,
                                float *__restrict__ work, int N, int C, int HW,
                                int S) {
  __shared__ float smem[32];
  const int c = blockIdx.x, s = blockIdx.y;
  const float shift = to_f<T>(x[(size_t)c * HW]);
  float sum = 0.f, sq = 0.f;
  for (int n = s; n < N; n += S) {
    const T *p = x + ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
      const float v = to_f<T>(p[i]) - shift;
      sum += v;
      sq += v * v;
    }
  }
  sum = block_sum(sum, smem);
  sq = block_sum(sq, smem);
  if (threadIdx.x == 0) {
    work[((size_t)c * S + s) * 2 + 0] = sum;
    work[((size_t)c * S + s) * 2 + 1] = sq;
  }
}

template <typename T>
__global__ void bn_apply_train_kernel(
    const T *__restrict__ x, const float *__restrict__ scale,
    const float *__restrict__ bias, float *__restrict__ rmean,
    float *__restrict__ rvar, float factor, float eps, T *__restrict__ y,
    float *__restrict__ mean_out, float *__restrict__ invstd_out,
    const float *__restrict__ work, int N, int C, int HW, int S) {
  const int c = blockIdx.x, s = blockIdx.y;
  float sum = 0.f, sq = 0.f;
  for (int i = 0; i < S; ++i) {
    sum += work[((size_t)c * S + i) * 2 + 0];
    sq += work[((size_t)c * S + i) * 2 + 1];
  }
  const float cnt = (float)N * (float)HW;
  const float shift = to_f<T>(x[(size_t)c * HW]);
  const float dm = sum / cnt;
  const float mean = shift + dm;
  float var = sq / cnt - dm * dm;
  var = fmaxf(var, 0.f);
  const float invstd = rsqrtf(var + eps);
  if (s == 0 && threadIdx.x == 0) {
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    const float unbiased = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
    rmean[c] = (1.f - factor) * rmean[c] + factor * mean;
    rvar[c] = (1.f - factor) * rvar[c] + factor * unbiased;
  }
  const float g = scale[c] * invstd;
  const float b = bias[c] - mean * g;
  for (int n = s; n < N; n += S) {
    const T *p = x + ((size_t)n * C + c) * HW;
    T *q = y + ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x)
      q[i] = from_f<T>(fmaf(to_f<T>(p[i]), g, b));
  }
}

template <typename T>
__global__ void bn_infer_kernel(const T *__restrict__ x,
                                const float *__restrict__ scale,
                                const float *__restrict__ bias,
                                const float *__restrict__ rmean,
                                const float *__restrict__ rvar, float eps,
                                T *__restrict__ y, int64_t total, int C,
                                Div32 div_hw, Div32 div_c) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    uint32_t nc, hw, n, c;
    div_hw.divmod((uint32_t)i, nc, hw);
    div_c.divmod(nc, n, c);
    const float g = scale[c] * rsqrtf(rvar[c] + eps);
    y[i] = from_f<T>((to_f<T>(x[i]) - rmean[c]) * g + bias[c]);
  }
}

// pass 1 of backward: sum(dy), sum(dy * xhat)
template <typename T>
__global__ void bn_bwd_stats_kernel(const T *__restrict__ dy,
                                    const T *__restrict__ x,
                                    const float *__restrict__ mean,
                                    const float *__restrict__ invstd,
                                    float *__restrict__ work, int N, int C,
                                    int HW, int S) {
  __shared__ float smem[32];
  const int c = blockIdx.x, s = blockIdx.y;
  const float m = mean[c], iv = invstd[c];
  float sdy = 0.f, sdyx = 0.f;
  for (int n = s; n < N; n += S) {
    const size_t off = ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
      const float d = to_f<T>(dy[off + i]);
      sdy += d;
      sdyx += d * (to_f<T>(x[off + i]) - m) * iv;
    }
  }
  sdy = block_sum(sdy, smem);
  sdyx = block_sum(sdyx, smem);
  if (threadIdx.x == 0) {
    work[((size_t)c * S + s) * 2 + 0] = sdy;
    work[((size_t)c * S + s) * 2 + 1] = sdyx;
  }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(
    const T *__restrict__ dy, const T *__restrict__ x,
    const float *__restrict__ scale, const float *__restrict__ mean,
    const float *__restrict__ invstd, T *__restrict__ dx,
    float *__restrict__ dscale, float *__restrict__ dbias,
    const float *__restrict__ work, int N, int C, int HW, int S) {
  const int c = blockIdx.x, s = blockIdx.y;
  float sdy = 0.f, sdyx = 0.f;
  for (int i = 0; i < S; ++i) {
    sdy += work[((size_t)c * S + i) * 2 + 0];
    sdyx += work[((size_t)c * S + i) * 2 + 1];
  }
  if (s == 0 && threadIdx.x == 0) {
    dscale[c] = sdyx;
    dbias[c] = sdy;
  }
  const float cnt = (float)N * (float)HW;
  const float m = mean[c], iv = invstd[c];
  const float g = scale[c] * iv;
  const float k1 = sdy / cnt, k2 = sdyx / cnt;
  for (int n = s; n < N; n += S) {
    const size_t off = ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
      const float xhat = (to_f<T>(x[off + i]) - m) * iv;
      dx[off + i] = from_f<T>(g * (to_f<T>(dy[off + i]) - k1 - xhat * k2));
    }
  }
}

}  // namespace sa

using namespace sa;

int64_t sa_bn_workspace(int N, int C, int HW) {
  return (int64_t)C * bn_splits(N, C) * 2;
}

static inline int bn_block(int HW) { return HW >= 256 ? 256 : (HW > 64 ? 128 : 64); }

int sa_bn_fwd_train(int dtype, const void *x, const float *scale,
                    const float *bias, float *rmean, float *rvar, float factor,
                    float eps, void *y, float *mean, float *invstd, int N,
                    int C, int HW, float *work, hipStream_t s) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  const int S = bn_splits(N, C);
  const dim3 grid(C, S), block(bn_block(HW));
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((bn_stats_kernel<T>), grid, block, 0, s, (const T *)x,
                       work, N, C, HW, S);
    hipLaunchKernelGGL((bn_apply_train_kernel<T>), grid, block, 0, s,
                       (const T *)x, scale, bias, rmean, rvar, factor, eps,
                       (T *)y, mean, invstd, work, N, C, HW, S);
  });
  return 0;
}

int sa_bn_fwd_infer(int dtype, const void *x, const float *scale,
                    const float *bias, const float *rmean, const float *rvar,
                    float eps, void *y, int N, int C, int HW, hipStream_t s) {
  const int64_t total = (int64_t)N * C * HW;
  if (total <= 0) return 0;
  if (total >= ((int64_t)1 << 31)) return -2;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((bn_infer_kernel<T>), dim3(grid_for(total, 256, 4)),
                       dim3(256), 0, s, (const T *)x, scale, bias, rmean, rvar,
                       eps, (T *)y, total, C, Div32(HW), Div32(C));
  });
  return 0;
}

int sa_bn_bwd(int dtype, const void *dy, const void *x, const float *scale,
              const float *mean, const float *invstd, void *dx, float *dscale,
              float *dbias, int N, int C, int HW, float *work, hipStream_t s) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  const int S = bn_splits(N, C);
  const dim3 grid(C, S), block(bn_block(HW));
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T>), grid, block, 0, s,
                       (const T *)dy, (const T *)x, mean, invstd, work, N, C,
                       HW, S);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), grid, block, 0, s,
                       (const T *)dy, (const T *)x, scale, mean, invstd,
                       (T *)dx, dscale, dbias, work, N, C, HW, S);
  });
  return 0;
}
