// Layer normalisation over the last dimension and the erf-form GELU, each as
// one pass over HBM per direction.
//
//   y = (x - mean(x)) * rsqrt(var(x) + eps) * gamma + beta        per row
//
// The reference has neither operator: its ONNX importer runs the exported
// decomposition (ReduceMean, Sub, Pow, ReduceMean, Add, Sqrt, Div, Mul, Add:
// nine kernels and eight intermediate tensors per layer norm, five for GELU;
// python/singa/sonnx.py).  Here the importer recognises both chains and the
// operators are also available directly (autograd.LayerNorm, autograd.Gelu).
//
// forward: one workgroup per row; the row is read once into registers when
// it has at most 8 * blockDim elements (two-pass statistics on the
// registers), otherwise it is re-read from L2.  mean and 1/std are kept in
// float for the backward pass.
// backward: dx per row with two row reductions; dgamma / dbeta are column
// sums over all rows, done by a second kernel that strides rows with a
// [8 row-lanes][32 columns] workgroup and reduces through LDS.
#include <initializer_list>

#include "common.cuh"
#include "kernels.h"

namespace sa {

constexpr int kLnCache = 8;  // row elements a thread can keep in registers

template <typename T>
__global__ void layernorm_fwd_kernel(const T *__restrict__ x,
                                     const T *__restrict__ gamma,
                                     const T *__restrict__ beta,
                                     T *__restrict__ y,
                                     float *__restrict__ mean_out,
                                     float *__restrict__ rstd_out, int64_t rows,
                                     int D, float eps) {
  __shared__ float smem[32];
  const int64_t r = blockIdx.x;
  const T *xr = x + r * D;
  T *yr = y + r * D;
  const bool cached = D <= kLnCache * (int)blockDim.x;
  float v[kLnCache];
  float sum = 0.f;
  if (cached) {
#pragma unroll
    for (int k = 0; k < kLnCache; ++k) {
      const int i = threadIdx.x + k * blockDim.x;
      v[k] = i < D ? to_f<T>(xr[i]) : 0.f;
      sum += v[k];
    }
  } else {
    for (int i = threadIdx.x; i < D; i += blockDim.x) sum += to_f<T>(xr[i]);
  }
  const float mean = block_sum(sum, smem) / (float)D;
  float sq = 0.f;
  if (cached) {
#pragma unroll
    for (int k = 0; k < kLnCache; ++k) {
      const int i = threadIdx.x + k * blockDim.x;
      const float d = i < D ? v[k] - mean : 0.f;
      sq += d * d;
    }
  } else {
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
      const float d = to_f<T>(xr[i]) - mean;
      sq += d * d;
    }
  }
  const float rstd = rsqrtf(block_sum(sq, smem) / (float)D + eps);
  if (threadIdx.x == 0) {
    mean_out[r] = mean;
    rstd_out[r] = rstd;
  }
  if (cached) {
#pragma unroll
    for (int k = 0; k < kLnCache; ++k) {
      const int i = threadIdx.x + k * blockDim.x;
      if (i < D) {
        const float g = gamma ? to_f<T>(gamma[i]) : 1.f;
        const float b = beta ? to_f<T>(beta[i]) : 0.f;
        yr[i] = from_f<T>((v[k] - mean) * rstd * g + b);
      }
    }
  } else {
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
      const float g = gamma ? to_f<T>(gamma[i]) : 1.f;
      const float b = beta ? to_f<T>(beta[i]) : 0.f;
      yr[i] = from_f<T>((to_f<T>(xr[i]) - mean) * rstd * g + b);
    }
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))
template <typename T>
__global__ void layernorm_bwd_dx_kernel(const T *__restrict__ dy,
                                        const T *__restrict__ x,
                                        const T *__restrict__ gamma,
                                        const float *__restrict__ mean,
                                        const float *__restrict__ rstd,
                                        T *__restrict__ dx, int64_t rows,
                                        int D) {
  __shared__ float smem[32];
  const int64_t r = blockIdx.x;
  const T *xr = x + r * D, *dr = dy + r * D;
  T *or_ = dx + r * D;
  const float m = mean[r], rs = rstd[r];
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const float g = gamma ? to_f<T>(gamma[i]) : 1.f;
    const float gd = g * to_f<T>(dr[i]);
    const float xh = (to_f<T>(xr[i]) - m) * rs;
    s1 += gd;
    s2 += gd * xh;
  }
  s1 = block_sum(s1, smem) / (float)D;
  s2 = block_sum(s2, smem) / (float)D;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const float g = gamma ? to_f<T>(gamma[i]) : 1.f;
    const float gd = g * to_f<T>(dr[i]);
    const float xh = (to_f<T>(xr[i]) - m) * rs;
    or_[i] = from_f<T>(rs * (gd - s1 - xh * s2));
  }
}

// dgamma[c] = sum_r dy[r,c] * xhat[r,c], dbeta[c] = sum_r dy[r,c]
// block = 32 columns x 8 row-lanes; grid.x tiles the columns, grid.y splits
// the rows (partials combined with float atomics on zeroed outputs).
template <typename T>
__global__ void layernorm_bwd_param_kernel(const T *__restrict__ dy,
                                           const T *__restrict__ x,
                                           const float *__restrict__ mean,
                                           const float *__restrict__ rstd,
                                           float *__restrict__ dgamma,
                                           float *__restrict__ dbeta,
                                           int64_t rows, int D) {
  __shared__ float sg[8][33], sb[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float ag = 0.f, ab = 0.f;
  if (c < D) {
    for (int64_t r = (int64_t)blockIdx.y * 8 + ry; r < rows;
         r += (int64_t)gridDim.y * 8) {
      const float d = to_f<T>(dy[r * D + c]);
      ag += d * (to_f<T>(x[r * D + c]) - mean[r]) * rstd[r];
      ab += d;
    }
  }
  sg[ry][cx] = ag;
  sb[ry][cx] = ab;
  __syncthreads();
  if (ry == 0 && c < D) {
    float tg = 0.f, tb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      tg += sg[k][cx];
      tb += sb[k][cx];
    }
    atomicAdd(dgamma + c, tg);
    atomicAdd(dbeta + c, tb);
  }
}

// The same sums in ONE launch with outputs of any float type: the row slices
// of a column block meet in `sums` (2 * D floats: gamma's, then beta's) through
// atomics, the slice that draws the last ticket of its column block swaps the
// totals out -- zeros in -- and writes them in the outputs' type.  `sums` and
// `tickets` hold zeros before and after the launch, so there is no fill pass
// before it and no conversion pass behind it.  The scratch is only touched
// with atomics, which are performed where the chip's L2 caches meet; a thread
// has the value its add returned, i.e. the add is done, before the workgroup
// draws its ticket.
template <typename T, typename O>
__global__ void layernorm_bwd_param_fin_kernel(
    const T *__restrict__ dy, const T *__restrict__ x,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    O *__restrict__ dgamma, O *__restrict__ dbeta, float *__restrict__ sums,
    unsigned *__restrict__ tickets, int64_t rows, int D) {
  __shared__ float sg[8][33], sb[8][33];
  __shared__ int is_last;
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float ag = 0.f, ab = 0.f;
  if (c < D) {
    for (int64_t r = (int64_t)blockIdx.y * 8 + ry; r < rows;
         r += (int64_t)gridDim.y * 8) {
      const float d = to_f<T>(dy[r * D + c]);
      ag += d * (to_f<T>(x[r * D + c]) - mean[r]) * rstd[r];
      ab += d;
    }
  }
  sg[ry][cx] = ag;
  sb[ry][cx] = ab;
  __syncthreads();
  const bool mine = ry == 0 && c < D;
  float tg = 0.f, tb = 0.f;
  if (mine) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      tg += sg[k][cx];
      tb += sb[k][cx];
    }
  }
  if (gridDim.y == 1) {
    if (mine) {
      dgamma[c] = from_f<O>(tg);
      dbeta[c] = from_f<O>(tb);
    }
    return;
  }
  if (mine) {
    const float b0 = atomicAdd(sums + c, tg);
    const float b1 = atomicAdd(sums + D + c, tb);
    asm volatile("" ::"v"(b0), "v"(b1) : "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0)
    is_last = atomicAdd(tickets + blockIdx.x, 1u) == gridDim.y - 1;
  __syncthreads();
  if (!is_last) return;
  if (mine) {
    dgamma[c] = from_f<O>(atomicExch(sums + c, 0.f));
    dbeta[c] = from_f<O>(atomicExch(sums + D + c, 0.f));
  }
  if (threadIdx.x == 0) atomicExch(tickets + blockIdx.x, 0u);
}

// ---- one wave per row ------------------------------------------------------
// Rows of up to 2048 elements (a multiple of 4): the 64 lanes of a wave keep the
// row in registers as NV groups of four consecutive elements (8- or 16-byte
// accesses), the two row statistics are wave reductions -- no LDS, no
// barrier, four rows per 256-thread workgroup.
template <typename T>
__device__ __forceinline__ void ld4(const T *p, float *f) {
  T e[4];
  if (sizeof(T) == 4) {
    const float4 w = *reinterpret_cast<const float4 *>(p);
    __builtin_memcpy(e, &w, 16);
  } else {
    const uint2 w = *reinterpret_cast<const uint2 *>(p);
    __builtin_memcpy(e, &w, 8);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) f[k] = to_f<T>(e[k]);
}
template <typename T>
__device__ __forceinline__ void st4(T *p, const float *f) {
  T e[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) e[k] = from_f<T>(f[k]);
  if (sizeof(T) == 4) {
    float4 w;
    __builtin_memcpy(&w, e, 16);
    *reinterpret_cast<float4 *>(p) = w;
  } else {
    uint2 w;
    __builtin_memcpy(&w, e, 8);
    *reinterpret_cast<uint2 *>(p) = w;
  }
}

template <typename T, int NV>
__global__ void __launch_bounds__(256)
layernorm_fwd_wave_kernel(const T *__restrict__ x, const T *__restrict__ gamma,
                          const T *__restrict__ beta, T *__restrict__ y,
                          float *__restrict__ mean_out,
                          float *__restrict__ rstd_out, int64_t rows, int D,
                          float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const T *xr = x + r * D;
  float v[NV][4];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = (lane + 64 * k) * 4;
    if (i < D) {
      ld4<T>(xr + i, v[k]);
    } else {
      v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.f;
    }
    sum += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if ((lane + 64 * k) * 4 < D) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[k][j] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
  if (lane == 0) {
    mean_out[r] = mean;
    rstd_out[r] = rstd;
  }
  T *yr = y + r * D;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = (lane + 64 * k) * 4;
    if (i < D) {
      float g[4] = {1.f, 1.f, 1.f, 1.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
      if (gamma) ld4<T>(gamma + i, g);
      if (beta) ld4<T>(beta + i, b);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[k][j] - mean) * rstd * g[j] + b[j];
      st4<T>(yr + i, o);
    }
  }
}

template <typename T, int NV>
__global__ void __launch_bounds__(256)
layernorm_bwd_dx_wave_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                             const T *__restrict__ gamma,
                             const float *__restrict__ mean,
                             const float *__restrict__ rstd,
                             T *__restrict__ dx, int64_t rows, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const T *xr = x + r * D, *dr = dy + r * D;
  const float m = mean[r], rs = rstd[r];
  float gd[NV][4], xh[NV][4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = (lane + 64 * k) * 4;
    if (i < D) {
      float g[4] = {1.f, 1.f, 1.f, 1.f}, d[4], xv[4];
      if (gamma) ld4<T>(gamma + i, g);
      ld4<T>(dr + i, d);
      ld4<T>(xr + i, xv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gd[k][j] = g[j] * d[j];
        xh[k][j] = (xv[j] - m) * rs;
        s1 += gd[k][j];
        s2 += gd[k][j] * xh[k][j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) gd[k][j] = xh[k][j] = 0.f;
    }
  }
  s1 = wave_sum(s1) / (float)D;
  s2 = wave_sum(s2) / (float)D;
  T *or_ = dx + r * D;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = (lane + 64 * k) * 4;
    if (i < D) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rs * (gd[k][j] - s1 - xh[k][j] * s2);
      st4<T>(or_ + i, o);
    }
  }
}

__device__ __forceinline__ float gelu_f(float v) {
  return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
}
__device__ __forceinline__ float gelu_grad_f(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
  return cdf + v * pdf;
}

template <typename T>
__global__ void gelu_fwd_kernel(const T *__restrict__ x, T *__restrict__ y,
                                int64_t n) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    y[i] = from_f<T>(gelu_f(to_f<T>(x[i])));
}

template <typename T>
__global__ void gelu_bwd_kernel(const T *__restrict__ dy,
                                const T *__restrict__ x, T *__restrict__ dx,
                                int64_t n) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    dx[i] = from_f<T>(to_f<T>(dy[i]) * gelu_grad_f(to_f<T>(x[i])));
}

// 16 bytes per thread and step
template <typename T, bool BWD>
__global__ void gelu_vec_kernel(const T *__restrict__ dy,
                                const T *__restrict__ x, T *__restrict__ out,
                                int64_t nvec) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += step) {
    const uint4 xw = reinterpret_cast<const uint4 *>(x)[i];
    T xe[VEC], de[VEC], oe[VEC];
    __builtin_memcpy(xe, &xw, 16);
    if (BWD) {
      const uint4 dw = reinterpret_cast<const uint4 *>(dy)[i];
      __builtin_memcpy(de, &dw, 16);
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float v = to_f<T>(xe[k]);
      oe[k] = from_f<T>(BWD ? to_f<T>(de[k]) * gelu_grad_f(v) : gelu_f(v));
    }
    uint4 ow;
    __builtin_memcpy(&ow, oe, 16);
    reinterpret_cast<uint4 *>(out)[i] = ow;
  }
}

}  // namespace sa

using namespace sa;

// rows a wave can hold: a multiple of 4 elements, at most 8 groups per lane,
// every operand 16-byte aligned (rows then start on 8 / 16-byte boundaries)
static inline bool ln_wave_ok(int D, std::initializer_list<const void *> ps) {
  if (D % 4 != 0 || D > 2048) return false;
  for (const void *p : ps)
    if (p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15) != 0) return false;
  return true;
}

#define SA_LN_NV(D, NV, ...)                         \
  do {                                               \
    const int groups_ = ((D) / 4 + 63) / 64;         \
    if (groups_ <= 1) { constexpr int NV = 1; __VA_ARGS__; }       \
    else if (groups_ == 2) { constexpr int NV = 2; __VA_ARGS__; }  \
    else if (groups_ == 3) { constexpr int NV = 3; __VA_ARGS__; }  \
    else if (groups_ == 4) { constexpr int NV = 4; __VA_ARGS__; }  \
    else if (groups_ <= 6) { constexpr int NV = 6; __VA_ARGS__; }  \
    else { constexpr int NV = 8; __VA_ARGS__; }      \
  } while (0)

static inline int ln_block(int D) {
  return D >= 2048 ? 512 : (D >= 512 ? 256 : (D > 64 ? 128 : 64));
}

int sa_layernorm_fwd(int dtype, const void *x, const void *gamma,
                     const void *beta, void *y, float *mean, float *rstd,
                     int64_t rows, int D, float eps, hipStream_t s) {
  if (rows <= 0 || D <= 0) return 0;
  if (rows >= ((int64_t)1 << 31)) return -2;
  if (ln_wave_ok(D, {x, gamma, beta, y})) {
    const dim3 grid((unsigned)((rows + 3) / 4));
    SA_DISPATCH_FLOAT(dtype, T, {
      SA_LN_NV(D, NV, hipLaunchKernelGGL(
          (layernorm_fwd_wave_kernel<T, NV>), grid, dim3(256), 0, s,
          (const T *)x, (const T *)gamma, (const T *)beta, (T *)y, mean, rstd,
          rows, D, eps));
    });
    return 0;
  }
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((layernorm_fwd_kernel<T>), dim3((unsigned)rows),
                       dim3(ln_block(D)), 0, s, (const T *)x, (const T *)gamma,
                       (const T *)beta, (T *)y, mean, rstd, rows, D, eps);
  });
  return 0;
}

// dgamma / dbeta: float buffers of D elements (zeroed here), or null
int sa_layernorm_bwd(int dtype, const void *dy, const void *x,
                     const void *gamma, const float *mean, const float *rstd,
                     void *dx, float *dgamma, float *dbeta, int64_t rows, int D,
                     hipStream_t s) {
  if (rows <= 0 || D <= 0) return 0;
  if (rows >= ((int64_t)1 << 31)) return -2;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return -5;
  const bool wave = ln_wave_ok(D, {dy, x, gamma, dx});
  SA_DISPATCH_FLOAT(dtype, T, {
    if (wave) {
      SA_LN_NV(D, NV, hipLaunchKernelGGL(
          (layernorm_bwd_dx_wave_kernel<T, NV>),
          dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const T *)dy,
          (const T *)x, (const T *)gamma, mean, rstd, (T *)dx, rows, D));
    } else {
      hipLaunchKernelGGL((layernorm_bwd_dx_kernel<T>), dim3((unsigned)rows),
                         dim3(ln_block(D)), 0, s, (const T *)dy, (const T *)x,
                         (const T *)gamma, mean, rstd, (T *)dx, rows, D);
    }
    if (dgamma) {
      zero_async(dgamma, (size_t)D * sizeof(float), s);
      zero_async(dbeta, (size_t)D * sizeof(float), s);
      const int gx = (D + 31) / 32;
      int gy = (int)i64min(i64max(1024 / gx, 1), (rows + 63) / 64);
      if (gy < 1) gy = 1;
      hipLaunchKernelGGL((layernorm_bwd_param_kernel<T>), dim3(gx, gy),
                         dim3(256), 0, s, (const T *)dy, (const T *)x, mean,
                         rstd, dgamma, dbeta, rows, D);
    }
  });
  return 0;
}

// Parameter gradients only, one launch.  dgamma / dbeta: D elements of the
// inputs' type (out_typed != 0) or float; sums (>= 2 * D floats) and tickets
// (>= ceil(D / 32) counters) hold zeros before and after.  -3: the scratch is
// too small (the caller uses sa_layernorm_bwd's float outputs).
int sa_layernorm_bwd_param(int dtype, const void *dy, const void *x,
                           const float *mean, const float *rstd, void *dgamma,
                           void *dbeta, int out_typed, float *sums,
                           int64_t sums_cap, unsigned *tickets,
                           int64_t tickets_cap, int64_t rows, int D,
                           hipStream_t s) {
  if (rows <= 0 || D <= 0) return 0;
  if (rows >= ((int64_t)1 << 31)) return -2;
  if (dgamma == nullptr || dbeta == nullptr) return -5;
  const int gx = (D + 31) / 32;
  int gy = (int)i64min(i64max(1024 / gx, 1), (rows + 63) / 64);
  if (gy < 1) gy = 1;
  if (gy > 1 && (sums == nullptr || tickets == nullptr ||
                 2 * (int64_t)D > sums_cap || gx > tickets_cap))
    return -3;
  SA_DISPATCH_FLOAT(dtype, T, {
    if (out_typed)
      hipLaunchKernelGGL((layernorm_bwd_param_fin_kernel<T, T>), dim3(gx, gy),
                         dim3(256), 0, s, (const T *)dy, (const T *)x, mean,
                         rstd, (T *)dgamma, (T *)dbeta, sums, tickets, rows,
                         D);
    else
      hipLaunchKernelGGL((layernorm_bwd_param_fin_kernel<T, float>),
                         dim3(gx, gy), dim3(256), 0, s, (const T *)dy,
                         (const T *)x, mean, rstd, (float *)dgamma,
                         (float *)dbeta, sums, tickets, rows, D);
  });
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

static inline bool gelu_vec_ok(int64_t n, size_t elem,
                               std::initializer_list<const void *> ps) {
  if (n % (int64_t)(16 / elem) != 0) return false;
  for (const void *p : ps)
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) return false;
  return true;
}

int sa_gelu_fwd(int dtype, const void *x, void *y, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    if (gelu_vec_ok(n, sizeof(T), {x, y})) {
      const int64_t nvec = n / (16 / (int64_t)sizeof(T));
      hipLaunchKernelGGL((gelu_vec_kernel<T, false>),
                         dim3(grid_for(nvec, 256, 2)), dim3(256), 0, s,
                         (const T *)nullptr, (const T *)x, (T *)y, nvec);
    } else {
      hipLaunchKernelGGL((gelu_fwd_kernel<T>), dim3(grid_for(n, 256, 4)),
                         dim3(256), 0, s, (const T *)x, (T *)y, n);
    }
  });
  return 0;
}

int sa_gelu_bwd(int dtype, const void *dy, const void *x, void *dx, int64_t n,
                hipStream_t s) {
  if (n <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    if (gelu_vec_ok(n, sizeof(T), {dy, x, dx})) {
      const int64_t nvec = n / (16 / (int64_t)sizeof(T));
      hipLaunchKernelGGL((gelu_vec_kernel<T, true>),
                         dim3(grid_for(nvec, 256, 2)), dim3(256), 0, s,
                         (const T *)dy, (const T *)x, (T *)dx, nvec);
    } else {
      hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3(grid_for(n, 256, 4)),
                         dim3(256), 0, s, (const T *)dy, (const T *)x, (T *)dx,
                         n);
    }
  });
  return 0;
}
