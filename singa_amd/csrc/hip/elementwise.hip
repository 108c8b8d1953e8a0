// Element-wise kernels: unary math, tensor (x) tensor with broadcasting,
// tensor (x) scalar, clamp, fill, cast / strided copy, axpy with a host or a
// device-resident scalar.
//
// One generic kernel walks up to 8 (collapsed) dimensions with per-operand
// strides, so transposed and broadcast (stride 0) views are consumed in place
// and no operand is materialised first; a second kernel covers the fully
// contiguous case with 16-byte accesses.  Memory bound by design: the grid is
// capped at 2048 blocks of 256 threads and grid-strides over the rest.
//
// Replaces the reference's src/core/tensor/math_kernel.cu element-wise
// kernels and the cudnnOpTensor / cudnnTransformTensor / cudnnAddTensor call
// sites of tensor_math_cuda.h:160-370.
#include "common.cuh"
#include "kernels.h"

namespace sa {

__device__ __forceinline__ float round_half_away(float x) {
  return copysignf(floorf(fabsf(x) + 0.5f), x);
}

__device__ __forceinline__ float apply_unary(int op, float x, float a, float b) {
  switch (op) {
    case U_COPY: return x;
    case U_ABS: return fabsf(x);
    case U_EXP: return __expf(x);
    case U_LOG: return __logf(x);
    case U_SQRT: return sqrtf(x);
    case U_SQUARE: return x * x;
    case U_SIGN: return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    case U_CEIL: return ceilf(x);
    case U_FLOOR: return floorf(x);
    case U_ROUND: return round_half_away(x);
    case U_ROUNDE: return rintf(x);
    case U_SIGMOID: return 1.f / (1.f + __expf(-x));
    case U_TANH: return tanhf(x);
    case U_RELU: return fmaxf(x, 0.f);
    case U_ERF: return erff(x);
    case U_COS: return cosf(x);
    case U_COSH: return coshf(x);
    case U_ACOS: return acosf(x);
    case U_ACOSH: return acoshf(x);
    case U_SIN: return sinf(x);
    case U_SINH: return sinhf(x);
    case U_ASIN: return asinf(x);
    case U_ASINH: return asinhf(x);
    case U_TAN: return tanf(x);
    case U_ATAN: return atanf(x);
    case U_ATANH: return atanhf(x);
    case U_SOFTPLUS: return log1pf(__expf(x));
    case U_SOFTSIGN: return x / (1.f + fabsf(x));
    case U_CLAMP: return fminf(fmaxf(x, a), b);
    case U_ADD_S: return x + a;
    case U_SUB_S: return x - a;
    case U_MUL_S: return x * a;
    case U_DIV_S: return x / a;
    case U_POW_S: return powf(x, a);
    case U_RDIV_S: return a / x;
    case U_LT_S: return x < a ? 1.f : 0.f;
    case U_LE_S: return x <= a ? 1.f : 0.f;
    case U_GT_S: return x > a ? 1.f : 0.f;
    case U_GE_S: return x >= a ? 1.f : 0.f;
    case U_EQ_S: return x == a ? 1.f : 0.f;
    case U_MAX_S: return fmaxf(x, a);
    case U_MIN_S: return fminf(x, a);
    default: return x;
  }
}

__device__ __forceinline__ float apply_binary(int op, float x, float y) {
  switch (op) {
    case B_ADD: return x + y;
    case B_SUB: return x - y;
    case B_MUL: return x * y;
    case B_DIV: return x / y;
    case B_POW: return powf(x, y);
    case B_LT: return x < y ? 1.f : 0.f;
    case B_LE: return x <= y ? 1.f : 0.f;
    case B_GT: return x > y ? 1.f : 0.f;
    case B_GE: return x >= y ? 1.f : 0.f;
    case B_EQ: return x == y ? 1.f : 0.f;
    case B_MAX: return fmaxf(x, y);
    case B_MIN: return fminf(x, y);
    case B_RELU_BWD: return y > 0.f ? x : 0.f;  // x = dy, y = forward input
    default: return x;
  }
}

// ---- contiguous fast paths ------------------------------------------------
template <typename TI, typename TO, int VEC>
__global__ void unary_contig_kernel(int op, const TI *__restrict__ x,
                                    TO *__restrict__ y, int64_t n, float a,
                                    float b) {
  const int64_t nvec = n / VEC;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  struct alignas(sizeof(TI) * VEC) VI { TI v[VEC]; };
  struct alignas(sizeof(TO) * VEC) VO { TO v[VEC]; };
  const VI *xv = reinterpret_cast<const VI *>(x);
  VO *yv = reinterpret_cast<VO *>(y);
  for (int64_t i = tid; i < nvec; i += step) {
    VI in = xv[i];
    VO out;
#pragma unroll
    for (int k = 0; k < VEC; ++k)
      out.v[k] = from_f<TO>(apply_unary(op, to_f<TI>(in.v[k]), a, b));
    yv[i] = out;
  }
  for (int64_t i = nvec * VEC + tid; i < n; i += step)
    y[i] = from_f<TO>(apply_unary(op, to_f<TI>(x[i]), a, b));
}

template <typename T, int VEC>
__global__ void binary_contig_kernel(int op, const T *__restrict__ x,
                                     const T *__restrict__ y,
                                     T *__restrict__ z, int64_t n) {
  const int64_t nvec = n / VEC;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  struct alignas(sizeof(T) * VEC) V { T v[VEC]; };
  const V *xv = reinterpret_cast<const V *>(x);
  const V *yv = reinterpret_cast<const V *>(y);
  V *zv = reinterpret_cast<V *>(z);
  for (int64_t i = tid; i < nvec; i += step) {
    V a = xv[i], b = yv[i], o;
#pragma unroll
    for (int k = 0; k < VEC; ++k)
      o.v[k] = from_f<T>(apply_binary(op, to_f<T>(a.v[k]), to_f<T>(b.v[k])));
    zv[i] = o;
  }
  for (int64_t i = nvec * VEC + tid; i < n; i += step)
    z[i] = from_f<T>(apply_binary(op, to_f<T>(x[i]), to_f<T>(y[i])));
}

// ---- generic strided / broadcast path --------------------------------------
__device__ __forceinline__ void unravel(int64_t lin, const Desc &d,
                                        int64_t &o0, int64_t &o1, int64_t &o2) {
  o0 = o1 = o2 = 0;
#pragma unroll 1
  for (int k = d.ndim - 1; k >= 0; --k) {
    const int64_t idx = lin % d.shape[k];
    lin /= d.shape[k];
    o0 += idx * d.s0[k];
    o1 += idx * d.s1[k];
    o2 += idx * d.s2[k];
  }
}

template <typename TI, typename TO>
__global__ void unary_strided_kernel(int op, const TI *__restrict__ x,
                                     TO *__restrict__ y, Desc d, int64_t n,
                                     float a, float b) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step) {
    int64_t ox, oy, unused;
    unravel(i, d, ox, unused, oy);
    y[oy] = from_f<TO>(apply_unary(op, to_f<TI>(x[ox]), a, b));
  }
}

template <typename T>
__global__ void binary_strided_kernel(int op, const T *__restrict__ x,
                                      const T *__restrict__ y,
                                      T *__restrict__ z, Desc d, int64_t n) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step) {
    int64_t ox, oy, oz;
    unravel(i, d, ox, oy, oz);
    z[oz] = from_f<T>(apply_binary(op, to_f<T>(x[ox]), to_f<T>(y[oy])));
  }
}

template <typename T>
__global__ void fill_kernel(T *__restrict__ y, int64_t n, float v) {
  const T tv = from_f<T>(v);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    y[i] = tv;
}

// y += alpha * x, alpha read from device memory when alpha_ptr != nullptr
template <typename T>
__global__ void axpy_kernel(const float *__restrict__ alpha_ptr, float alpha,
                            const T *__restrict__ x, T *__restrict__ y,
                            int64_t n) {
  const float a = alpha_ptr ? alpha_ptr[0] : alpha;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  constexpr int VEC = 16 / sizeof(T);
  struct alignas(16) V { T v[VEC]; };
  const int64_t nvec = n / VEC;
  const V *xv = reinterpret_cast<const V *>(x);
  V *yv = reinterpret_cast<V *>(y);
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = tid; i < nvec; i += step) {
    V xx = xv[i], yy = yv[i];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
      yy.v[k] = from_f<T>(fmaf(a, to_f<T>(xx.v[k]), to_f<T>(yy.v[k])));
    yv[i] = yy;
  }
  for (int64_t i = nvec * VEC + tid; i < n; i += step)
    y[i] = from_f<T>(fmaf(a, to_f<T>(x[i]), to_f<T>(y[i])));
}

template <typename TI, typename TO>
int launch_unary(int op, const void *x, void *y, const Desc *d, int64_t n,
                 float a, float b, hipStream_t s) {
  if (n <= 0) return 0;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) |
                         reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (d == nullptr) {
    constexpr int VEC = (sizeof(TI) >= sizeof(TO) ? 16 / sizeof(TI)
                                                   : 16 / sizeof(TO));
    if (aligned) {
      hipLaunchKernelGGL((unary_contig_kernel<TI, TO, VEC>),
                         dim3(grid_for(n, 256, VEC)), dim3(256), 0, s, op,
                         (const TI *)x, (TO *)y, n, a, b);
    } else {
      hipLaunchKernelGGL((unary_contig_kernel<TI, TO, 1>),
                         dim3(grid_for(n, 256)), dim3(256), 0, s, op,
                         (const TI *)x, (TO *)y, n, a, b);
    }
  } else {
    hipLaunchKernelGGL((unary_strided_kernel<TI, TO>), dim3(grid_for(n, 256)),
                       dim3(256), 0, s, op, (const TI *)x, (TO *)y, *d, n, a,
                       b);
  }
  return 0;
}

}  // namespace sa

using namespace sa;

#define CAST_CASE(DI, TI)                                                   \
  case DI:                                                                  \
    switch (dt_out) {                                                       \
      case kF32: return launch_unary<TI, float>(op, x, y, d, n, a, b, s);   \
      case kF16: return launch_unary<TI, __half>(op, x, y, d, n, a, b, s);  \
      case kBF16:                                                           \
        return launch_unary<TI, __hip_bfloat16>(op, x, y, d, n, a, b, s);   \
      case kI32: return launch_unary<TI, int>(op, x, y, d, n, a, b, s);     \
      default: return -1;                                                   \
    }

int sa_unary(int op, int dt_in, int dt_out, const void *x, void *y,
             const Desc *d, int64_t n, float a, float b, hipStream_t s) {
  switch (dt_in) {
    CAST_CASE(kF32, float)
    CAST_CASE(kF16, __half)
    CAST_CASE(kBF16, __hip_bfloat16)
    CAST_CASE(kI32, int)
    default: return -1;
  }
}

int sa_binary(int op, int dtype, const void *x, const void *y, void *z,
              const Desc *d, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  SA_DISPATCH_ALL(dtype, T, {
    if (d == nullptr) {
      const bool aligned = ((reinterpret_cast<uintptr_t>(x) |
                             reinterpret_cast<uintptr_t>(y) |
                             reinterpret_cast<uintptr_t>(z)) & 15) == 0;
      constexpr int VEC = 16 / sizeof(T);
      if (aligned)
        hipLaunchKernelGGL((binary_contig_kernel<T, VEC>),
                           dim3(grid_for(n, 256, VEC)), dim3(256), 0, s, op,
                           (const T *)x, (const T *)y, (T *)z, n);
      else
        hipLaunchKernelGGL((binary_contig_kernel<T, 1>),
                           dim3(grid_for(n, 256)), dim3(256), 0, s, op,
                           (const T *)x, (const T *)y, (T *)z, n);
    } else {
      hipLaunchKernelGGL((binary_strided_kernel<T>), dim3(grid_for(n, 256)),
                         dim3(256), 0, s, op, (const T *)x, (const T *)y,
                         (T *)z, *d, n);
    }
  });
  return 0;
}

int sa_fill(int dtype, void *y, int64_t n, float v, hipStream_t s) {
  if (n <= 0) return 0;
  SA_DISPATCH_ALL(dtype, T, {
    hipLaunchKernelGGL((fill_kernel<T>), dim3(grid_for(n, 256, 4)), dim3(256),
                       0, s, (T *)y, n, v);
  });
  return 0;
}

int sa_axpy(int dtype, const float *alpha_ptr, float alpha, const void *x,
            void *y, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) |
                         reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (!aligned) return -2;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((axpy_kernel<T>),
                       dim3(grid_for(n, 256, 16 / sizeof(T))), dim3(256), 0, s,
                       alpha_ptr, alpha, (const T *)x, (T *)y, n);
  });
  return 0;
}

int sa_copy_bytes(void *dst, const void *src, int64_t bytes, hipStream_t s) {
  if (bytes <= 0) return 0;
  if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 3)
    return -2;
  sa::copy_async(dst, src, (size_t)bytes, s);
  return 0;
}

int sa_zero_bytes(void *dst, int64_t bytes, hipStream_t s) {
  if (bytes <= 0) return 0;
  if (reinterpret_cast<uintptr_t>(dst) & 3) return -2;
  sa::zero_async(dst, (size_t)bytes, s);
  return 0;
}
