// Tensor math of the OpenCL device: float32 (and int32 where the C++ API
// needs it) element-wise, reduction, softmax / cross-entropy, GEMM and random
// kernels, written for 64-wide wavefronts (work-groups of 256 = 4 waves).
// The operator codes U_*, B_*, R_* are passed as -D options from the C++
// enums, so the two sides cannot drift apart.
//
// Role: reference src/core/tensor/tensor_math_opencl.cl + distribution.cl.

typedef struct {
  int ndim;
  long shape[8];
  long s0[8];
  long s1[8];
  long s2[8];
} Desc;

inline float unary_op(int op, float x, float a, float b) {
  switch (op) {
    case U_COPY: return x;
    case U_ABS: return fabs(x);
    case U_EXP: return exp(x);
    case U_LOG: return log(x);
    case U_SQRT: return sqrt(x);
    case U_SQUARE: return x * x;
    case U_SIGN: return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    case U_CEIL: return ceil(x);
    case U_FLOOR: return floor(x);
    case U_ROUND: return round(x);
    case U_ROUNDE: return rint(x);
    case U_SIGMOID: return 1.f / (1.f + exp(-x));
    case U_TANH: return tanh(x);
    case U_RELU: return x > 0.f ? x : 0.f;
    case U_ERF: return erf(x);
    case U_COS: return cos(x);
    case U_COSH: return cosh(x);
    case U_ACOS: return acos(x);
    case U_ACOSH: return acosh(x);
    case U_SIN: return sin(x);
    case U_SINH: return sinh(x);
    case U_ASIN: return asin(x);
    case U_ASINH: return asinh(x);
    case U_TAN: return tan(x);
    case U_ATAN: return atan(x);
    case U_ATANH: return atanh(x);
    case U_SOFTPLUS: return log(1.f + exp(x));
    case U_SOFTSIGN: return x / (1.f + fabs(x));
    case U_CLAMP: return fmin(fmax(x, a), b);
    case U_ADD_S: return x + a;
    case U_SUB_S: return x - a;
    case U_MUL_S: return x * a;
    case U_DIV_S: return x / a;
    case U_POW_S: return pow(x, a);
    case U_RDIV_S: return a / x;
    case U_LT_S: return x < a ? 1.f : 0.f;
    case U_LE_S: return x <= a ? 1.f : 0.f;
    case U_GT_S: return x > a ? 1.f : 0.f;
    case U_GE_S: return x >= a ? 1.f : 0.f;
    case U_EQ_S: return x == a ? 1.f : 0.f;
    case U_MAX_S: return fmax(x, a);
    case U_MIN_S: return fmin(x, a);
    default: return NAN;
  }
}

inline float binary_op(int op, float x, float y) {
  switch (op) {
    case B_ADD: return x + y;
    case B_SUB: return x - y;
    case B_MUL: return x * y;
    case B_DIV: return x / y;
    case B_POW: return pow(x, y);
    case B_LT: return x < y ? 1.f : 0.f;
    case B_LE: return x <= y ? 1.f : 0.f;
    case B_GT: return x > y ? 1.f : 0.f;
    case B_GE: return x >= y ? 1.f : 0.f;
    case B_EQ: return x == y ? 1.f : 0.f;
    case B_MAX: return fmax(x, y);
    case B_MIN: return fmin(x, y);
    case B_RELU_BWD: return y > 0.f ? x : 0.f;
    default: return NAN;
  }
}

inline float load_as_float(__global const void *p, long i, int is_int) {
  return is_int ? (float)((__global const int *)p)[i]
                : ((__global const float *)p)[i];
}

inline void store_from_float(__global void *p, long i, int is_int, float v) {
  if (is_int) ((__global int *)p)[i] = (int)v;
  else ((__global float *)p)[i] = v;
}

// offsets of linear index i under a collapsed descriptor
inline void desc_offsets(const Desc *d, long i, long *o0, long *o1, long *o2) {
  long a = 0, b = 0, c = 0;
  for (int k = d->ndim - 1; k >= 0; --k) {
    const long idx = i % d->shape[k];
    i /= d->shape[k];
    a += idx * d->s0[k];
    b += idx * d->s1[k];
    c += idx * d->s2[k];
  }
  *o0 = a; *o1 = b; *o2 = c;
}

__kernel void k_unary(int op, __global const void *x, int in_is_int,
                      __global void *y, int out_is_int, long n, float a,
                      float b) {
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step)
    store_from_float(y, i, out_is_int,
                     unary_op(op, load_as_float(x, i, in_is_int), a, b));
}

__kernel void k_unary_desc(int op, __global const void *x, int in_is_int,
                           __global void *y, int out_is_int, Desc d, long n,
                           float a, float b) {
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step) {
    long o0, o1, o2;
    desc_offsets(&d, i, &o0, &o1, &o2);
    store_from_float(y, o2, out_is_int,
                     unary_op(op, load_as_float(x, o0, in_is_int), a, b));
  }
}

__kernel void k_binary(int op, __global const void *x, __global const void *y,
                       __global void *z, int is_int, long n) {
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step)
    store_from_float(z, i, is_int,
                     binary_op(op, load_as_float(x, i, is_int),
                               load_as_float(y, i, is_int)));
}

__kernel void k_binary_desc(int op, __global const void *x,
                            __global const void *y, __global void *z,
                            int is_int, Desc d, long n) {
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step) {
    long o0, o1, o2;
    desc_offsets(&d, i, &o0, &o1, &o2);
    store_from_float(z, o2, is_int,
                     binary_op(op, load_as_float(x, o0, is_int),
                               load_as_float(y, o1, is_int)));
  }
}

__kernel void k_fill(__global void *y, int is_int, long n, float v) {
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step)
    store_from_float(y, i, is_int, v);
}

__kernel void k_axpy(float alpha, __global const float *x, __global float *y,
                     long n) {
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step) y[i] += alpha * x[i];
}

// ---- reductions -----------------------------------------------------------------
inline float reduce_identity(int op) {
  return op == R_MAX ? -INFINITY : (op == R_MIN ? INFINITY : 0.f);
}

inline float reduce_term(int op, float x) {
  return op == R_ASUM ? fabs(x) : (op == R_SUMSQ ? x * x : x);
}

inline float reduce_join(int op, float a, float b) {
  return op == R_MAX ? fmax(a, b) : (op == R_MIN ? fmin(a, b) : a + b);
}

// the 256 values of a work-group -> one, through local memory
inline float group_reduce(int op, float v, __local float *buf) {
  const int t = get_local_id(0);
  buf[t] = v;
  barrier(CLK_LOCAL_MEM_FENCE);
  for (int s = get_local_size(0) / 2; s > 0; s >>= 1) {
    if (t < s) buf[t] = reduce_join(op, buf[t], buf[t + s]);
    barrier(CLK_LOCAL_MEM_FENCE);
  }
  const float r = buf[0];
  barrier(CLK_LOCAL_MEM_FENCE);
  return r;
}

// out[r] = op over the columns of row r; one work-group per row
__kernel void k_reduce_rows(int op, __global const float *x,
                            __global float *out, long rows, long cols) {
  __local float buf[256];
  for (long r = get_group_id(0); r < rows; r += get_num_groups(0)) {
    float acc = reduce_identity(op);
    for (long c = get_local_id(0); c < cols; c += get_local_size(0))
      acc = reduce_join(op, acc, reduce_term(op, x[r * cols + c]));
    acc = group_reduce(op, acc, buf);
    if (get_local_id(0) == 0) out[r] = op == R_MEAN ? acc / (float)cols : acc;
  }
}

// out[c] = sum over rows; one work-item per column (neighbouring items read
// neighbouring addresses)
__kernel void k_reduce_cols(__global const float *x, __global float *out,
                            long rows, long cols) {
  const long step = get_global_size(0);
  for (long c = get_global_id(0); c < cols; c += step) {
    float acc = 0.f;
    for (long r = 0; r < rows; ++r) acc += x[r * cols + c];
    out[c] = acc;
  }
}

// partial[g] = op over a grid-strided share of x; k_reduce_all_final joins
__kernel void k_reduce_all(int op, __global const float *x,
                           __global float *partial, long n) {
  __local float buf[256];
  float acc = reduce_identity(op);
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step)
    acc = reduce_join(op, acc, reduce_term(op, x[i]));
  acc = group_reduce(op, acc, buf);
  if (get_local_id(0) == 0) partial[get_group_id(0)] = acc;
}

__kernel void k_reduce_all_final(int op, __global const float *partial,
                                 int count, __global float *out) {
  __local float buf[256];
  float acc = reduce_identity(op);
  for (int i = get_local_id(0); i < count; i += get_local_size(0))
    acc = reduce_join(op, acc, partial[i]);
  acc = group_reduce(op, acc, buf);
  if (get_local_id(0) == 0) out[0] = acc;
}

// ---- softmax / cross-entropy ---------------------------------------------------
__kernel void k_softmax_fwd(__global const float *x, __global float *y,
                            long rows, long cols) {
  __local float buf[256];
  for (long r = get_group_id(0); r < rows; r += get_num_groups(0)) {
    __global const float *xr = x + r * cols;
    __global float *yr = y + r * cols;
    float m = -INFINITY;
    for (long c = get_local_id(0); c < cols; c += get_local_size(0))
      m = fmax(m, xr[c]);
    m = group_reduce(R_MAX, m, buf);
    float s = 0.f;
    for (long c = get_local_id(0); c < cols; c += get_local_size(0)) {
      const float e = exp(xr[c] - m);
      yr[c] = e;
      s += e;
    }
    s = group_reduce(R_SUM, s, buf);
    const float inv = 1.f / s;
    for (long c = get_local_id(0); c < cols; c += get_local_size(0))
      yr[c] *= inv;
  }
}

// dx = y * (dy - sum(dy * y)) per row
__kernel void k_softmax_bwd(__global const float *dy, __global const float *y,
                            __global float *dx, long rows, long cols) {
  __local float buf[256];
  for (long r = get_group_id(0); r < rows; r += get_num_groups(0)) {
    float s = 0.f;
    for (long c = get_local_id(0); c < cols; c += get_local_size(0))
      s += dy[r * cols + c] * y[r * cols + c];
    s = group_reduce(R_SUM, s, buf);
    for (long c = get_local_id(0); c < cols; c += get_local_size(0))
      dx[r * cols + c] = y[r * cols + c] * (dy[r * cols + c] - s);
  }
}

// loss[r] = -log(p[r, t[r]]) for integer labels
__kernel void k_cross_entropy_fwd(__global const float *p,
                                  __global const int *t, __global float *loss,
                                  long rows, long cols) {
  const long step = get_global_size(0);
  for (long r = get_global_id(0); r < rows; r += step) {
    const int label = t[r];
    const float v = (label >= 0 && label < cols) ? p[r * cols + label] : 1.f;
    loss[r] = -log(fmax(v, FLT_MIN));
  }
}

// p[r, t[r]] -= 1 : gradient of cross-entropy(softmax) w.r.t. the logits
__kernel void k_softmax_ce_bwd(__global float *p, __global const int *t,
                               long rows, long cols) {
  const long step = get_global_size(0);
  for (long r = get_global_id(0); r < rows; r += step) {
    const int label = t[r];
    if (label >= 0 && label < cols) p[r * cols + label] -= 1.f;
  }
}

// ---- GEMM ------------------------------------------------------------------------
// C[M x N] = alpha * A * B + beta * C with arbitrary row / column strides of
// A and B (transposes are strides).  A work-group of 16 x 16 items computes a
// 64 x 64 tile, each item a 4 x 4 block; the K dimension is walked in steps
// of 16 through local memory.
#define GT 64
#define GK 16
__kernel __attribute__((reqd_work_group_size(16, 16, 1)))
void k_gemm(long M, long N, long K, float alpha, __global const float *A,
            long a_rs, long a_cs, __global const float *B, long b_rs,
            long b_cs, float beta, __global float *C, long ldc) {
  __local float As[GK][GT + 1];
  __local float Bs[GK][GT + 1];
  const int tx = get_local_id(0), ty = get_local_id(1);
  const int t = ty * 16 + tx;
  const long m0 = (long)get_group_id(1) * GT, n0 = (long)get_group_id(0) * GT;
  float acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (long k0 = 0; k0 < K; k0 += GK) {
    // 64 x 16 elements of each operand, 4 per item
    for (int e = t; e < GT * GK; e += 256) {
      const int kk = e / GT, mm = e % GT;
      const long m = m0 + mm, n = n0 + mm, k = k0 + kk;
      As[kk][mm] = (m < M && k < K) ? A[m * a_rs + k * a_cs] : 0.f;
      Bs[kk][mm] = (n < N && k < K) ? B[k * b_rs + n * b_cs] : 0.f;
    }
    barrier(CLK_LOCAL_MEM_FENCE);
    for (int kk = 0; kk < GK; ++kk) {
      float a[4], b[4];
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
    barrier(CLK_LOCAL_MEM_FENCE);
  }
  for (int i = 0; i < 4; ++i) {
    const long m = m0 + ty * 4 + i;
    if (m >= M) continue;
    for (int j = 0; j < 4; ++j) {
      const long n = n0 + tx * 4 + j;
      if (n >= N) continue;
      const float prev = beta != 0.f ? beta * C[m * ldc + n] : 0.f;
      C[m * ldc + n] = alpha * acc[i][j] + prev;
    }
  }
}

// ---- random numbers --------------------------------------------------------------
// Counter-based: element i of a call draws from hash(seed, offset + i), so a
// result does not depend on the launch geometry.
inline uint mix32(ulong z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9UL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebUL;
  z ^= z >> 31;
  return (uint)(z >> 16);
}

inline float unit_open(ulong seed, ulong counter) {
  // (0, 1): 24 random bits, centred
  const uint r = mix32(seed * 0x9e3779b97f4a7c15UL + counter);
  return ((float)(r >> 8) + 0.5f) * (1.f / 16777216.f);
}

__kernel void k_rng_uniform(__global float *y, long n, float lo, float hi,
                            ulong seed, ulong offset) {
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step)
    y[i] = lo + (hi - lo) * unit_open(seed, offset + (ulong)i);
}

__kernel void k_rng_normal(__global float *y, long n, float mean, float std,
                           ulong seed, ulong offset) {
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step) {
    const float u1 = unit_open(seed, 2 * (offset + (ulong)i));
    const float u2 = unit_open(seed, 2 * (offset + (ulong)i) + 1);
    y[i] = mean + std * sqrt(-2.f * log(u1)) * cospi(2.f * u2);
  }
}

__kernel void k_rng_bernoulli(__global float *y, long n, float p, ulong seed,
                              ulong offset) {
  const long step = get_global_size(0);
  for (long i = get_global_id(0); i < n; i += step)
    y[i] = unit_open(seed, offset + (ulong)i) < p ? 1.f : 0.f;
}
