// Reductions: per-row (sum / mean / max / min over the fastest dimension),
// per-column sum (over the slowest dimension) and whole-tensor reductions.
//
// Row reductions give one wave to short rows and one 256-thread block to long
// ones; column sums tile the matrix 64 columns wide so every wave reads whole
// 256-byte lines, accumulate in registers down a strip of rows and finish
// with one float atomic per (strip, column).  Whole-tensor reductions do a
// grid-stride pass, a block reduction and one atomic per block.
//
// Replaces cudnnReduceTensor / cublasSasum / cublasSnrm2 / the GEMV-with-ones
// reductions and KernelRowMax of the reference (tensor_math_cuda.h:1000-1109,
// 1492-1563).
#include "common.cuh"
#include "kernels.h"

namespace sa {

template <int OP> __device__ __forceinline__ float r_init() {
  return OP == R_MAX ? -3.4e38f : (OP == R_MIN ? 3.4e38f : 0.f);
}
template <int OP> __device__ __forceinline__ float r_map(float x) {
  if (OP == R_ASUM) return fabsf(x);
  if (OP == R_SUMSQ) return x * x;
  return x;
}
template <int OP> __device__ __forceinline__ float r_comb(float a, float b) {
  if (OP == R_MAX) return fmaxf(a, b);
  if (OP == R_MIN) return fminf(a, b);
  return a + b;
}
template <int OP> __device__ __forceinline__ float r_wave(float v) {
  if (OP == R_MAX) return wave_max(v);
  if (OP == R_MIN) return wave_min(v);
  return wave_sum(v);
}

// one wave per row
template <typename T, int OP>
__global__ void reduce_rows_wave_kernel(const T *__restrict__ x,
                                        T *__restrict__ out, int64_t rows,
                                        int64_t cols) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const T *row = x + r * cols;
    float acc = r_init<OP>();
    for (int64_t c = lane; c < cols; c += 64)
      acc = r_comb<OP>(acc, r_map<OP>(to_f<T>(row[c])));
    acc = r_wave<OP>(acc);
    if (lane == 0) {
      if (OP == R_MEAN) acc /= (float)cols;
      out[r] = from_f<T>(acc);
    }
  }
}

// one block per row (long rows)
template <typename T, int OP>
__global__ void reduce_rows_block_kernel(const T *__restrict__ x,
                                         T *__restrict__ out, int64_t rows,
                                         int64_t cols) {
  __shared__ float smem[32];
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const T *row = x + r * cols;
    float acc = r_init<OP>();
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x)
      acc = r_comb<OP>(acc, r_map<OP>(to_f<T>(row[c])));
    acc = r_wave<OP>(acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = acc;
    __syncthreads();
    if (wid == 0) {
      float v = lane < (blockDim.x >> 6) ? smem[lane] : r_init<OP>();
      v = r_wave<OP>(v);
      if (lane == 0) {
        if (OP == R_MEAN) v /= (float)cols;
        out[r] = from_f<T>(v);
      }
    }
  }
}

// Where a workgroup's column sums go.  One row slice: straight to the typed
// output.  Several: atomically into `sums` (float, all zero on entry); the
// slice that arrives last at a column block (ticket counter) rounds the
// totals into the output and puts zeros back, so the scratch is clean for the
// next launch on the stream without a fill pass, and a 16-bit result needs no
// conversion pass.
struct ColsOut {
  void *out;            // cols elements of type float (typed == 0) or T
  float *sums;          // scratch, >= cols floats
  unsigned *tickets;    // scratch, >= gridDim.x counters
  int typed;
};

template <typename T>
__device__ __forceinline__ void cols_store(const ColsOut &o, int64_t col,
                                           float v) {
  if (o.typed) static_cast<T *>(o.out)[col] = from_f<T>(v);
  else static_cast<float *>(o.out)[col] = v;
}

// called by every thread of the workgroup; `mine` threads hold the sum `v` of
// column `col`.  The scratch is only ever touched with device-scope atomics,
// which execute at the point where the chip's eight L2 caches meet, so no
// device-wide fence (a write-back of the whole L2 here) is needed: a thread
// waits for the value its add returns, i.e. for the add to have been
// performed, before the workgroup takes its ticket; the holder of the last
// ticket therefore swaps complete totals out (and zeros in).
template <typename T>
__device__ __forceinline__ void cols_finish(const ColsOut &o, bool mine,
                                            int64_t col, float v) {
  if (gridDim.y == 1) {
    if (mine) cols_store<T>(o, col, v);
    return;
  }
  __shared__ int is_last;
  if (mine) {
    const float before = atomicAdd(o.sums + col, v);
    asm volatile("" ::"v"(before) : "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(o.tickets + blockIdx.x, 1u);
    is_last = t == gridDim.y - 1;
  }
  __syncthreads();
  if (!is_last) return;
  if (mine) cols_store<T>(o, col, atomicExch(o.sums + col, 0.f));
  if (threadIdx.x == 0) atomicExch(o.tickets + blockIdx.x, 0u);
}

// column sums: block = 256 threads = 4 row-lanes x 64 columns
template <typename T>
__global__ void reduce_cols_kernel(const T *__restrict__ x,
                                   float *__restrict__ out, int64_t rows,
                                   int64_t cols, int64_t rows_per_block) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float acc = 0.f;
  if (c < cols)
    for (int64_t r = r0 + ry; r < r1; r += 4) acc += to_f<T>(x[r * cols + c]);
  part[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && c < cols) {
    acc = part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx];
    if (gridDim.y == 1)
      out[c] = acc;
    else
      atomicAdd(out + c, acc);
  }
}

// the same with 16-byte loads: a workgroup is 16 column lanes (16 * VEC
// columns, 256 contiguous bytes of a row) x 16 row lanes, eight rows in flight
// per thread; the row lanes are combined through LDS, so a workgroup issues
// one atomic add per column
template <typename T>
__global__ void __launch_bounds__(256)
reduce_cols_vec_kernel(const T *__restrict__ x, float *__restrict__ out,
                       int64_t rows, int64_t cols, int64_t rows_per_block) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int W = 16 * VEC;  // columns of a workgroup
  __shared__ float part[16][W + 1];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int64_t c = ((int64_t)blockIdx.x * 16 + cx) * VEC;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  if (c < cols) {
    int64_t r = r0 + ry;
    for (; r + 16 * 7 < r1; r += 16 * 8) {
      uint4 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        w[u] = *reinterpret_cast<const uint4 *>(x + (r + 16 * u) * cols + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        T e[VEC];
        __builtin_memcpy(e, &w[u], 16);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += to_f<T>(e[k]);
      }
    }
    for (; r < r1; r += 16) {
      const uint4 w = *reinterpret_cast<const uint4 *>(x + r * cols + c);
      T e[VEC];
      __builtin_memcpy(e, &w, 16);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += to_f<T>(e[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) part[ry][cx * VEC + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < W) {
    const int64_t col = (int64_t)blockIdx.x * W + threadIdx.x;
    if (col < cols) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) v += part[k][threadIdx.x];
      if (gridDim.y == 1)
        out[col] = v;
      else
        atomicAdd(out + col, v);
    }
  }
}

template <typename T>
__global__ void reduce_cols_fin_kernel(const T *__restrict__ x, ColsOut o,
                                       int64_t rows, int64_t cols,
                                       int64_t rows_per_block) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float acc = 0.f;
  if (c < cols)
    for (int64_t r = r0 + ry; r < r1; r += 4) acc += to_f<T>(x[r * cols + c]);
  part[ry][cx] = acc;
  __syncthreads();
  const bool mine = ry == 0 && c < cols;
  if (mine) acc = part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx];
  cols_finish<T>(o, mine, c, acc);
}

template <typename T>
__global__ void __launch_bounds__(256)
reduce_cols_vec_fin_kernel(const T *__restrict__ x, ColsOut o, int64_t rows,
                           int64_t cols, int64_t rows_per_block) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int W = 16 * VEC;
  __shared__ float part[16][W + 1];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int64_t c = ((int64_t)blockIdx.x * 16 + cx) * VEC;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  if (c < cols) {
    int64_t r = r0 + ry;
    for (; r + 16 * 7 < r1; r += 16 * 8) {
      uint4 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        w[u] = *reinterpret_cast<const uint4 *>(x + (r + 16 * u) * cols + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        T e[VEC];
        __builtin_memcpy(e, &w[u], 16);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += to_f<T>(e[k]);
      }
    }
    for (; r < r1; r += 16) {
      const uint4 w = *reinterpret_cast<const uint4 *>(x + r * cols + c);
      T e[VEC];
      __builtin_memcpy(e, &w, 16);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += to_f<T>(e[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) part[ry][cx * VEC + k] = acc[k];
  __syncthreads();
  const int64_t col = (int64_t)blockIdx.x * W + threadIdx.x;
  const bool mine = threadIdx.x < W && col < cols;
  float v = 0.f;
  if (mine) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][threadIdx.x];
  }
  cols_finish<T>(o, mine, col, v);
}

template <typename T>
__global__ void cast_from_f32_kernel(const float *__restrict__ in,
                                     T *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = from_f<T>(in[i]);
}

template <typename T, int OP>
__global__ void reduce_all_kernel(const T *__restrict__ x,
                                  float *__restrict__ acc_out, int64_t n) {
  __shared__ float smem[32];
  float acc = r_init<OP>();
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    acc = r_comb<OP>(acc, r_map<OP>(to_f<T>(x[i])));
  acc = r_wave<OP>(acc);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem[wid] = acc;
  __syncthreads();
  if (wid == 0) {
    float v = lane < (blockDim.x >> 6) ? smem[lane] : r_init<OP>();
    v = r_wave<OP>(v);
    if (lane == 0) {
      if (OP == R_MAX) {
        // float max through the integer atomic on the ordered bit pattern
        int *ai = reinterpret_cast<int *>(acc_out);
        int old = *ai, assumed;
        do {
          assumed = old;
          if (__int_as_float(assumed) >= v) break;
          old = atomicCAS(ai, assumed, __float_as_int(v));
        } while (assumed != old);
      } else if (OP == R_MIN) {
        int *ai = reinterpret_cast<int *>(acc_out);
        int old = *ai, assumed;
        do {
          assumed = old;
          if (__int_as_float(assumed) <= v) break;
          old = atomicCAS(ai, assumed, __float_as_int(v));
        } while (assumed != old);
      } else {
        atomicAdd(acc_out, v);
      }
    }
  }
}

__global__ void set_scalar_kernel(float *p, float v) { *p = v; }

template <typename T>
__global__ void finish_scalar_kernel(const float *acc, T *out) {
  *out = from_f<T>(*acc);
}

template <typename T, int OP>
int launch_rows(const void *x, void *out, int64_t rows, int64_t cols,
                hipStream_t s) {
  if (cols <= 1024) {
    int64_t waves = rows;
    int blocks = (int)i64min((waves + 3) / 4, kMaxBlocks);
    hipLaunchKernelGGL((reduce_rows_wave_kernel<T, OP>), dim3(blocks < 1 ? 1 : blocks),
                       dim3(256), 0, s, (const T *)x, (T *)out, rows, cols);
  } else {
    int blocks = (int)i64min(rows, kMaxBlocks);
    hipLaunchKernelGGL((reduce_rows_block_kernel<T, OP>), dim3(blocks < 1 ? 1 : blocks),
                       dim3(256), 0, s, (const T *)x, (T *)out, rows, cols);
  }
  return 0;
}

}  // namespace sa

using namespace sa;

namespace sa {
int &deterministic_flag() {
  static int flag = 0;
  return flag;
}
}  // namespace sa
void sa_set_deterministic(int on) { sa::deterministic_flag() = on; }
int sa_get_deterministic() { return sa::deterministic_flag(); }

int sa_reduce_rows(int op, int dtype, const void *x, void *out, int64_t rows,
                   int64_t cols, hipStream_t s) {
  if (rows <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    switch (op) {
      case R_SUM: return launch_rows<T, R_SUM>(x, out, rows, cols, s);
      case R_MEAN: return launch_rows<T, R_MEAN>(x, out, rows, cols, s);
      case R_MAX: return launch_rows<T, R_MAX>(x, out, rows, cols, s);
      case R_MIN: return launch_rows<T, R_MIN>(x, out, rows, cols, s);
      default: return -1;
    }
  });
  return 0;
}

int sa_reduce_cols(int dtype, const void *x, void *out, int64_t rows,
                   int64_t cols, hipStream_t s) {
  if (cols <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    constexpr int VEC = 16 / (int)sizeof(T);
    const bool vec = cols % VEC == 0 &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) {
      // 128 rows per workgroup (eight per thread, all in flight); fewer row
      // slices when the columns alone fill the chip
      const int64_t col_blocks = (cols + 16 * VEC - 1) / (16 * VEC);
      int64_t splits = i64min((rows + 127) / 128,
                              i64max(1, 2048 / col_blocks));
      if (deterministic()) splits = 1;   // one slice: a fixed summation order
      const int64_t rpb = (rows + splits - 1) / splits;
      splits = (rows + rpb - 1) / rpb;
      if (splits > 1) zero_async(out, cols * sizeof(float), s);
      hipLaunchKernelGGL((reduce_cols_vec_kernel<T>),
                         dim3((unsigned)col_blocks, (unsigned)splits),
                         dim3(256), 0, s, (const T *)x, (float *)out, rows,
                         cols, rpb);
    } else {
      const int64_t col_blocks = (cols + 63) / 64;
      int64_t splits = 1;
      if (rows > 256 && !deterministic())
        splits = i64min((rows + 255) / 256, i64max(1, 1024 / col_blocks));
      const int64_t rpb = (rows + splits - 1) / splits;
      splits = (rows + rpb - 1) / rpb;
      if (splits > 1) zero_async(out, cols * sizeof(float), s);
      hipLaunchKernelGGL((reduce_cols_kernel<T>),
                         dim3((unsigned)col_blocks, (unsigned)splits),
                         dim3(256), 0, s, (const T *)x, (float *)out, rows,
                         cols, rpb);
    }
  });
  return 0;
}

// Column sums in one launch whatever the number of row slices: out is float
// (out_typed == 0) or of x's type; sums / tickets are scratch buffers that
// hold zeros on entry and again on exit, used by one stream at a time.
// -3 when the scratch is too small for this shape.
int sa_reduce_cols_fin(int dtype, const void *x, void *out, int out_typed,
                       int64_t rows, int64_t cols, float *sums,
                       int64_t sums_cap, unsigned *tickets,
                       int64_t tickets_cap, hipStream_t s) {
  if (cols <= 0) return 0;
  if (cols > sums_cap) return -3;
  int rc = 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    constexpr int VEC = 16 / (int)sizeof(T);
    ColsOut o{out, sums, tickets, out_typed};
    const bool vec = cols % VEC == 0 &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) {
      const int64_t col_blocks = (cols + 16 * VEC - 1) / (16 * VEC);
      int64_t splits = i64min((rows + 127) / 128,
                              i64max(1, 2048 / col_blocks));
      if (deterministic()) splits = 1;   // one slice: a fixed summation order
      const int64_t rpb = (rows + splits - 1) / splits;
      splits = (rows + rpb - 1) / rpb;
      if (col_blocks > tickets_cap) rc = -3;
      else
        hipLaunchKernelGGL((reduce_cols_vec_fin_kernel<T>),
                           dim3((unsigned)col_blocks, (unsigned)splits),
                           dim3(256), 0, s, (const T *)x, o, rows, cols, rpb);
    } else {
      const int64_t col_blocks = (cols + 63) / 64;
      int64_t splits = 1;
      if (rows > 256 && !deterministic())
        splits = i64min((rows + 255) / 256, i64max(1, 1024 / col_blocks));
      const int64_t rpb = (rows + splits - 1) / splits;
      splits = (rows + rpb - 1) / rpb;
      if (col_blocks > tickets_cap) rc = -3;
      else
        hipLaunchKernelGGL((reduce_cols_fin_kernel<T>),
                           dim3((unsigned)col_blocks, (unsigned)splits),
                           dim3(256), 0, s, (const T *)x, o, rows, cols, rpb);
    }
  });
  return rc;
}

int sa_reduce_all(int op, int dtype, const void *x, void *out, int64_t n,
                  hipStream_t s) {
  // the float accumulator lives in the first 4 bytes of a static scratch
  // word per stream is not capture friendly, so `out` must provide at least
  // 4 bytes: for f32 it is the accumulator itself.
  if (dtype != kF32) return -3;
  float init = op == R_MAX ? -3.4e38f : (op == R_MIN ? 3.4e38f : 0.f);
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, s, (float *)out,
                     init);
  if (n <= 0) return 0;
  // one workgroup adds in a fixed order; several combine with atomics
  const int blocks = deterministic() ? 1 : grid_for(n, 256, 8);
  switch (op) {
    case R_SUM:
      hipLaunchKernelGGL((reduce_all_kernel<float, R_SUM>), dim3(blocks),
                         dim3(256), 0, s, (const float *)x, (float *)out, n);
      break;
    case R_ASUM:
      hipLaunchKernelGGL((reduce_all_kernel<float, R_ASUM>), dim3(blocks),
                         dim3(256), 0, s, (const float *)x, (float *)out, n);
      break;
    case R_SUMSQ:
      hipLaunchKernelGGL((reduce_all_kernel<float, R_SUMSQ>), dim3(blocks),
                         dim3(256), 0, s, (const float *)x, (float *)out, n);
      break;
    case R_MAX:
      hipLaunchKernelGGL((reduce_all_kernel<float, R_MAX>), dim3(blocks),
                         dim3(256), 0, s, (const float *)x, (float *)out, n);
      break;
    case R_MIN:
      hipLaunchKernelGGL((reduce_all_kernel<float, R_MIN>), dim3(blocks),
                         dim3(256), 0, s, (const float *)x, (float *)out, n);
      break;
    default:
      return -1;
  }
  return 0;
}
