// Winograd transforms for 3 x 3, stride 1, pad 1 convolutions.
//
// F(m x m, 3 x 3) computes an m x m block of outputs from an (m + 2)^2 input
// patch with (m + 2)^2 multiplications per (input channel, output channel)
// pair instead of 9 m^2: 2.25 x fewer for m = 2, 4 x fewer for m = 4.  The
// convolution becomes
//
//   U[p][k][c] = (G g G^T)[p]            filter transform, p = 0 .. (m+2)^2-1
//   V[p][c][t] = (B^T d B)[p]            input transform, t = image tile
//   M[p][k][t] = sum_c U[p][k][c] V[p][c][t]   (m+2)^2 independent products
//   y tile     = A^T M[.][k][t] A        output transform
//
// Only the transforms live here; the batched product runs on the float32 MFMA
// GEMM kernels (gemm.hip).  The deep, narrow layers of a residual network
// (14 x 14 and 7 x 7 images with 256 / 512 channels at batch 32) gain from the
// smaller product; on the wide early layers the (m+2)^2 / m^2 larger
// intermediate tensors cost more memory traffic than the arithmetic saves,
// which is why the caller times this route against the direct kernels per
// geometry.  The data gradient is the same computation on dy with the filter
// rotated by 180 degrees and its channel roles exchanged; the filter
// transform does both on the fly.  The reference leaves the choice of such
// algorithms to the vendor library (src/model/operation/convolution.cc:514-640).
//
// The transforms compute in float32 whatever the element type; they add and
// subtract neighbouring values, so float32 results differ from the direct
// kernels by rounding only (a few 1e-6 of the output magnitude for m = 2,
// 1e-5 for m = 4).  With 16-bit tensors the transformed operands are rounded
// to 16 bits before the product, which m = 2 (coefficients 1 and 1/2) takes
// without visible loss; the caller does not use m = 4 there.
#include "common.cuh"
#include "kernels.h"

namespace sa {

template <int M> struct WinoMat;

template <> struct WinoMat<2> {
  static constexpr int A = 4;
  static constexpr float BT[4][4] = {
      {1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
  static constexpr float G[4][3] = {
      {1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
  static constexpr float AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
};

template <> struct WinoMat<4> {
  static constexpr int A = 6;
  static constexpr float BT[6][6] = {
      {4, 0, -5, 0, 1, 0},  {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
      {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  static constexpr float G[6][3] = {
      {1.f / 4, 0, 0},
      {-1.f / 6, -1.f / 6, -1.f / 6},
      {-1.f / 6, 1.f / 6, -1.f / 6},
      {1.f / 24, 1.f / 12, 1.f / 6},
      {1.f / 24, -1.f / 12, 1.f / 6},
      {0, 0, 1}};
  static constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0},
                                     {0, 1, -1, 2, -2, 0},
                                     {0, 1, 1, 4, 4, 0},
                                     {0, 1, -1, 8, -8, 1}};
};

// one thread per (output channel, input channel) pair of the product:
// (ko, ci) = (k, c) of w[K][C][3][3] forward, (c, k) with the taps reversed for
// the data gradient
template <int M, typename T>
__device__ __forceinline__ void wino_filter_body(int64_t idx,
                                                 const T *__restrict__ w,
                                                 T *__restrict__ U, int K,
                                                 int C, int flip) {
  using W = WinoMat<M>;
  constexpr int A = W::A;
  const int KO = flip ? C : K, CI = flip ? K : C;
  if (idx >= (int64_t)KO * CI) return;
  const int ci = (int)(idx % CI), ko = (int)(idx / CI);
  const T *src = flip ? w + ((int64_t)ci * C + ko) * 9
                      : w + ((int64_t)ko * C + ci) * 9;
  float g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      g[r][s] = to_f<T>(flip ? src[(2 - r) * 3 + (2 - s)] : src[r * 3 + s]);
  float t[A][3];
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      t[i][s] = W::G[i][0] * g[0][s] + W::G[i][1] * g[1][s] +
                W::G[i][2] * g[2][s];
  const int64_t plane = (int64_t)KO * CI;
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j)
      U[(i * A + j) * plane + idx] = from_f<T>(t[i][0] * W::G[j][0] +
                                               t[i][1] * W::G[j][1] +
                                               t[i][2] * W::G[j][2]);
}

// one thread per (channel, tile); tiles are numbered (n, tile row, tile
// column) with the column fastest, so a wave writes consecutive words of V
// With Tp < T the tiles are cut into T / Tp slices and slice s of plane p is
// the matrix [(p * S + s)][C][Tp]: the weight gradient sums over the tiles,
// and the slices are the split of that reduction over more workgroups.
template <int M, typename TT>
__device__ __forceinline__ void wino_input_body(int64_t idx,
                                                const TT *__restrict__ x,
                                                TT *__restrict__ V, int C,
                                                int H, int Wd, int th, int tw,
                                                int64_t T, int64_t Tp) {
  using W = WinoMat<M>;
  constexpr int A = W::A;
  if (idx >= (int64_t)C * T) return;
  const int64_t t = idx % T;
  const int c = (int)(idx / T);
  const int tc = (int)(t % tw);
  const int tr = (int)((t / tw) % th);
  const int n = (int)(t / ((int64_t)tw * th));
  const TT *img = x + ((int64_t)n * C + c) * H * Wd;
  const int h0 = tr * M - 1, w0 = tc * M - 1;
  float d[A][A];
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const int h = h0 + a;
    const bool hin = h >= 0 && h < H;
#pragma unroll
    for (int b = 0; b < A; ++b) {
      const int ww = w0 + b;
      d[a][b] = (hin && ww >= 0 && ww < Wd)
                    ? to_f<TT>(img[(int64_t)h * Wd + ww]) : 0.f;
    }
  }
  float u[A][A];  // B^T d
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int b = 0; b < A; ++b) {
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < A; ++a)
        if (W::BT[i][a] != 0.f) s += W::BT[i][a] * d[a][b];
      u[i][b] = s;
    }
  const int64_t plane = (int64_t)C * T;
  const int64_t at = ((t / Tp) * C + c) * Tp + t % Tp;
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float s = 0.f;
#pragma unroll
      for (int b = 0; b < A; ++b)
        if (W::BT[j][b] != 0.f) s += u[i][b] * W::BT[j][b];
      V[(i * A + j) * plane + at] = from_f<TT>(s);
    }
}

// dM[p][k][t] = (A dY A^T)[p] of the m x m tile t of dy (zero outside the
// image): the gradient of the loss with respect to the products M.  Same
// slice layout as wino_input_kernel.
template <int M, typename TT>
__device__ __forceinline__ void wino_outgrad_body(
    int64_t idx, const TT *__restrict__ dy, TT *__restrict__ dM, int K,
    int H, int Wd, int th, int tw, int64_t T, int64_t Tp) {
  using W = WinoMat<M>;
  constexpr int A = W::A;
  if (idx >= (int64_t)K * T) return;
  const int64_t t = idx % T;
  const int k = (int)(idx / T);
  const int tc = (int)(t % tw);
  const int tr = (int)((t / tw) % th);
  const int n = (int)(t / ((int64_t)tw * th));
  const TT *img = dy + ((int64_t)n * K + k) * H * Wd;
  float d[M][M];
#pragma unroll
  for (int a = 0; a < M; ++a) {
    const int h = tr * M + a;
#pragma unroll
    for (int b = 0; b < M; ++b) {
      const int ww = tc * M + b;
      d[a][b] = (h < H && ww < Wd) ? to_f<TT>(img[(int64_t)h * Wd + ww]) : 0.f;
    }
  }
  float u[A][M];  // A dY
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int b = 0; b < M; ++b) {
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < M; ++a)
        if (W::AT[a][i] != 0.f) s += W::AT[a][i] * d[a][b];
      u[i][b] = s;
    }
  const int64_t plane = (int64_t)K * T;
  const int64_t at = ((t / Tp) * K + k) * Tp + t % Tp;
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float s = 0.f;
#pragma unroll
      for (int b = 0; b < M; ++b)
        if (W::AT[b][j] != 0.f) s += u[i][b] * W::AT[b][j];
      dM[(i * A + j) * plane + at] = from_f<TT>(s);
    }
}

// What has to happen before the batched product, as ONE launch (the pieces
// are independent and short; at batch 32 a launch of its own costs each of
// them more than its work): the first `nb_first` workgroups transform the
// image, the others the filter (forward / data gradient) or dy (weight
// gradient).
template <int M, typename TT>
__global__ void wino_prepare_kernel(const TT *__restrict__ x,
                                    TT *__restrict__ V, int C, int H,
                                    int Wd, int th, int tw, int64_t T,
                                    const TT *__restrict__ w,
                                    TT *__restrict__ U, int K, int Cw,
                                    int flip, int nb_first) {
  if ((int)blockIdx.x < nb_first)
    wino_input_body<M, TT>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, x, V,
                       C, H, Wd, th, tw, T, T);
  else
    wino_filter_body<M, TT>(
        (int64_t)(blockIdx.x - nb_first) * blockDim.x + threadIdx.x, w, U, K,
        Cw, flip);
}

template <int M, typename TT>
__global__ void wino_prepare_grad_kernel(const TT *__restrict__ x,
                                         TT *__restrict__ V, int C,
                                         const TT *__restrict__ dy,
                                         TT *__restrict__ dM, int K, int H,
                                         int Wd, int th, int tw, int64_t T,
                                         int64_t Tp, int nb_first) {
  if ((int)blockIdx.x < nb_first)
    wino_input_body<M, TT>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, x, V,
                       C, H, Wd, th, tw, T, Tp);
  else
    wino_outgrad_body<M, TT>(
        (int64_t)(blockIdx.x - nb_first) * blockDim.x + threadIdx.x, dy, dM,
        K, H, Wd, th, tw, T, Tp);
}

template <int M, typename TT>
__global__ void wino_filter_kernel(const TT *__restrict__ w,
                                   TT *__restrict__ U, int K, int C,
                                   int flip) {
  wino_filter_body<M, TT>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, w, U, K,
                      C, flip);
}

template <int M, typename TT>
__global__ void wino_input_kernel(const TT *__restrict__ x,
                                  TT *__restrict__ V, int C, int H, int Wd,
                                  int th, int tw, int64_t T, int64_t Tp) {
  wino_input_body<M, TT>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, x, V, C,
                     H, Wd, th, tw, T, Tp);
}

template <int M, typename TT>
__global__ void wino_outgrad_kernel(const TT *__restrict__ dy,
                                    TT *__restrict__ dM, int K, int H,
                                    int Wd, int th, int tw, int64_t T,
                                    int64_t Tp) {
  wino_outgrad_body<M, TT>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, dy, dM,
                       K, H, Wd, th, tw, T, Tp);
}

// dw[k][c] (3 x 3) = G^T (sum over the S slices of dU[p * S + s][k][c]) G
template <int M, typename TT, typename TI = TT>
__global__ void wino_filtergrad_kernel(const TI *__restrict__ dU,
                                       TT *__restrict__ dw, int K, int C,
                                       int S) {
  using W = WinoMat<M>;
  constexpr int A = W::A;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t plane = (int64_t)K * C;
  if (idx >= plane) return;
  float t[3][A];  // G^T dU
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < A; ++j) t[r][j] = 0.f;
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      const TI *src = dU + (int64_t)(i * A + j) * S * plane + idx;
      float v = 0.f;
      for (int s = 0; s < S; ++s) v += to_f<TI>(src[s * plane]);
#pragma unroll
      for (int r = 0; r < 3; ++r)
        if (W::G[i][r] != 0.f) t[r][j] += W::G[i][r] * v;
    }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < A; ++j)
        if (W::G[j][q] != 0.f) v += t[r][j] * W::G[j][q];
      dw[idx * 9 + r * 3 + q] = from_f<TT>(v);
    }
}

// one thread per (output channel, tile): y tile = A^T m A [+ bias] [+ addend]
template <int M, typename TT>
__global__ void wino_output_kernel(const TT *__restrict__ Mm,
                                   const TT *__restrict__ bias,
                                   const TT *__restrict__ addend,
                                   TT *__restrict__ y, int K, int H, int Wd,
                                   int th, int tw, int64_t T) {
  using W = WinoMat<M>;
  constexpr int A = W::A;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)K * T) return;
  const int64_t t = idx % T;
  const int k = (int)(idx / T);
  const int tc = (int)(t % tw);
  const int tr = (int)((t / tw) % th);
  const int n = (int)(t / ((int64_t)tw * th));
  const int64_t plane = (int64_t)K * T;
  float m[A][A];
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j)
      m[i][j] = to_f<TT>(Mm[(i * A + j) * plane + idx]);
  float u[M][A];  // A^T m
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < A; ++i)
        if (W::AT[a][i] != 0.f) s += W::AT[a][i] * m[i][j];
      u[a][j] = s;
    }
  const float bk = bias ? to_f<TT>(bias[k]) : 0.f;
  const int64_t base = ((int64_t)n * K + k) * H * Wd;
#pragma unroll
  for (int a = 0; a < M; ++a) {
    const int h = tr * M + a;
    if (h >= H) continue;
#pragma unroll
    for (int b = 0; b < M; ++b) {
      const int ww = tc * M + b;
      if (ww >= Wd) continue;
      float s = bk;
#pragma unroll
      for (int j = 0; j < A; ++j)
        if (W::AT[b][j] != 0.f) s += u[a][j] * W::AT[b][j];
      const int64_t o = base + (int64_t)h * Wd + ww;
      y[o] = from_f<TT>(addend ? s + to_f<TT>(addend[o]) : s);
    }
  }
}

}  // namespace sa

using namespace sa;

static int wino_grid(int64_t n) { return (int)((n + 255) / 256); }

// m = 2 or 4 and the element type become template arguments
#define WINO_DISPATCH(m, dtype, KERNEL, ...)                               \
  SA_DISPATCH_FLOAT(dtype, TT, {                                           \
    if (m == 2)                                                            \
      hipLaunchKernelGGL((KERNEL<2, TT>), __VA_ARGS__);                    \
    else                                                                   \
      hipLaunchKernelGGL((KERNEL<4, TT>), __VA_ARGS__);                    \
  })

int sa_wino_filter(int dtype, int m, const void *w, void *U, int K, int C,
                   int flip, hipStream_t s) {
  if (m != 2 && m != 4) return -2;
  if (K <= 0 || C <= 0 || (int64_t)K * C > ((int64_t)1 << 38)) return -2;
  const dim3 grid(wino_grid((int64_t)K * C)), block(256);
  WINO_DISPATCH(m, dtype, wino_filter_kernel, grid, block, 0, s,
                (const TT *)w, (TT *)U, K, C, flip)
  return 0;
}

int sa_wino_input(int dtype, int m, const void *x, void *V, int N, int C,
                  int H, int W, int slices, hipStream_t s) {
  if (m != 2 && m != 4) return -2;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || slices <= 0) return -2;
  const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
  const int64_t T = (int64_t)N * th * tw;
  if (T % slices) return -2;
  const int64_t Tp = T / slices;
  if ((int64_t)C * T > ((int64_t)1 << 38)) return -2;
  const dim3 grid(wino_grid((int64_t)C * T)), block(256);
  WINO_DISPATCH(m, dtype, wino_input_kernel, grid, block, 0, s,
                (const TT *)x, (TT *)V, C, H, W, th, tw, T, Tp)
  return 0;
}

int sa_wino_outgrad(int dtype, int m, const void *dy, void *dM, int N, int K,
                    int H, int W, int slices, hipStream_t s) {
  if (m != 2 && m != 4) return -2;
  if (N <= 0 || K <= 0 || H <= 0 || W <= 0 || slices <= 0) return -2;
  const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
  const int64_t T = (int64_t)N * th * tw;
  if (T % slices || (int64_t)K * T > ((int64_t)1 << 38)) return -2;
  const int64_t Tp = T / slices;
  const dim3 grid(wino_grid((int64_t)K * T)), block(256);
  WINO_DISPATCH(m, dtype, wino_outgrad_kernel, grid, block, 0, s,
                (const TT *)dy, (TT *)dM, K, H, W, th, tw, T, Tp)
  return 0;
}

int sa_wino_filtergrad(int dtype, int m, const void *dU, int du_float,
                       void *dw, int K, int C, int slices, hipStream_t s) {
  if (m != 2 && m != 4) return -2;
  if (K <= 0 || C <= 0 || slices <= 0) return -2;
  const dim3 grid(wino_grid((int64_t)K * C)), block(256);
  if (du_float && dtype != kF32) {
    // float products of 16-bit operands, gradient in the 16-bit type
    SA_DISPATCH_FLOAT(dtype, TT, {
      if (m == 2)
        hipLaunchKernelGGL((wino_filtergrad_kernel<2, TT, float>), grid,
                           block, 0, s, (const float *)dU, (TT *)dw, K, C,
                           slices);
      else
        hipLaunchKernelGGL((wino_filtergrad_kernel<4, TT, float>), grid,
                           block, 0, s, (const float *)dU, (TT *)dw, K, C,
                           slices);
    })
    return 0;
  }
  WINO_DISPATCH(m, dtype, wino_filtergrad_kernel, grid, block, 0, s,
                (const TT *)dU, (TT *)dw, K, C, slices)
  return 0;
}

int sa_wino_output(int dtype, int m, const void *M, const void *bias,
                   const void *addend, void *y, int N, int K, int H, int W,
                   hipStream_t s) {
  if (m != 2 && m != 4) return -2;
  if (N <= 0 || K <= 0 || H <= 0 || W <= 0) return -2;
  const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
  const int64_t T = (int64_t)N * th * tw;
  if ((int64_t)K * T > ((int64_t)1 << 38)) return -2;
  const dim3 grid(wino_grid((int64_t)K * T)), block(256);
  WINO_DISPATCH(m, dtype, wino_output_kernel, grid, block, 0, s,
                (const TT *)M, (const TT *)bias, (const TT *)addend, (TT *)y,
                K, H, W, th, tw, T)
  return 0;
}

int sa_wino_prepare(int dtype, int m, const void *x, void *V, int N, int CI,
                    int H, int W, const void *w, void *U, int K, int C,
                    int flip, hipStream_t s) {
  if (m != 2 && m != 4) return -2;
  if (N <= 0 || CI <= 0 || H <= 0 || W <= 0 || K <= 0 || C <= 0) return -2;
  if (CI != (flip ? K : C)) return -2;
  const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
  const int64_t T = (int64_t)N * th * tw;
  if ((int64_t)CI * T > ((int64_t)1 << 38)) return -2;
  const int nb_first = wino_grid((int64_t)CI * T);
  const dim3 grid(nb_first + wino_grid((int64_t)K * C)), block(256);
  WINO_DISPATCH(m, dtype, wino_prepare_kernel, grid, block, 0, s,
                (const TT *)x, (TT *)V, CI, H, W, th, tw, T, (const TT *)w,
                (TT *)U, K, C, flip, nb_first)
  return 0;
}

int sa_wino_prepare_grad(int dtype, int m, const void *x, void *V,
                         const void *dy, void *dM, int N, int C, int K, int H,
                         int W, int slices, hipStream_t s) {
  if (m != 2 && m != 4) return -2;
  if (N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || slices <= 0) return -2;
  const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
  const int64_t T = (int64_t)N * th * tw;
  if (T % slices || (int64_t)(C > K ? C : K) * T > ((int64_t)1 << 38))
    return -2;
  const int nb_first = wino_grid((int64_t)C * T);
  const dim3 grid(nb_first + wino_grid((int64_t)K * T)), block(256);
  WINO_DISPATCH(m, dtype, wino_prepare_grad_kernel, grid, block, 0, s,
                (const TT *)x, (TT *)V, C, (const TT *)dy, (TT *)dM, K, H, W,
                th, tw, T, T / slices, nb_first)
  return 0;
}

// ---- the whole route behind one call (the C++ API's convolution uses it) --------
int64_t sa_conv3x3_winograd_workspace(int m, int N, int C, int K, int H,
                                      int W) {
  if ((m != 2 && m != 4) || N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0)
    return -1;
  const int64_t P = (int64_t)(m + 2) * (m + 2);
  const int64_t T = (int64_t)N * ((H + m - 1) / m) * ((W + m - 1) / m);
  return P * ((int64_t)K * C + (int64_t)C * T + (int64_t)K * T);
}

// mode 0: out[N][K][H][W] = conv3x3(act[N][C][H][W], w[K][C][3][3]) + bias;
// mode 1: out[N][C][H][W] = data gradient of act[N][K][H][W] (+ addend).
// `workspace` holds sa_conv3x3_winograd_workspace() elements of `dtype`.
int sa_conv3x3_winograd(int dtype, int m, int mode, const void *act,
                        const void *w, const void *bias, const void *addend,
                        void *out, int N, int C, int K, int H, int W,
                        void *workspace, hipStream_t s) {
  if (mode != 0 && mode != 1) return -2;
  if (sa_conv3x3_winograd_workspace(m, N, C, K, H, W) < 0 || !workspace)
    return -2;
  const int CI = mode == 0 ? C : K, KO = mode == 0 ? K : C;
  const int64_t P = (int64_t)(m + 2) * (m + 2);
  const int64_t T = (int64_t)N * ((H + m - 1) / m) * ((W + m - 1) / m);
  const int64_t esize = dtype == kF32 ? 4 : 2;
  char *base = static_cast<char *>(workspace);
  void *U = base;
  void *V = base + P * KO * CI * esize;
  void *M = base + P * (KO * (int64_t)CI + CI * T) * esize;
  int rc = sa_wino_prepare(dtype, m, act, V, N, CI, H, W, w, U, K, C,
                           mode == 1, s);
  if (rc != 0) return rc;
  // M[p] (KO x T) = U[p] (KO x CI) * V[p] (CI x T), P products in one launch
  rc = sa_gemm(dtype, KO, T, CI, 1.f, U, CI, 1, (int64_t)KO * CI, V, T, 1,
               CI * T, 0.f, M, T, KO * T, P, nullptr, 0, s);
  if (rc != 0) return rc;
  return sa_wino_output(dtype, m, M, bias, addend, out, N, KO, H, W, s);
}
