// Depthwise convolution (groups == C == K: every output channel filters one
// input channel).  As an implicit GEMM each group would be a 1 x (R*S) x (N*P*Q)
// product that fills 1/64 of an MFMA tile, so these layers get kernels of
// their own: they are bound by memory traffic, one thread per output element
// with the filter taps of a channel coming out of the scalar / L1 path, and
// the filter gradient is a per-channel block reduction.
// Role parity: the grouped cuDNN convolution the reference uses for
// SeparableConv2d (src/model/operation/convolution.cc:230-330 with
// group == channels).
#include <algorithm>

#include "common.cuh"
#include "kernels.h"

namespace sa {

struct DwArgs {
  int N, C, H, W, R, S, P, Q;
  int sh, sw, ph, pw;
};

static DwArgs dw_args(const ConvGeom &g) {
  return DwArgs{g.N, g.C, g.H, g.W, g.R, g.S, g.P, g.Q,
                g.stride_h, g.stride_w, g.pad_h, g.pad_w};
}

template <typename T>
__global__ void __launch_bounds__(256)
dw_fwd_kernel(const T *__restrict__ x, const T *__restrict__ w,
              const T *__restrict__ bias, T *__restrict__ y, DwArgs a,
              int total) {
  const int step = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const int q = i % a.Q;
    int t = i / a.Q;
    const int p = t % a.P;
    const int nc = t / a.P;
    const int c = nc % a.C;
    const T *xp = x + (int64_t)nc * a.H * a.W;
    const T *wp = w + c * a.R * a.S;
    const int h0 = p * a.sh - a.ph, w0 = q * a.sw - a.pw;
    float acc = bias ? to_f(bias[c]) : 0.f;
    for (int r = 0; r < a.R; ++r) {
      const int h = h0 + r;
      if (h < 0 || h >= a.H) continue;
      for (int s = 0; s < a.S; ++s) {
        const int ww = w0 + s;
        if (ww < 0 || ww >= a.W) continue;
        acc += to_f(xp[h * a.W + ww]) * to_f(wp[r * a.S + s]);
      }
    }
    y[i] = from_f<T>(acc);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
dw_bwd_data_kernel(const T *__restrict__ dy, const T *__restrict__ w,
                   const T *__restrict__ addend, T *__restrict__ dx, DwArgs a,
                   int total) {
  const int step = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const int ww = i % a.W;
    int t = i / a.W;
    const int h = t % a.H;
    const int nc = t / a.H;
    const int c = nc % a.C;
    const T *dyp = dy + (int64_t)nc * a.P * a.Q;
    const T *wp = w + c * a.R * a.S;
    float acc = addend ? to_f(addend[i]) : 0.f;
    for (int r = 0; r < a.R; ++r) {
      const int hh = h + a.ph - r;
      if (hh < 0 || hh % a.sh) continue;
      const int p = hh / a.sh;
      if (p >= a.P) continue;
      for (int s = 0; s < a.S; ++s) {
        const int wq = ww + a.pw - s;
        if (wq < 0 || wq % a.sw) continue;
        const int q = wq / a.sw;
        if (q >= a.Q) continue;
        acc += to_f(dyp[p * a.Q + q]) * to_f(wp[r * a.S + s]);
      }
    }
    dx[i] = from_f<T>(acc);
  }
}

// Stride-1 planes through LDS: a block stages `ppb` whole input planes (small
// images) or one band of rows of one plane (large images) into a zero-bordered
// float tile, so the R*S taps of an output are unconditional LDS reads.  With
// FLIP the filter is applied rotated by 180 degrees, which turns the same loop
// into backward-data (in = dy, out = dx, pad = R-1-pad).
template <typename T, int R, int S, bool FLIP>
__global__ void __launch_bounds__(256)
dw_plane_kernel(const T *__restrict__ in, const T *__restrict__ w,
                const T *__restrict__ bias, const T *__restrict__ addend,
                T *__restrict__ out, int planes, int C, int IH, int IW, int OH,
                int OW, int padh, int padw, int ppb, int band, int nbands) {
  extern __shared__ float lds[];
  const int LW = OW + S - 1;
  const int LH = band + R - 1;
  float *taps = lds + ppb * LH * LW;
  const int blk = blockIdx.x / nbands;
  const int p0 = (blockIdx.x - blk * nbands) * band;   // first output row
  const int plane0 = blk * ppb;
  const int np = min(ppb, planes - plane0);
  const int rows = min(band, OH - p0);
  for (int i = threadIdx.x; i < np * R * S; i += blockDim.x) {
    const int j = i / (R * S), k = i - j * (R * S);
    const int c = (plane0 + j) % C;
    taps[i] = to_f(w[c * R * S + (FLIP ? R * S - 1 - k : k)]);
  }
  const int fill = np * LH * LW;
  for (int i = threadIdx.x; i < fill; i += blockDim.x) {
    const int j = i / (LH * LW);
    const int rem = i - j * (LH * LW);
    const int lr = rem / LW, lc = rem - lr * LW;
    const int ir = p0 + lr - padh, ic = lc - padw;
    float v = 0.f;
    if (ir >= 0 && ir < IH && ic >= 0 && ic < IW)
      v = to_f(in[((int64_t)(plane0 + j) * IH + ir) * IW + ic]);
    lds[i] = v;
  }
  __syncthreads();
  const int work = np * rows * OW;
  for (int i = threadIdx.x; i < work; i += blockDim.x) {
    const int j = i / (rows * OW);
    const int rem = i - j * (rows * OW);
    const int p = rem / OW, q = rem - p * OW;
    const float *t = lds + j * LH * LW + p * LW + q;
    const float *k = taps + j * R * S;
    float acc = bias ? to_f(bias[(plane0 + j) % C]) : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) acc += t[r * LW + s2] * k[r * S + s2];
    const int64_t o = ((int64_t)(plane0 + j) * OH + p0 + p) * OW + q;
    if (addend) acc += to_f(addend[o]);
    out[o] = from_f<T>(acc);
  }
}

// false when the geometry is outside what the LDS kernel covers
template <typename T, int R, int S, bool FLIP>
static bool launch_dw_plane(const void *in, const void *w, const void *bias,
                            const void *addend, void *out, int planes, int C,
                            int IH, int IW, int OH, int OW, int padh, int padw,
                            hipStream_t s) {
  if (padh < 0 || padw < 0 || padh > R - 1 || padw > S - 1) return false;
  const int LW = OW + S - 1;
  const int kLimit = 12 * 1024;  // floats of LDS per block: 3 blocks per CU
  if ((R + 7) * LW > kLimit) return false;
  int band = OH, nbands = 1, ppb = 1;
  if ((OH + R - 1) * LW > kLimit) {
    band = kLimit / LW - (R - 1);
    nbands = (OH + band - 1) / band;
    band = (OH + nbands - 1) / nbands;   // even bands
  } else {
    ppb = std::max(1, std::min(kLimit / ((OH + R - 1) * LW),
                               2048 / std::max(1, OH * OW)));
  }
  const int64_t blocks = (int64_t)((planes + ppb - 1) / ppb) * nbands;
  if (blocks > 0x7fffffff) return false;
  const size_t bytes =
      sizeof(float) * ((size_t)ppb * (band + R - 1) * LW + (size_t)ppb * R * S);
  hipLaunchKernelGGL((dw_plane_kernel<T, R, S, FLIP>), dim3((unsigned)blocks),
                     dim3(256), bytes, s, (const T *)in, (const T *)w,
                     (const T *)bias, (const T *)addend, (T *)out, planes, C,
                     IH, IW, OH, OW, padh, padw, ppb, band, nbands);
  return true;
}

// grid (C, nsplit): block (c, j) reduces images j, j + nsplit, ... of channel
// c; with nsplit > 1 the partial sums meet in dw through atomics (dw zeroed by
// the launcher).
template <typename T, int R, int S>
__global__ void __launch_bounds__(256)
dw_bwd_weight_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                     float *__restrict__ dw, DwArgs a) {
  __shared__ float red[17];
  const int c = blockIdx.x;
  float acc[R * S];
#pragma unroll
  for (int k = 0; k < R * S; ++k) acc[k] = 0.f;
  const int PQ = a.P * a.Q;
  for (int n = blockIdx.y; n < a.N; n += gridDim.y) {
    const T *dyp = dy + ((int64_t)n * a.C + c) * PQ;
    const T *xp = x + ((int64_t)n * a.C + c) * a.H * a.W;
    for (int i = threadIdx.x; i < PQ; i += blockDim.x) {
      const int p = i / a.Q, q = i - p * a.Q;
      const float g = to_f(dyp[i]);
      const int h0 = p * a.sh - a.ph, w0 = q * a.sw - a.pw;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int h = h0 + r;
        const bool hin = h >= 0 && h < a.H;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int ww = w0 + s;
          if (hin && ww >= 0 && ww < a.W)
            acc[r * S + s] += g * to_f(xp[h * a.W + ww]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < R * S; ++k) {
    const float v = block_sum(acc[k], red);
    if (threadIdx.x == 0) {
      if (gridDim.y > 1)
        atomicAdd(dw + c * R * S + k, v);
      else
        dw[c * R * S + k] = v;
    }
  }
}

// Stride-1 filter gradient through LDS: block (c, j) walks its images in
// stages of `ipb` whole planes (small images) or one band of rows (large
// images); a stage holds the zero-bordered x rows and the dy rows it needs, so
// the R*S products of an output position are unconditional LDS reads.
template <typename T, int R, int S>
__global__ void __launch_bounds__(256)
dw_bwd_weight_lds_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                         float *__restrict__ dw, DwArgs a, int ipb, int band,
                         int nbands) {
  extern __shared__ float lds[];
  __shared__ float red[17];
  const int c = blockIdx.x;
  const int LW = a.Q + S - 1, LH = band + R - 1;
  float *gl = lds + ipb * LH * LW;   // dy rows of the stage
  float acc[R * S];
#pragma unroll
  for (int k = 0; k < R * S; ++k) acc[k] = 0.f;
  // images of this block: blockIdx.y, blockIdx.y + gridDim.y, ...
  const int mine = (a.N - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
  for (int first = 0; first < mine; first += ipb) {
    const int ni = min(ipb, mine - first);
    for (int b = 0; b < nbands; ++b) {
      const int p0 = b * band;
      const int rows = min(band, a.P - p0);
      __syncthreads();
      for (int i = threadIdx.x; i < ni * LH * LW; i += blockDim.x) {
        const int j = i / (LH * LW);
        const int rem = i - j * (LH * LW);
        const int lr = rem / LW, lc = rem - lr * LW;
        const int n = blockIdx.y + (first + j) * gridDim.y;
        const int h = p0 + lr - a.ph, w = lc - a.pw;
        float v = 0.f;
        if (h >= 0 && h < a.H && w >= 0 && w < a.W)
          v = to_f(x[(((int64_t)n * a.C + c) * a.H + h) * a.W + w]);
        lds[i] = v;
      }
      for (int i = threadIdx.x; i < ni * rows * a.Q; i += blockDim.x) {
        const int j = i / (rows * a.Q);
        const int rem = i - j * (rows * a.Q);
        const int n = blockIdx.y + (first + j) * gridDim.y;
        gl[j * band * a.Q + rem] =
            to_f(dy[(((int64_t)n * a.C + c) * a.P + p0) * a.Q + rem]);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < ni * rows * a.Q; i += blockDim.x) {
        const int j = i / (rows * a.Q);
        const int rem = i - j * (rows * a.Q);
        const int p = rem / a.Q, q = rem - p * a.Q;
        const float g = gl[j * band * a.Q + rem];
        const float *t = lds + j * LH * LW + p * LW + q;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int s2 = 0; s2 < S; ++s2) acc[r * S + s2] += g * t[r * LW + s2];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < R * S; ++k) {
    const float v = block_sum(acc[k], red);
    if (threadIdx.x == 0) {
      if (gridDim.y > 1)
        atomicAdd(dw + c * R * S + k, v);
      else
        dw[c * R * S + k] = v;
    }
  }
}

template <typename T, int R, int S>
static void launch_dw_bwd_weight(const ConvGeom &g, const void *dy,
                                 const void *x, float *dw, hipStream_t s) {
  // enough blocks to cover the chip several times over, never more than N
  int nsplit = 1;
  while (g.C * nsplit < 1024 && nsplit * 2 <= g.N) nsplit *= 2;
  if (nsplit > 1) zero_async(dw, sizeof(float) * g.C * R * S, s);
  const int LW = g.Q + S - 1;
  const int kLimit = 12 * 1024;  // floats of LDS per block
  if (g.stride_h == 1 && g.stride_w == 1 && g.pad_h < R && g.pad_w < S &&
      g.P == g.H + 2 * g.pad_h - R + 1 && g.Q == g.W + 2 * g.pad_w - S + 1 &&
      (R + 7) * LW + 8 * g.Q <= kLimit) {
    int band = g.P, nbands = 1, ipb = 1;
    const int plane = (g.P + R - 1) * LW + g.P * g.Q;
    if (plane > kLimit) {
      band = (kLimit - (R - 1) * LW) / (LW + g.Q);
      nbands = (g.P + band - 1) / band;
      band = (g.P + nbands - 1) / nbands;
    } else {
      const int per_block = (g.N + nsplit - 1) / nsplit;
      ipb = std::max(1, std::min({kLimit / plane, per_block,
                                  4096 / std::max(1, g.P * g.Q) + 1}));
    }
    const size_t bytes =
        sizeof(float) * (size_t)ipb * ((band + R - 1) * LW + band * g.Q);
    hipLaunchKernelGGL((dw_bwd_weight_lds_kernel<T, R, S>), dim3(g.C, nsplit),
                       dim3(256), bytes, s, (const T *)dy, (const T *)x, dw,
                       dw_args(g), ipb, band, nbands);
    return;
  }
  hipLaunchKernelGGL((dw_bwd_weight_kernel<T, R, S>), dim3(g.C, nsplit),
                     dim3(256), 0, s, (const T *)dy, (const T *)x, dw,
                     dw_args(g));
}

}  // namespace sa

using namespace sa;

bool sa_dwconv_applies(const ConvGeom *g) {
  // the grid-stride loops index with 32-bit ints: leave one full stride of
  // headroom below 2^31
  const int64_t lim = ((int64_t)1 << 31) - ((int64_t)1 << 22);
  return g->groups > 1 && g->groups == g->C && g->K == g->C &&
         (int64_t)g->N * g->C * g->H * g->W < lim &&
         (int64_t)g->N * g->C * g->P * g->Q < lim;
}

bool sa_dwconv_bwd_weight_applies(const ConvGeom *g) {
  return sa_dwconv_applies(g) && g->R == g->S &&
         (g->R == 3 || g->R == 5 || g->R == 7);
}

int sa_dwconv_fwd(int dtype, const ConvGeom *g, const void *x, const void *w,
                  const void *bias, void *y, hipStream_t s) {
  const int64_t total = (int64_t)g->N * g->C * g->P * g->Q;
  if (total == 0) return 0;
  if (g->stride_h == 1 && g->stride_w == 1 && g->R == 3 && g->S == 3) {
    bool done = false;
    SA_DISPATCH_FLOAT(dtype, T, {
      done = launch_dw_plane<T, 3, 3, false>(
          x, w, bias, nullptr, y, g->N * g->C, g->C, g->H, g->W, g->P, g->Q,
          g->pad_h, g->pad_w, s);
    });
    if (done) return 0;
  }
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((dw_fwd_kernel<T>), dim3(grid_for(total, 256, 2)),
                       dim3(256), 0, s, (const T *)x, (const T *)w,
                       (const T *)bias, (T *)y, dw_args(*g), (int)total);
  });
  return 0;
}

int sa_dwconv_bwd_data(int dtype, const ConvGeom *g, const void *dy,
                       const void *w, const void *addend, void *dx,
                       hipStream_t s) {
  const int64_t total = (int64_t)g->N * g->C * g->H * g->W;
  if (total == 0) return 0;
  if (g->stride_h == 1 && g->stride_w == 1 && g->R == 3 && g->S == 3 &&
      g->P == g->H + 2 * g->pad_h - 2 && g->Q == g->W + 2 * g->pad_w - 2) {
    bool done = false;
    SA_DISPATCH_FLOAT(dtype, T, {
      done = launch_dw_plane<T, 3, 3, true>(
          dy, w, nullptr, addend, dx, g->N * g->C, g->C, g->P, g->Q, g->H,
          g->W, 2 - g->pad_h, 2 - g->pad_w, s);
    });
    if (done) return 0;
  }
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((dw_bwd_data_kernel<T>), dim3(grid_for(total, 256, 2)),
                       dim3(256), 0, s, (const T *)dy, (const T *)w,
                       (const T *)addend, (T *)dx, dw_args(*g), (int)total);
  });
  return 0;
}

int sa_dwconv_bwd_weight(int dtype, const ConvGeom *g, const void *dy,
                         const void *x, void *dw, hipStream_t s) {
  if (!sa_dwconv_bwd_weight_applies(g)) return -2;
  SA_DISPATCH_FLOAT(dtype, T, {
    if (g->R == 3)
      launch_dw_bwd_weight<T, 3, 3>(*g, dy, x, (float *)dw, s);
    else if (g->R == 5)
      launch_dw_bwd_weight<T, 5, 5>(*g, dy, x, (float *)dw, s);
    else
      launch_dw_bwd_weight<T, 7, 7>(*g, dy, x, (float *)dw, s);
  });
  return 0;
}
