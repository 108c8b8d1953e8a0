// 2-d max / average pooling, NCHW.  One thread per output element forward;
// backward is gather-style (one thread per *input* element walks the output
// windows that cover it), so no atomics are needed and results are
// deterministic.  Max backward routes the gradient to every input position
// equal to the window maximum, which needs only x and y like the cuDNN call
// it replaces; average pooling divides by the number of in-bounds cells.
//
// Replaces cudnnPoolingForward / cudnnPoolingBackward of the reference
// (src/model/operation/pooling.cc:120-227).
#include "common.cuh"
#include "kernels.h"

namespace sa {

template <typename T>
__global__ void pool_fwd_kernel(PoolGeom g, const T *__restrict__ x,
                                T *__restrict__ y, int64_t total) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    const int q = i % g.Q;
    const int p = (i / g.Q) % g.P;
    const int64_t nc = i / ((int64_t)g.P * g.Q);
    const T *src = x + nc * g.H * g.W;
    const int h0 = p * g.sh - g.ph, w0 = q * g.sw - g.pw;
    const int h1 = min(h0 + g.kh, g.H), w1 = min(w0 + g.kw, g.W);
    const int hs = max(h0, 0), ws = max(w0, 0);
    float acc = g.is_max ? -3.4e38f : 0.f;
    for (int h = hs; h < h1; ++h)
      for (int w = ws; w < w1; ++w) {
        const float v = to_f<T>(src[h * g.W + w]);
        acc = g.is_max ? fmaxf(acc, v) : acc + v;
      }
    if (!g.is_max) {
      const int cnt = (h1 - hs) * (w1 - ws);
      acc = cnt > 0 ? acc / (float)cnt : 0.f;
    }
    y[i] = from_f<T>(acc);
  }
}

template <typename T>
__global__ void pool_bwd_kernel(PoolGeom g, const T *__restrict__ dy,
                                const T *__restrict__ x,
                                const T *__restrict__ y, T *__restrict__ dx,
                                int64_t total) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    const int w = i % g.W;
    const int h = (i / g.W) % g.H;
    const int64_t nc = i / ((int64_t)g.H * g.W);
    const T *dyp = dy + nc * g.P * g.Q;
    const T *yp = y + nc * g.P * g.Q;
    // output windows covering (h, w): p*sh - ph <= h < p*sh - ph + kh
    const int hp = h + g.ph, wp = w + g.pw;
    const int p_lo = hp < g.kh ? 0 : (hp - g.kh) / g.sh + 1;
    const int p_hi = min(hp / g.sh, g.P - 1);
    const int q_lo = wp < g.kw ? 0 : (wp - g.kw) / g.sw + 1;
    const int q_hi = min(wp / g.sw, g.Q - 1);
    const float xv = to_f<T>(x[i]);
    float acc = 0.f;
    for (int p = p_lo; p <= p_hi; ++p)
      for (int q = q_lo; q <= q_hi; ++q) {
        const float d = to_f<T>(dyp[p * g.Q + q]);
        if (g.is_max) {
          if (to_f<T>(yp[p * g.Q + q]) == xv) acc += d;
        } else {
          const int h0 = p * g.sh - g.ph, w0 = q * g.sw - g.pw;
          const int cnt = (min(h0 + g.kh, g.H) - max(h0, 0)) *
                          (min(w0 + g.kw, g.W) - max(w0, 0));
          acc += d / (float)cnt;
        }
      }
    dx[i] = from_f<T>(acc);
  }
}

// Max-pooling gradient with one (n, c) plane per blockIdx.y and four
// consecutive columns per thread (W a multiple of 4): 32-bit index arithmetic,
// 8- / 16-byte loads of x and stores of dx, the few rows of dy / y a thread
// needs stay in L1 across its four columns.
// KS = window edge and stride as compile-time constants (KS = 32 for the
// 3 x 3 / stride 2 windows of the residual networks: the window bounds then
// need shifts instead of integer divisions); 0 = taken from g.
template <typename T, int KS>
__global__ void __launch_bounds__(256)
maxpool_bwd_plane_kernel(PoolGeom gg, const T *__restrict__ dy,
                         const T *__restrict__ x, const T *__restrict__ y,
                         T *__restrict__ dx, int planes) {
  struct Win {
    int kh, kw, sh, sw;
  };
  const PoolGeom &g = gg;
  const Win win = KS == 32 ? Win{3, 3, 2, 2} : Win{g.kh, g.kw, g.sh, g.sw};
  const int quads = g.W >> 2, per_plane = g.H * quads;
  for (int nc = blockIdx.y; nc < planes; nc += gridDim.y) {
    const T *dyp = dy + (int64_t)nc * g.P * g.Q;
    const T *yp = y + (int64_t)nc * g.P * g.Q;
    const T *xp = x + (int64_t)nc * g.H * g.W;
    T *dxp = dx + (int64_t)nc * g.H * g.W;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < per_plane;
         e += gridDim.x * blockDim.x) {
      const int h = e / quads, w0 = (e - h * quads) << 2;
      T xe[4], oe[4];
      if (sizeof(T) == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(xp + h * g.W + w0);
        __builtin_memcpy(xe, &v, 16);
      } else {
        const uint2 v = *reinterpret_cast<const uint2 *>(xp + h * g.W + w0);
        __builtin_memcpy(xe, &v, 8);
      }
      const int hp = h + g.ph;
      const int p_lo = hp < win.kh ? 0 : (hp - win.kh) / win.sh + 1;
      const int p_hi = min(hp / win.sh, g.P - 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int wp = w0 + k + g.pw;
        const int q_lo = wp < win.kw ? 0 : (wp - win.kw) / win.sw + 1;
        const int q_hi = min(wp / win.sw, g.Q - 1);
        const float xv = to_f<T>(xe[k]);
        float acc = 0.f;
        for (int p = p_lo; p <= p_hi; ++p)
          for (int q = q_lo; q <= q_hi; ++q)
            if (to_f<T>(yp[p * g.Q + q]) == xv)
              acc += to_f<T>(dyp[p * g.Q + q]);
        oe[k] = from_f<T>(acc);
      }
      if (sizeof(T) == 4) {
        float4 v;
        __builtin_memcpy(&v, oe, 16);
        *reinterpret_cast<float4 *>(dxp + h * g.W + w0) = v;
      } else {
        uint2 v;
        __builtin_memcpy(&v, oe, 8);
        *reinterpret_cast<uint2 *>(dxp + h * g.W + w0) = v;
      }
    }
  }
}

}  // namespace sa

using namespace sa;

int sa_pool_fwd(int dtype, const PoolGeom *g, const void *x, void *y,
                hipStream_t s) {
  const int64_t total = (int64_t)g->N * g->C * g->P * g->Q;
  if (total <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((pool_fwd_kernel<T>), dim3(grid_for(total, 256)),
                       dim3(256), 0, s, *g, (const T *)x, (T *)y, total);
  });
  return 0;
}

int sa_pool_bwd(int dtype, const PoolGeom *g, const void *dy, const void *x,
                const void *y, void *dx, hipStream_t s) {
  const int64_t total = (int64_t)g->N * g->C * g->H * g->W;
  if (total <= 0) return 0;
  const int64_t planes = (int64_t)g->N * g->C;
  const bool plane_path =
      g->is_max && g->W % 4 == 0 && (int64_t)g->H * g->W < (1 << 28) &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) &
       15) == 0;
  if (plane_path) {
    const int per_plane = g->H * (g->W / 4);
    const dim3 grid((unsigned)((per_plane + 255) / 256),
                    (unsigned)i64min(planes, 32768));
    const bool k3s2 = g->kh == 3 && g->kw == 3 && g->sh == 2 && g->sw == 2;
    SA_DISPATCH_FLOAT(dtype, T, {
      if (k3s2)
        hipLaunchKernelGGL((maxpool_bwd_plane_kernel<T, 32>), grid, dim3(256),
                           0, s, *g, (const T *)dy, (const T *)x,
                           (const T *)y, (T *)dx, (int)planes);
      else
        hipLaunchKernelGGL((maxpool_bwd_plane_kernel<T, 0>), grid, dim3(256),
                           0, s, *g, (const T *)dy, (const T *)x,
                           (const T *)y, (T *)dx, (int)planes);
    });
    return 0;
  }
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((pool_bwd_kernel<T>), dim3(grid_for(total, 256)),
                       dim3(256), 0, s, *g, (const T *)dy, (const T *)x,
                       (const T *)y, (T *)dx, total);
  });
  return 0;
}
