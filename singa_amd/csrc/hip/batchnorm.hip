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

// enough (channel, image-group) blocks to keep ~8 MB of loads in flight
static inline int bn_splits(int N, int C) {
  int s = 8192 / (C > 0 ? C : 1);
  if (s < 1) s = 1;
  if (s > N) s = N;
  return s;
}

// Four consecutive elements as floats: one 16-byte access for float tensors,
// one 8-byte access for the 16-bit types.
template <typename T> struct V4 {
  static __device__ __forceinline__ float4 ld(const T *p, int i) {
    const uint2 w = reinterpret_cast<const uint2 *>(p)[i];
    T e[4];
    __builtin_memcpy(e, &w, 8);
    return make_float4(to_f<T>(e[0]), to_f<T>(e[1]), to_f<T>(e[2]),
                       to_f<T>(e[3]));
  }
  static __device__ __forceinline__ void st(T *p, int i, float4 v) {
    T e[4] = {from_f<T>(v.x), from_f<T>(v.y), from_f<T>(v.z), from_f<T>(v.w)};
    uint2 w;
    __builtin_memcpy(&w, e, 8);
    reinterpret_cast<uint2 *>(p)[i] = w;
  }
};
template <> struct V4<float> {
  static __device__ __forceinline__ float4 ld(const float *p, int i) {
    return reinterpret_cast<const float4 *>(p)[i];
  }
  static __device__ __forceinline__ void st(float *p, int i, float4 v) {
    reinterpret_cast<float4 *>(p)[i] = v;
  }
};

// work layout: [C][S][2]
//
// VEC = 4 variants move four elements per lane and access (channel rows are
// 16-byte aligned for float / 8-byte aligned for 16-bit types when
// HW % 4 == 0); VEC = 1 is the general path.  ACT: 0 none, 1 ReLU,
// 2 add a residual tensor then ReLU (the tail of a ResNet bottleneck).
template <typename T, int VEC>
__global__ void bn_stats_kernel(const T *__restrict__ x,
                                float *__restrict__ work, int N, int C, int HW,
                                int S) {
  __shared__ float smem[32];
  const int c = blockIdx.x, s = blockIdx.y;
  const float shift = to_f<T>(x[(size_t)c * HW]);
  float sum = 0.f, sq = 0.f;
  for (int n = s; n < N; n += S) {
    const T *p = x + ((size_t)n * C + c) * HW;
    if constexpr (VEC == 4) {
      for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
        const float4 v = V4<T>::ld(p, i);
        const float a = v.x - shift, b = v.y - shift, cc = v.z - shift,
                    d = v.w - shift;
        sum += (a + b) + (cc + d);
        sq += (a * a + b * b) + (cc * cc + d * d);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float v = to_f<T>(p[i]) - shift;
        sum += v;
        sq += v * v;
      }
    }
  }
  sum = block_sum(sum, smem);
  sq = block_sum(sq, smem);
  if (threadIdx.x == 0) {
    work[((size_t)c * S + s) * 2 + 0] = sum;
    work[((size_t)c * S + s) * 2 + 1] = sq;
  }
}

template <int ACT>
__device__ __forceinline__ float bn_act(float z, float r) {
  if (ACT == 2) z += r;
  if (ACT >= 1) z = fmaxf(z, 0.f);
  return z;
}

// SUMS: the statistics were taken by the producer of x (a convolution's
// epilogue, ConvGeom::stats): `work` is float[kStatCopies][2][C] holding
// sums of x - rmean[c] (the running mean as it is BEFORE this step's update),
// followed by one unsigned counter per channel.  Every block reads the sums
// and the shift, then counts itself in; the last block of a channel to do so
// writes the saved and running statistics and clears the channel's sums and
// counter for the next step (nobody reads them any more at that point).
template <typename T, int VEC, int ACT, bool SUMS = false>
__global__ void bn_apply_train_kernel(
    const T *__restrict__ x, const float *__restrict__ scale,
    const float *__restrict__ bias, float *__restrict__ rmean,
    float *__restrict__ rvar, float factor, float eps, T *__restrict__ y,
    float *__restrict__ mean_out, float *__restrict__ invstd_out,
    float *__restrict__ work, const T *__restrict__ res, int N, int C,
    int HW, int S) {
  const int c = blockIdx.x, s = blockIdx.y;
  float sum = 0.f, sq = 0.f, shift;
  if constexpr (SUMS) {
    __shared__ float got[3];
    if (threadIdx.x == 0) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int k = 0; k < kStatCopies; ++k) {
        a += work[((size_t)k * 2 + 0) * C + c];
        b += work[((size_t)k * 2 + 1) * C + c];
      }
      got[0] = a;
      got[1] = b;
      got[2] = rmean[c];
    }
    __syncthreads();
    sum = got[0];
    sq = got[1];
    shift = got[2];
  } else {
    for (int i = 0; i < S; ++i) {
      sum += work[((size_t)c * S + i) * 2 + 0];
      sq += work[((size_t)c * S + i) * 2 + 1];
    }
    shift = to_f<T>(x[(size_t)c * HW]);
  }
  const float cnt = (float)N * (float)HW;
  const float dm = sum / cnt;
  const float mean = shift + dm;
  float var = sq / cnt - dm * dm;
  var = fmaxf(var, 0.f);
  const float invstd = rsqrtf(var + eps);
  bool writer = s == 0 && threadIdx.x == 0;
  if constexpr (SUMS) {
    writer = false;
    if (threadIdx.x == 0) {
      unsigned *counter =
          reinterpret_cast<unsigned *>(work + (size_t)kStatCopies * 2 * C);
      __threadfence();
      writer = atomicAdd(counter + c, 1u) == gridDim.y - 1;
      if (writer) {
#pragma unroll
        for (int k = 0; k < kStatCopies; ++k) {
          work[((size_t)k * 2 + 0) * C + c] = 0.f;
          work[((size_t)k * 2 + 1) * C + c] = 0.f;
        }
        counter[c] = 0u;
      }
    }
  }
  if (writer) {
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    const float unbiased = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
    rmean[c] = (1.f - factor) * rmean[c] + factor * mean;
    rvar[c] = (1.f - factor) * rvar[c] + factor * unbiased;
  }
  const float g = scale[c] * invstd;
  const float b = bias[c] - mean * g;
  for (int n = s; n < N; n += S) {
    const size_t off = ((size_t)n * C + c) * HW;
    if constexpr (VEC == 4) {
      for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
        const float4 v = V4<T>::ld(x + off, i);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ACT == 2) r = V4<T>::ld(res + off, i);
        float4 o;
        o.x = bn_act<ACT>(fmaf(v.x, g, b), r.x);
        o.y = bn_act<ACT>(fmaf(v.y, g, b), r.y);
        o.z = bn_act<ACT>(fmaf(v.z, g, b), r.z);
        o.w = bn_act<ACT>(fmaf(v.w, g, b), r.w);
        V4<T>::st(y + off, i, o);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float r = ACT == 2 ? to_f<T>(res[off + i]) : 0.f;
        y[off + i] =
            from_f<T>(bn_act<ACT>(fmaf(to_f<T>(x[off + i]), g, b), r));
      }
    }
  }
}

template <typename T>
__global__ void bn_infer_kernel(const T *__restrict__ x,
                                const float *__restrict__ scale,
                                const float *__restrict__ bias,
                                const float *__restrict__ rmean,
                                const float *__restrict__ rvar, float eps,
                                T *__restrict__ y, int64_t total, int C,
                                Div32 div_hw, Div32 div_c, int act,
                                const T *__restrict__ res) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    uint32_t nc, hw, n, c;
    div_hw.divmod((uint32_t)i, nc, hw);
    div_c.divmod(nc, n, c);
    const float g = scale[c] * rsqrtf(rvar[c] + eps);
    float z = (to_f<T>(x[i]) - rmean[c]) * g + bias[c];
    if (act == 2) z += to_f<T>(res[i]);
    if (act >= 1) z = fmaxf(z, 0.f);
    y[i] = from_f<T>(z);
  }
}

// VEC consecutive elements per thread (HW % VEC == 0, so they share a channel
// and the per-channel factors are formed once per vector); one load and one
// store of VEC * sizeof(T) bytes per tensor.
template <typename T, int VEC>
__global__ void bn_infer_vec_kernel(const T *__restrict__ x,
                                    const float *__restrict__ scale,
                                    const float *__restrict__ bias,
                                    const float *__restrict__ rmean,
                                    const float *__restrict__ rvar, float eps,
                                    T *__restrict__ y, int64_t units, int C,
                                    Div32 div_per, Div32 div_c, int act,
                                    const T *__restrict__ res) {
  struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < units;
       u += step) {
    uint32_t nc, in_plane, n, c;
    div_per.divmod((uint32_t)u, nc, in_plane);
    div_c.divmod(nc, n, c);
    const float g = scale[c] * rsqrtf(rvar[c] + eps);
    const float b = fmaf(-rmean[c], g, bias[c]);
    const Pack xv = reinterpret_cast<const Pack *>(x)[u];
    Pack rv, out;
    if (act == 2) rv = reinterpret_cast<const Pack *>(res)[u];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float z = fmaf(to_f<T>(xv.v[j]), g, b);
      if (act == 2) z += to_f<T>(rv.v[j]);
      if (act >= 1) z = fmaxf(z, 0.f);
      out.v[j] = from_f<T>(z);
    }
    reinterpret_cast<Pack *>(y)[u] = out;
  }
}

// pass 1 of backward: sum(dy), sum(dy * xhat).  With an activation the
// incoming gradient is first masked by (output > 0): for ACT 1 the output sign
// is recomputed from x (no extra tensor is read), for ACT 2 the saved output y
// is read.
template <int ACT>
__device__ __forceinline__ float bn_mask(float d, float xv, float g, float b,
                                         float yv) {
  if (ACT == 1) return fmaf(xv, g, b) > 0.f ? d : 0.f;
  if (ACT == 2) return yv > 0.f ? d : 0.f;
  return d;
}

// With ACT 2 the masked gradient is also the gradient of the residual
// branch: this pass writes it to `dres`, and the second pass reads it back
// instead of reading dy and y again (7 passes over the tensors, not 8).
template <typename T, int VEC, int ACT>
__global__ void bn_bwd_stats_kernel(const T *__restrict__ dy,
                                    const T *__restrict__ x,
                                    const T *__restrict__ y,
                                    const float *__restrict__ scale,
                                    const float *__restrict__ bias,
                                    const float *__restrict__ mean,
                                    const float *__restrict__ invstd,
                                    float *__restrict__ work,
                                    T *__restrict__ dres, int N, int C,
                                    int HW, int S) {
  __shared__ float smem[32];
  const int c = blockIdx.x, s = blockIdx.y;
  const float m = mean[c], iv = invstd[c];
  const float g = ACT == 1 ? scale[c] * iv : 0.f;
  const float b = ACT == 1 ? bias[c] - m * g : 0.f;
  float sdy = 0.f, sdyx = 0.f;
  for (int n = s; n < N; n += S) {
    const size_t off = ((size_t)n * C + c) * HW;
    if constexpr (VEC == 4) {
      for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
        const float4 d = V4<T>::ld(dy + off, i), xv = V4<T>::ld(x + off, i);
        float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ACT == 2) yv = V4<T>::ld(y + off, i);
        const float d0 = bn_mask<ACT>(d.x, xv.x, g, b, yv.x);
        const float d1 = bn_mask<ACT>(d.y, xv.y, g, b, yv.y);
        const float d2 = bn_mask<ACT>(d.z, xv.z, g, b, yv.z);
        const float d3 = bn_mask<ACT>(d.w, xv.w, g, b, yv.w);
        if (ACT == 2) V4<T>::st(dres + off, i, make_float4(d0, d1, d2, d3));
        sdy += (d0 + d1) + (d2 + d3);
        sdyx += (d0 * (xv.x - m) + d1 * (xv.y - m)) +
                (d2 * (xv.z - m) + d3 * (xv.w - m));
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float xv = to_f<T>(x[off + i]);
        const float yv = ACT == 2 ? to_f<T>(y[off + i]) : 0.f;
        const float d = bn_mask<ACT>(to_f<T>(dy[off + i]), xv, g, b, yv);
        if (ACT == 2) dres[off + i] = from_f<T>(d);
        sdy += d;
        sdyx += d * (xv - m);
      }
    }
  }
  sdy = block_sum(sdy, smem);
  sdyx = block_sum(sdyx, smem) * iv;
  if (threadIdx.x == 0) {
    work[((size_t)c * S + s) * 2 + 0] = sdy;
    work[((size_t)c * S + s) * 2 + 1] = sdyx;
  }
}

// pass 2: dx; for ACT 2 also writes the masked gradient `dres` that flows
// into the residual branch
template <typename T, int VEC, int ACT>
__global__ void bn_bwd_apply_kernel(
    const T *__restrict__ dy, const T *__restrict__ x, const T *__restrict__ y,
    const float *__restrict__ scale, const float *__restrict__ bias,
    const float *__restrict__ mean, const float *__restrict__ invstd,
    T *__restrict__ dx, T *__restrict__ dres, float *__restrict__ dscale,
    float *__restrict__ dbias, const float *__restrict__ work, int N, int C,
    int HW, int S) {
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
  const float b = ACT == 1 ? bias[c] - m * g : 0.f;
  const float k1 = sdy / cnt, k2 = sdyx / cnt * iv;
  // dx = g * (d - k1 - (x - m) * k2)  with k2 already carrying one 1/std
  for (int n = s; n < N; n += S) {
    const size_t off = ((size_t)n * C + c) * HW;
    // ACT 2: the masked gradient was written to dres by the first pass
    if constexpr (VEC == 4) {
      for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
        const float4 d = V4<T>::ld((ACT == 2 ? dres : dy) + off, i),
                     xv = V4<T>::ld(x + off, i);
        float4 dm, o;
        dm.x = ACT == 2 ? d.x : bn_mask<ACT>(d.x, xv.x, g, b, 0.f);
        dm.y = ACT == 2 ? d.y : bn_mask<ACT>(d.y, xv.y, g, b, 0.f);
        dm.z = ACT == 2 ? d.z : bn_mask<ACT>(d.z, xv.z, g, b, 0.f);
        dm.w = ACT == 2 ? d.w : bn_mask<ACT>(d.w, xv.w, g, b, 0.f);
        o.x = g * (dm.x - k1 - (xv.x - m) * k2);
        o.y = g * (dm.y - k1 - (xv.y - m) * k2);
        o.z = g * (dm.z - k1 - (xv.z - m) * k2);
        o.w = g * (dm.w - k1 - (xv.w - m) * k2);
        V4<T>::st(dx + off, i, o);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float xv = to_f<T>(x[off + i]);
        const float d0 = to_f<T>((ACT == 2 ? dres : dy)[off + i]);
        const float d = ACT == 2 ? d0 : bn_mask<ACT>(d0, xv, g, b, 0.f);
        dx[off + i] = from_f<T>(g * (d - k1 - (xv - m) * k2));
      }
    }
  }
}

// ---------------------------------------------------------------------------
// One workgroup per channel: the channel lives in registers
// ---------------------------------------------------------------------------
// When a channel of the batch (N * HW values) fits the registers of one
// workgroup -- the 28 x 28, 14 x 14 and 7 x 7 layers of a ResNet at batch 32 --
// the two passes above become one kernel that reads every tensor ONCE: each
// thread keeps up to UNITS pieces (VEC values each) of the channel, the block
// reduces the statistics, and the normalisation (or its gradient) is applied
// to the values still held in registers.  Half the launches and one pass over
// the tensor less than the split-channel kernels, which remain for channels
// too large for one workgroup.
constexpr int kBnUnits = 7;

template <typename T, int VEC> struct BnPiece {
  float v[VEC];
  __device__ __forceinline__ void ld(const T *p, size_t off, int i) {
    if constexpr (VEC == 4) {
      const float4 q = V4<T>::ld(p + off, i);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      v[0] = to_f<T>(p[off + i]);
    }
  }
  __device__ __forceinline__ void st(T *p, size_t off, int i) const {
    if constexpr (VEC == 4)
      V4<T>::st(p + off, i, make_float4(v[0], v[1], v[2], v[3]));
    else
      p[off + i] = from_f<T>(v[0]);
  }
};

// unit u of the channel: image u / per_img, piece u % per_img of that image
struct BnUnit {
  size_t off;
  int i;
};
__device__ __forceinline__ BnUnit bn_unit(int u, int c, int C, int HW,
                                          Div32 div_per) {
  uint32_t n, i;
  div_per.divmod((uint32_t)u, n, i);
  return BnUnit{((size_t)n * C + c) * HW, (int)i};
}

template <typename T, int VEC, int ACT>
__global__ void __launch_bounds__(1024)
bn_fwd_channel_kernel(const T *__restrict__ x, const float *__restrict__ scale,
                      const float *__restrict__ bias, float *__restrict__ rmean,
                      float *__restrict__ rvar, float factor, float eps,
                      T *__restrict__ y, float *__restrict__ mean_out,
                      float *__restrict__ invstd_out, const T *__restrict__ res,
                      int N, int C, int HW, int units, Div32 div_per) {
  __shared__ float smem[32];
  const int c = blockIdx.x;
  const float shift = to_f<T>(x[(size_t)c * HW]);
  BnPiece<T, VEC> xv[kBnUnits];
  float sum = 0.f, sq = 0.f;
#pragma unroll
  for (int k = 0; k < kBnUnits; ++k) {
    const int u = threadIdx.x + k * blockDim.x;
    if (u < units) {
      const BnUnit at = bn_unit(u, c, C, HW, div_per);
      xv[k].ld(x, at.off, at.i);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float d = xv[k].v[e] - shift;
        sum += d;
        sq += d * d;
      }
    }
  }
  sum = block_sum(sum, smem);
  sq = block_sum(sq, smem);
  const float cnt = (float)N * (float)HW;
  const float dm = sum / cnt;
  const float mean = shift + dm;
  const float var = fmaxf(sq / cnt - dm * dm, 0.f);
  const float invstd = rsqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    const float unbiased = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
    rmean[c] = (1.f - factor) * rmean[c] + factor * mean;
    rvar[c] = (1.f - factor) * rvar[c] + factor * unbiased;
  }
  const float g = scale[c] * invstd;
  const float b = bias[c] - mean * g;
#pragma unroll
  for (int k = 0; k < kBnUnits; ++k) {
    const int u = threadIdx.x + k * blockDim.x;
    if (u < units) {
      const BnUnit at = bn_unit(u, c, C, HW, div_per);
      BnPiece<T, VEC> r, o;
      if (ACT == 2) r.ld(res, at.off, at.i);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        o.v[e] = bn_act<ACT>(fmaf(xv[k].v[e], g, b), ACT == 2 ? r.v[e] : 0.f);
      o.st(y, at.off, at.i);
    }
  }
}

template <typename T, int VEC, int ACT>
__global__ void __launch_bounds__(1024)
bn_bwd_channel_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                      const T *__restrict__ y, const float *__restrict__ scale,
                      const float *__restrict__ bias,
                      const float *__restrict__ mean,
                      const float *__restrict__ invstd, T *__restrict__ dx,
                      T *__restrict__ dres, float *__restrict__ dscale,
                      float *__restrict__ dbias, int N, int C, int HW,
                      int units, Div32 div_per) {
  __shared__ float smem[32];
  const int c = blockIdx.x;
  const float m = mean[c], iv = invstd[c];
  const float g = scale[c] * iv;
  const float b = ACT == 1 ? bias[c] - m * g : 0.f;
  BnPiece<T, VEC> dv[kBnUnits], xv[kBnUnits];
  float sdy = 0.f, sdyx = 0.f;
#pragma unroll
  for (int k = 0; k < kBnUnits; ++k) {
    const int u = threadIdx.x + k * blockDim.x;
    if (u < units) {
      const BnUnit at = bn_unit(u, c, C, HW, div_per);
      dv[k].ld(dy, at.off, at.i);
      xv[k].ld(x, at.off, at.i);
      BnPiece<T, VEC> yv;
      if (ACT == 2) yv.ld(y, at.off, at.i);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        // the masked gradient replaces dy in the registers
        const float d = bn_mask<ACT>(dv[k].v[e], xv[k].v[e], g, b,
                                     ACT == 2 ? yv.v[e] : 0.f);
        dv[k].v[e] = d;
        sdy += d;
        sdyx += d * (xv[k].v[e] - m);
      }
    }
  }
  sdy = block_sum(sdy, smem);
  sdyx = block_sum(sdyx, smem) * iv;
  if (threadIdx.x == 0) {
    dscale[c] = sdyx;
    dbias[c] = sdy;
  }
  const float cnt = (float)N * (float)HW;
  const float k1 = sdy / cnt, k2 = sdyx / cnt * iv;
#pragma unroll
  for (int k = 0; k < kBnUnits; ++k) {
    const int u = threadIdx.x + k * blockDim.x;
    if (u < units) {
      const BnUnit at = bn_unit(u, c, C, HW, div_per);
      BnPiece<T, VEC> o;
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        o.v[e] = g * (dv[k].v[e] - k1 - (xv[k].v[e] - m) * k2);
      o.st(dx, at.off, at.i);
      if (ACT == 2) dv[k].st(dres, at.off, at.i);
    }
  }
}


// ---------------------------------------------------------------------------
// One workgroup per channel, two sweeps over the channel
// ---------------------------------------------------------------------------
// For channels too large for the registers of a workgroup (the 112 x 112 and
// 56 x 56 layers at batch 32): the workgroup reads its channel once for the
// statistics and a second time to apply them.  The second sweep starts where
// the first one started, a few hundred microseconds earlier, so its reads
// are served by the Infinity Cache (256 channels x 400 KB per tensor stay
// well inside its 256 MiB) instead of by a second pass over HBM as with two
// kernels, and the statistics never leave the workgroup.
template <typename T, int VEC, int ACT>
__global__ void __launch_bounds__(1024)
bn_fwd_sweep_kernel(const T *__restrict__ x, const float *__restrict__ scale,
                    const float *__restrict__ bias, float *__restrict__ rmean,
                    float *__restrict__ rvar, float factor, float eps,
                    T *__restrict__ y, float *__restrict__ mean_out,
                    float *__restrict__ invstd_out, const T *__restrict__ res,
                    int N, int C, int HW, int units, Div32 div_per) {
  __shared__ float smem[32];
  const int c = blockIdx.x;
  const float shift = to_f<T>(x[(size_t)c * HW]);
  float sum = 0.f, sq = 0.f;
  for (int u = threadIdx.x; u < units; u += blockDim.x) {
    const BnUnit at = bn_unit(u, c, C, HW, div_per);
    BnPiece<T, VEC> xv;
    xv.ld(x, at.off, at.i);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = xv.v[e] - shift;
      sum += d;
      sq += d * d;
    }
  }
  sum = block_sum(sum, smem);
  sq = block_sum(sq, smem);
  const float cnt = (float)N * (float)HW;
  const float dm = sum / cnt;
  const float mean = shift + dm;
  const float var = fmaxf(sq / cnt - dm * dm, 0.f);
  const float invstd = rsqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    const float unbiased = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
    rmean[c] = (1.f - factor) * rmean[c] + factor * mean;
    rvar[c] = (1.f - factor) * rvar[c] + factor * unbiased;
  }
  const float g = scale[c] * invstd;
  const float b = bias[c] - mean * g;
  for (int u = threadIdx.x; u < units; u += blockDim.x) {
    const BnUnit at = bn_unit(u, c, C, HW, div_per);
    BnPiece<T, VEC> xv, r, o;
    xv.ld(x, at.off, at.i);
    if (ACT == 2) r.ld(res, at.off, at.i);
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      o.v[e] = bn_act<ACT>(fmaf(xv.v[e], g, b), ACT == 2 ? r.v[e] : 0.f);
    o.st(y, at.off, at.i);
  }
}

template <typename T, int VEC, int ACT>
__global__ void __launch_bounds__(1024)
bn_bwd_sweep_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                    const T *__restrict__ y, const float *__restrict__ scale,
                    const float *__restrict__ bias,
                    const float *__restrict__ mean,
                    const float *__restrict__ invstd, T *__restrict__ dx,
                    T *__restrict__ dres, float *__restrict__ dscale,
                    float *__restrict__ dbias, int N, int C, int HW, int units,
                    Div32 div_per) {
  __shared__ float smem[32];
  const int c = blockIdx.x;
  const float m = mean[c], iv = invstd[c];
  const float g = scale[c] * iv;
  const float b = ACT == 1 ? bias[c] - m * g : 0.f;
  float sdy = 0.f, sdyx = 0.f;
  for (int u = threadIdx.x; u < units; u += blockDim.x) {
    const BnUnit at = bn_unit(u, c, C, HW, div_per);
    BnPiece<T, VEC> dv, xv, yv;
    dv.ld(dy, at.off, at.i);
    xv.ld(x, at.off, at.i);
    if (ACT == 2) yv.ld(y, at.off, at.i);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = bn_mask<ACT>(dv.v[e], xv.v[e], g, b,
                                   ACT == 2 ? yv.v[e] : 0.f);
      dv.v[e] = d;
      sdy += d;
      sdyx += d * (xv.v[e] - m);
    }
    // the masked gradient is the residual branch's gradient; the second
    // sweep (same thread, same units) reads it back instead of dy and y
    if (ACT == 2) dv.st(dres, at.off, at.i);
  }
  sdy = block_sum(sdy, smem);
  sdyx = block_sum(sdyx, smem) * iv;
  if (threadIdx.x == 0) {
    dscale[c] = sdyx;
    dbias[c] = sdy;
  }
  const float cnt = (float)N * (float)HW;
  const float k1 = sdy / cnt, k2 = sdyx / cnt * iv;
  for (int u = threadIdx.x; u < units; u += blockDim.x) {
    const BnUnit at = bn_unit(u, c, C, HW, div_per);
    BnPiece<T, VEC> dv, xv, o;
    dv.ld(ACT == 2 ? dres : dy, at.off, at.i);
    xv.ld(x, at.off, at.i);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = ACT == 2 ? dv.v[e]
                               : bn_mask<ACT>(dv.v[e], xv.v[e], g, b, 0.f);
      o.v[e] = g * (d - k1 - (xv.v[e] - m) * k2);
    }
    o.st(dx, at.off, at.i);
  }
}

}  // namespace sa

using namespace sa;

int64_t sa_bn_workspace(int N, int C, int HW) {
  return (int64_t)C * bn_splits(N, C) * 2;
}

static inline int bn_block(int HW) { return HW >= 256 ? 256 : (HW > 64 ? 128 : 64); }

// tensors whose channel rows are aligned to four elements take the VEC 4 path
static inline bool bn_vec4(int dtype, int HW, const void *a, const void *b,
                           const void *c, const void *d) {
  uintptr_t bits = reinterpret_cast<uintptr_t>(a) |
                   reinterpret_cast<uintptr_t>(b) |
                   reinterpret_cast<uintptr_t>(c) |
                   reinterpret_cast<uintptr_t>(d);
  return HW % 4 == 0 && (bits & (dtype == kF32 ? 15 : 7)) == 0;
}


// Threads of the one-workgroup-per-channel kernels for `units` pieces per
// channel, 0 when a channel does not fit (or the switch SA_BN_CHANNEL=0 is
// set).
static int bn_channel_threads(int64_t units) {
  static const int enabled = [] {
    const char *v = getenv("SA_BN_CHANNEL");
    return v && *v ? atoi(v) : 1;
  }();
  if (!enabled || units > (int64_t)1024 * kBnUnits) return 0;
  int t = (int)((units + kBnUnits - 1) / kBnUnits);
  t = (t + 63) / 64 * 64;
  return t < 64 ? 64 : t;
}

// Channels that do not fit a workgroup's registers are swept twice by one
// workgroup each when there are at least SA_BN_SWEEP channels (default 256:
// one workgroup per CU; measured 10 % faster than the statistics + apply pair
// on the 256-channel 56 x 56 layers, slower below 128 channels), 0 = never.
static bool bn_sweep(int C) {
  static const int min_c = [] {
    const char *v = getenv("SA_BN_SWEEP");
    return v && *v ? atoi(v) : 256;
  }();
  return min_c > 0 && C >= min_c;
}

#define BN_ACT_SWITCH(act, ...)                         \
  switch (act) {                                        \
    case 0: { constexpr int ACT = 0; __VA_ARGS__; break; } \
    case 1: { constexpr int ACT = 1; __VA_ARGS__; break; } \
    case 2: { constexpr int ACT = 2; __VA_ARGS__; break; } \
    default: return -4;                                 \
  }

int sa_bn_fwd_train_act(int dtype, const void *x, const float *scale,
                        const float *bias, float *rmean, float *rvar,
                        float factor, float eps, void *y, float *mean,
                        float *invstd, int N, int C, int HW, float *work,
                        int act, const void *residual, hipStream_t s) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  if (act == 2 && residual == nullptr) return -5;
  const int S = bn_splits(N, C);
  {
    const bool v4 = bn_vec4(dtype, HW, x, y, residual, nullptr);
    const int per = v4 ? HW / 4 : HW;
    const int64_t units = (int64_t)N * per;
    const int threads = bn_channel_threads(units);
    if (threads) {
      SA_DISPATCH_FLOAT(dtype, T, {
        BN_ACT_SWITCH(act, {
          if (v4)
            hipLaunchKernelGGL((bn_fwd_channel_kernel<T, 4, ACT>), dim3(C),
                               dim3(threads), 0, s, (const T *)x, scale, bias,
                               rmean, rvar, factor, eps, (T *)y, mean, invstd,
                               (const T *)residual, N, C, HW, (int)units,
                               Div32(per));
          else
            hipLaunchKernelGGL((bn_fwd_channel_kernel<T, 1, ACT>), dim3(C),
                               dim3(threads), 0, s, (const T *)x, scale, bias,
                               rmean, rvar, factor, eps, (T *)y, mean, invstd,
                               (const T *)residual, N, C, HW, (int)units,
                               Div32(per));
        });
      });
      return 0;
    }
    if (v4 && bn_sweep(C) && units < ((int64_t)1 << 31)) {
      SA_DISPATCH_FLOAT(dtype, T, {
        BN_ACT_SWITCH(act, {
          hipLaunchKernelGGL((bn_fwd_sweep_kernel<T, 4, ACT>), dim3(C),
                             dim3(1024), 0, s, (const T *)x, scale, bias,
                             rmean, rvar, factor, eps, (T *)y, mean, invstd,
                             (const T *)residual, N, C, HW, (int)units,
                             Div32(per));
        });
      });
      return 0;
    }
  }
  if (bn_vec4(dtype, HW, x, y, residual, nullptr)) {
    const dim3 grid(C, S), block(bn_block(HW / 4));
    SA_DISPATCH_FLOAT(dtype, T, {
      hipLaunchKernelGGL((bn_stats_kernel<T, 4>), grid, block, 0, s,
                         (const T *)x, work, N, C, HW, S);
      BN_ACT_SWITCH(act, {
        hipLaunchKernelGGL((bn_apply_train_kernel<T, 4, ACT>), grid, block, 0,
                           s, (const T *)x, scale, bias, rmean, rvar, factor,
                           eps, (T *)y, mean, invstd, work,
                           (const T *)residual, N, C, HW, S);
      });
    });
    return 0;
  }
  const dim3 grid(C, S), block(bn_block(HW));
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((bn_stats_kernel<T, 1>), grid, block, 0, s,
                       (const T *)x, work, N, C, HW, S);
    BN_ACT_SWITCH(act, {
      hipLaunchKernelGGL((bn_apply_train_kernel<T, 1, ACT>), grid, block, 0,
                         s, (const T *)x, scale, bias, rmean, rvar, factor,
                         eps, (T *)y, mean, invstd, work, (const T *)residual,
                         N, C, HW, S);
    });
  });
  return 0;
}

// Training forward from statistics taken by the producer of x: one pass.
// stats: float[kStatCopies][2][C] + unsigned[C], see bn_apply_train_kernel.
int sa_bn_fwd_train_sums(int dtype, const void *x, const float *scale,
                         const float *bias, float *rmean, float *rvar,
                         float factor, float eps, void *y, float *mean,
                         float *invstd, int N, int C, int HW, float *stats,
                         int act, const void *residual, hipStream_t s) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  if (act == 2 && residual == nullptr) return -5;
  if (stats == nullptr) return -5;
  const int S = bn_splits(N, C);
  if (bn_vec4(dtype, HW, x, y, residual, nullptr)) {
    const dim3 grid(C, S), block(bn_block(HW / 4));
    SA_DISPATCH_FLOAT(dtype, T, {
      BN_ACT_SWITCH(act, {
        hipLaunchKernelGGL((bn_apply_train_kernel<T, 4, ACT, true>), grid,
                           block, 0, s, (const T *)x, scale, bias, rmean, rvar,
                           factor, eps, (T *)y, mean, invstd, stats,
                           (const T *)residual, N, C, HW, S);
      });
    });
    return 0;
  }
  const dim3 grid(C, S), block(bn_block(HW));
  SA_DISPATCH_FLOAT(dtype, T, {
    BN_ACT_SWITCH(act, {
      hipLaunchKernelGGL((bn_apply_train_kernel<T, 1, ACT, true>), grid, block,
                         0, s, (const T *)x, scale, bias, rmean, rvar, factor,
                         eps, (T *)y, mean, invstd, stats, (const T *)residual,
                         N, C, HW, S);
    });
  });
  return 0;
}

int sa_bn_fwd_train(int dtype, const void *x, const float *scale,
                    const float *bias, float *rmean, float *rvar, float factor,
                    float eps, void *y, float *mean, float *invstd, int N,
                    int C, int HW, float *work, hipStream_t s) {
  return sa_bn_fwd_train_act(dtype, x, scale, bias, rmean, rvar, factor, eps,
                             y, mean, invstd, N, C, HW, work, 0, nullptr, s);
}

int sa_bn_fwd_infer_act(int dtype, const void *x, const float *scale,
                        const float *bias, const float *rmean,
                        const float *rvar, float eps, void *y, int N, int C,
                        int HW, int act, const void *residual, hipStream_t s) {
  const int64_t total = (int64_t)N * C * HW;
  if (total <= 0) return 0;
  if (total >= ((int64_t)1 << 31)) return -2;
  if (act == 2 && residual == nullptr) return -5;
  {
    // widest vector (16 bytes at most) that divides the plane and keeps
    // every access aligned
    const int es = dtype == kF32 ? 4 : 2;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(x) |
                           reinterpret_cast<uintptr_t>(y) |
                           reinterpret_cast<uintptr_t>(residual);
    int vec = 16 / es;
    while (vec > 1 && (HW % vec != 0 || (bits & (uintptr_t)(vec * es - 1))))
      vec >>= 1;
    if (vec >= 4) {
      const int64_t units = total / vec;
      SA_DISPATCH_FLOAT(dtype, T, {
        if (vec == 8) {
          if constexpr (sizeof(T) == 2)
            hipLaunchKernelGGL((bn_infer_vec_kernel<T, 8>),
                               dim3(grid_for(units, 256, 2)), dim3(256), 0, s,
                               (const T *)x, scale, bias, rmean, rvar, eps,
                               (T *)y, units, C, Div32(HW / 8), Div32(C), act,
                               (const T *)residual);
        } else {
          hipLaunchKernelGGL((bn_infer_vec_kernel<T, 4>),
                             dim3(grid_for(units, 256, 2)), dim3(256), 0, s,
                             (const T *)x, scale, bias, rmean, rvar, eps,
                             (T *)y, units, C, Div32(HW / 4), Div32(C), act,
                             (const T *)residual);
        }
      });
      return 0;
    }
  }
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((bn_infer_kernel<T>), dim3(grid_for(total, 256, 4)),
                       dim3(256), 0, s, (const T *)x, scale, bias, rmean, rvar,
                       eps, (T *)y, total, C, Div32(HW), Div32(C), act,
                       (const T *)residual);
  });
  return 0;
}

int sa_bn_fwd_infer(int dtype, const void *x, const float *scale,
                    const float *bias, const float *rmean, const float *rvar,
                    float eps, void *y, int N, int C, int HW, hipStream_t s) {
  return sa_bn_fwd_infer_act(dtype, x, scale, bias, rmean, rvar, eps, y, N, C,
                             HW, 0, nullptr, s);
}

int sa_bn_bwd_act(int dtype, const void *dy, const void *x, const void *y,
                  const float *scale, const float *bias, const float *mean,
                  const float *invstd, void *dx, void *dres, float *dscale,
                  float *dbias, int N, int C, int HW, float *work, int act,
                  hipStream_t s) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  if (act == 2 && (y == nullptr || dres == nullptr)) return -5;
  if (act == 1 && bias == nullptr) return -5;
  const int S = bn_splits(N, C);
  {
    const bool v4 = bn_vec4(dtype, HW, dy, x, dx, act == 2 ? y : nullptr) &&
                    (reinterpret_cast<uintptr_t>(dres) & 15) == 0;
    const int per = v4 ? HW / 4 : HW;
    const int64_t units = (int64_t)N * per;
    const int threads = bn_channel_threads(units);
    if (threads) {
      SA_DISPATCH_FLOAT(dtype, T, {
        BN_ACT_SWITCH(act, {
          if (v4)
            hipLaunchKernelGGL((bn_bwd_channel_kernel<T, 4, ACT>), dim3(C),
                               dim3(threads), 0, s, (const T *)dy,
                               (const T *)x, (const T *)y, scale, bias, mean,
                               invstd, (T *)dx, (T *)dres, dscale, dbias, N, C,
                               HW, (int)units, Div32(per));
          else
            hipLaunchKernelGGL((bn_bwd_channel_kernel<T, 1, ACT>), dim3(C),
                               dim3(threads), 0, s, (const T *)dy,
                               (const T *)x, (const T *)y, scale, bias, mean,
                               invstd, (T *)dx, (T *)dres, dscale, dbias, N, C,
                               HW, (int)units, Div32(per));
        });
      });
      return 0;
    }
    if (v4 && bn_sweep(C) && units < ((int64_t)1 << 31)) {
      SA_DISPATCH_FLOAT(dtype, T, {
        BN_ACT_SWITCH(act, {
          hipLaunchKernelGGL((bn_bwd_sweep_kernel<T, 4, ACT>), dim3(C),
                             dim3(1024), 0, s, (const T *)dy, (const T *)x,
                             (const T *)y, scale, bias, mean, invstd, (T *)dx,
                             (T *)dres, dscale, dbias, N, C, HW, (int)units,
                             Div32(per));
        });
      });
      return 0;
    }
  }
  if (bn_vec4(dtype, HW, dy, x, dx, act == 2 ? y : nullptr) &&
      (reinterpret_cast<uintptr_t>(dres) & 15) == 0) {
    const dim3 grid(C, S), block(bn_block(HW / 4));
    SA_DISPATCH_FLOAT(dtype, T, {
      BN_ACT_SWITCH(act, {
        hipLaunchKernelGGL((bn_bwd_stats_kernel<T, 4, ACT>), grid, block, 0, s,
                           (const T *)dy, (const T *)x, (const T *)y, scale,
                           bias, mean, invstd, work, (T *)dres, N, C, HW, S);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 4, ACT>), grid, block, 0, s,
                           (const T *)dy, (const T *)x, (const T *)y, scale,
                           bias, mean, invstd, (T *)dx, (T *)dres, dscale,
                           dbias, work, N, C, HW, S);
      });
    });
    return 0;
  }
  const dim3 grid(C, S), block(bn_block(HW));
  SA_DISPATCH_FLOAT(dtype, T, {
    BN_ACT_SWITCH(act, {
      hipLaunchKernelGGL((bn_bwd_stats_kernel<T, 1, ACT>), grid, block, 0, s,
                         (const T *)dy, (const T *)x, (const T *)y, scale, bias,
                         mean, invstd, work, (T *)dres, N, C, HW, S);
      hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1, ACT>), grid, block, 0, s,
                         (const T *)dy, (const T *)x, (const T *)y, scale, bias,
                         mean, invstd, (T *)dx, (T *)dres, dscale, dbias, work,
                         N, C, HW, S);
    });
  });
  return 0;
}

int sa_bn_bwd(int dtype, const void *dy, const void *x, const float *scale,
              const float *mean, const float *invstd, void *dx, float *dscale,
              float *dbias, int N, int C, int HW, float *work, hipStream_t s) {
  return sa_bn_bwd_act(dtype, dy, x, nullptr, scale, nullptr, mean, invstd, dx,
                       nullptr, dscale, dbias, N, C, HW, work, 0, s);
}
