// Fused SGD step: gradient scaling (1 / world_size), L2 weight decay,
// momentum with dampening, optional Nesterov and the parameter update in one
// pass over (p, g, v): 3 reads + 2 writes per element instead of the 4-5
// separate axpy / scale launches (each re-reading its operands) of the
// composed update.  Hyper-parameters are read from device memory so a
// captured hipGraph follows learning-rate and momentum schedules.
//
// Semantics: reference python/singa/opt.py:269-310.
#include "common.cuh"
#include "kernels.h"

namespace sa {

template <typename T, bool HAS_BUF>
__global__ void sgd_kernel(T *__restrict__ p, const T *__restrict__ g,
                           T *__restrict__ buf, const float *__restrict__ lr_p,
                           const float *__restrict__ mom_p,
                           const float *__restrict__ damp_p,
                           const float *__restrict__ wd_p, int nesterov,
                           float scale, int64_t n) {
  const float lr = lr_p[0], mom = mom_p[0], damp = damp_p[0], wd = wd_p[0];
  constexpr int VEC = 16 / sizeof(T);
  struct alignas(16) V { T v[VEC]; };
  const int64_t nvec = n / VEC;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  auto upd = [&](float pv, float gv, float &bv) {
    gv = gv * scale + wd * pv;
    if (HAS_BUF) {
      bv = mom * bv + (1.f - damp) * gv;
      gv = nesterov ? gv + mom * bv : bv;
    }
    return pv - lr * gv;
  };
  V *pv = reinterpret_cast<V *>(p);
  const V *gv = reinterpret_cast<const V *>(g);
  V *bv = reinterpret_cast<V *>(buf);
  for (int64_t i = tid; i < nvec; i += step) {
    V pp = pv[i], gg = gv[i], bb;
    if (HAS_BUF) bb = bv[i];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float b = HAS_BUF ? to_f<T>(bb.v[k]) : 0.f;
      pp.v[k] = from_f<T>(upd(to_f<T>(pp.v[k]), to_f<T>(gg.v[k]), b));
      if (HAS_BUF) bb.v[k] = from_f<T>(b);
    }
    pv[i] = pp;
    if (HAS_BUF) bv[i] = bb;
  }
  for (int64_t i = nvec * VEC + tid; i < n; i += step) {
    float b = HAS_BUF ? to_f<T>(buf[i]) : 0.f;
    p[i] = from_f<T>(upd(to_f<T>(p[i]), to_f<T>(g[i]), b));
    if (HAS_BUF) buf[i] = from_f<T>(b);
  }
}

// ---- many tensors per launch ----------------------------------------------------
// The per-tensor pointers travel in the kernel argument block (not in a device
// table), so a captured hipGraph node carries them and no host-to-device copy
// is needed at replay.  Each workgroup updates one 4096-element chunk of one
// tensor; chunk_start[] is the prefix sum of chunks per tensor.
template <typename T>
__global__ void sgd_multi_kernel(SgdBatch batch, const float *__restrict__ lr_p,
                                 const float *__restrict__ mom_p,
                                 const float *__restrict__ damp_p,
                                 const float *__restrict__ wd_p, int nesterov,
                                 float scale) {
  const float lr = lr_p[0], mom = mom_p[0], damp = damp_p[0], wd = wd_p[0];
  int t = 0;
  const int chunk = blockIdx.x;
  while (t + 1 < batch.count && chunk >= batch.chunk_start[t + 1]) ++t;
  const int64_t base = (int64_t)(chunk - batch.chunk_start[t]) * kSgdChunk;
  const int64_t n = batch.n[t];
  T *__restrict__ p = static_cast<T *>(batch.p[t]) + base;
  const T *__restrict__ g = static_cast<const T *>(batch.g[t]) + base;
  T *__restrict__ buf =
      batch.buf[t] ? static_cast<T *>(batch.buf[t]) + base : nullptr;
  const int64_t len = n - base < kSgdChunk ? n - base : kSgdChunk;
  const bool has_buf = batch.buf[t] != nullptr;
  auto upd = [&](float pv, float gv, float &bv) {
    gv = gv * scale + wd * pv;
    if (has_buf) {
      bv = mom * bv + (1.f - damp) * gv;
      gv = nesterov ? gv + mom * bv : bv;
    }
    return pv - lr * gv;
  };
  if (batch.master[t]) {
    // 16-bit parameter with a float master copy and float momentum: steps
    // smaller than the parameter's rounding unit still accumulate
    float *__restrict__ mp = batch.master[t] + base;
    float *__restrict__ fb =
        has_buf ? static_cast<float *>(batch.buf[t]) + base : nullptr;
    for (int64_t i = threadIdx.x; i < len; i += blockDim.x) {
      float b = has_buf ? fb[i] : 0.f;
      const float v = upd(mp[i], to_f<T>(g[i]), b);
      mp[i] = v;
      p[i] = from_f<T>(v);
      if (has_buf) fb[i] = b;
    }
    return;
  }
  constexpr int VEC = 16 / sizeof(T);
  struct alignas(16) V { T v[VEC]; };
  const bool aligned =
      ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
        reinterpret_cast<uintptr_t>(buf)) & 15) == 0;
  const int64_t nvec = aligned ? len / VEC : 0;
  V *pv = reinterpret_cast<V *>(p);
  const V *gv = reinterpret_cast<const V *>(g);
  V *bv = reinterpret_cast<V *>(buf);
  for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x) {
    V pp = pv[i], gg = gv[i], bb;
    if (has_buf) bb = bv[i];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float b = has_buf ? to_f<T>(bb.v[k]) : 0.f;
      pp.v[k] = from_f<T>(upd(to_f<T>(pp.v[k]), to_f<T>(gg.v[k]), b));
      if (has_buf) bb.v[k] = from_f<T>(b);
    }
    pv[i] = pp;
    if (has_buf) bv[i] = bb;
  }
  for (int64_t i = nvec * VEC + threadIdx.x; i < len; i += blockDim.x) {
    float b = has_buf ? to_f<T>(buf[i]) : 0.f;
    p[i] = from_f<T>(upd(to_f<T>(p[i]), to_f<T>(g[i]), b));
    if (has_buf) buf[i] = from_f<T>(b);
  }
}

// ---- adaptive rules, many tensors per launch -------------------------------------
// RMSProp, AdaGrad and Adam as one pass over (p, g, state[, state2]) for a
// whole list of parameters: gradient scaling, L2 weight decay, the moment
// updates, Adam's bias correction and the step itself.  The composed form
// (reference python/singa/opt.py:392-441, 489-533, 608-683) is 8 to 14
// element-wise launches per parameter, each re-reading its operands.
//
//   rule 0  r = a r + (1 - a) g^2;            p -= lr g / sqrt(r + eps)
//   rule 1  h += g^2;                         p -= lr g / sqrt(h + eps)
//   rule 2  m = a m + (1 - a) g; v = b v + (1 - b) g^2; t = step + 1
//           p -= lr (m / (1 - a^t)) / (sqrt(v / (1 - b^t)) + eps)
template <typename T>
__global__ void adaptive_multi_kernel(AdaptBatch batch, int rule,
                                      const float *__restrict__ lr_p,
                                      const float *__restrict__ wd_p,
                                      const float *__restrict__ eps_p,
                                      const float *__restrict__ a_p,
                                      const float *__restrict__ b_p,
                                      const float *__restrict__ step_p,
                                      float scale) {
  const float lr = lr_p[0], wd = wd_p[0], eps = eps_p[0];
  const float a = a_p ? a_p[0] : 0.f, b = b_p ? b_p[0] : 0.f;
  float c1 = 1.f, c2 = 1.f;
  if (rule == 2) {
    const float t = step_p[0] + 1.f;
    c1 = 1.f / (1.f - powf(a, t));
    c2 = 1.f / (1.f - powf(b, t));
  }
  int t = 0;
  const int chunk = blockIdx.x;
  while (t + 1 < batch.count && chunk >= batch.chunk_start[t + 1]) ++t;
  const int64_t base = (int64_t)(chunk - batch.chunk_start[t]) * kSgdChunk;
  const int64_t n = batch.n[t];
  const int64_t len = n - base < kSgdChunk ? n - base : kSgdChunk;
  T *__restrict__ p = static_cast<T *>(batch.p[t]) + base;
  const T *__restrict__ g = static_cast<const T *>(batch.g[t]) + base;
  T *__restrict__ s1 = static_cast<T *>(batch.s1[t]) + base;
  T *__restrict__ s2 =
      batch.s2[t] ? static_cast<T *>(batch.s2[t]) + base : nullptr;
  for (int64_t i = threadIdx.x; i < len; i += blockDim.x) {
    const float pv = to_f<T>(p[i]);
    const float gv = to_f<T>(g[i]) * scale + wd * pv;
    float u = to_f<T>(s1[i]);
    float upd;
    if (rule == 0) {
      u = a * u + (1.f - a) * gv * gv;
      upd = gv / sqrtf(u + eps);
    } else if (rule == 1) {
      u += gv * gv;
      upd = gv / sqrtf(u + eps);
    } else {
      float w = to_f<T>(s2[i]);
      u = a * u + (1.f - a) * gv;
      w = b * w + (1.f - b) * gv * gv;
      s2[i] = from_f<T>(w);
      upd = (u * c1) / (sqrtf(w * c2) + eps);
    }
    s1[i] = from_f<T>(u);
    p[i] = from_f<T>(pv - lr * upd);
  }
}

}  // namespace sa

using namespace sa;

int sa_adaptive_update_multi(int dtype, int rule, int count, void *const *p,
                             const void *const *g, void *const *s1,
                             void *const *s2, const int64_t *n,
                             const float *lr, const float *wd,
                             const float *eps, const float *a, const float *b,
                             const float *step, float scale, hipStream_t s) {
  if (rule < 0 || rule > 2) return -2;
  if (rule == 2 && (s2 == nullptr || step == nullptr)) return -2;
  for (int first = 0; first < count; first += kSgdMaxTensors) {
    AdaptBatch bt;
    bt.count = count - first < kSgdMaxTensors ? count - first : kSgdMaxTensors;
    int chunks = 0;
    for (int i = 0; i < bt.count; ++i) {
      bt.p[i] = p[first + i];
      bt.g[i] = g[first + i];
      bt.s1[i] = s1[first + i];
      bt.s2[i] = s2 ? s2[first + i] : nullptr;
      bt.n[i] = n[first + i];
      bt.chunk_start[i] = chunks;
      chunks += (int)((n[first + i] + kSgdChunk - 1) / kSgdChunk);
    }
    bt.chunk_start[bt.count] = chunks;
    if (chunks == 0) continue;
    SA_DISPATCH_FLOAT(dtype, T, {
      hipLaunchKernelGGL((adaptive_multi_kernel<T>), dim3(chunks), dim3(256),
                         0, s, bt, rule, lr, wd, eps, a, b, step, scale);
    });
  }
  return 0;
}

int sa_sgd_update_multi(int dtype, int count, void *const *p,
                        const void *const *g, void *const *buf,
                        float *const *master,
                        const int64_t *n, const float *lr,
                        const float *momentum, const float *dampening,
                        const float *weight_decay, int nesterov, float scale,
                        hipStream_t s) {
  for (int first = 0; first < count; first += kSgdMaxTensors) {
    SgdBatch b;
    b.count = count - first < kSgdMaxTensors ? count - first : kSgdMaxTensors;
    int chunks = 0;
    for (int i = 0; i < b.count; ++i) {
      b.p[i] = p[first + i];
      b.g[i] = g[first + i];
      b.buf[i] = buf ? buf[first + i] : nullptr;
      b.master[i] = master ? master[first + i] : nullptr;
      b.n[i] = n[first + i];
      b.chunk_start[i] = chunks;
      chunks += (int)((n[first + i] + kSgdChunk - 1) / kSgdChunk);
    }
    b.chunk_start[b.count] = chunks;
    if (chunks == 0) continue;
    SA_DISPATCH_FLOAT(dtype, T, {
      hipLaunchKernelGGL((sgd_multi_kernel<T>), dim3(chunks), dim3(256), 0, s,
                         b, lr, momentum, dampening, weight_decay, nesterov,
                         scale);
    });
  }
  return 0;
}

int sa_sgd_update(int dtype, void *p, const void *g, void *buf,
                  const float *lr, const float *momentum,
                  const float *dampening, const float *weight_decay,
                  int nesterov, float scale, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  const uintptr_t bits = reinterpret_cast<uintptr_t>(p) |
                         reinterpret_cast<uintptr_t>(g) |
                         reinterpret_cast<uintptr_t>(buf);
  if (bits & 15) return -2;
  SA_DISPATCH_FLOAT(dtype, T, {
    const int grid = grid_for(n, 256, 16 / sizeof(T));
    if (buf)
      hipLaunchKernelGGL((sgd_kernel<T, true>), dim3(grid), dim3(256), 0, s,
                         (T *)p, (const T *)g, (T *)buf, lr, momentum,
                         dampening, weight_decay, nesterov, scale, n);
    else
      hipLaunchKernelGGL((sgd_kernel<T, false>), dim3(grid), dim3(256), 0, s,
                         (T *)p, (const T *)g, (T *)nullptr, lr, momentum,
                         dampening, weight_decay, nesterov, scale, n);
  });
  return 0;
}
