// Counter-based random numbers (Philox4x32-10): uniform, normal (Box-Muller)
// and Bernoulli masks.
//
// The generator state {seed, offset} lives in device memory.  Every launch
// reads the offset, each thread derives its counter from (offset + thread
// slot), and a one-thread kernel queued right behind advances the offset by
// the number of 128-bit blocks consumed.  Nothing is read back by the host,
// so the launches are hipGraph-capturable and every replay of a captured
// training step draws fresh numbers.
//
// Replaces curandGenerateUniform / curandGenerateNormal and the threshold
// kernel of the reference (tensor_math_cuda.h:938-996).
#include "common.cuh"
#include "kernels.h"

namespace sa {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32(uint64_t ctr_lo, uint64_t ctr_hi,
                                         uint64_t key) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
  uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}

// (0, 1]
__device__ __forceinline__ float u01(uint32_t v) {
  return ((float)(v >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

enum { RNG_UNIFORM = 0, RNG_NORMAL = 1, RNG_BERNOULLI = 2 };

template <typename T, int MODE>
__global__ void rng_kernel(T *__restrict__ y, int64_t n, float a, float b,
                           const int64_t *__restrict__ state) {
  const uint64_t seed = (uint64_t)state[0];
  const uint64_t base = (uint64_t)state[1];
  const int64_t nquads = (n + 3) / 4;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads;
       q += step) {
    const U4 r = philox4x32(base + (uint64_t)q, 0x5851F42D4C957F2Dull, seed);
    float v[4];
    if (MODE == RNG_NORMAL) {
      const float r0 = sqrtf(-2.f * __logf(u01(r.x)));
      const float r1 = sqrtf(-2.f * __logf(u01(r.z)));
      float s0, c0, s1, c1;
      __sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
      __sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
      v[0] = a + b * r0 * c0;
      v[1] = a + b * r0 * s0;
      v[2] = a + b * r1 * c1;
      v[3] = a + b * r1 * s1;
    } else if (MODE == RNG_UNIFORM) {
      v[0] = a + (b - a) * u01(r.x);
      v[1] = a + (b - a) * u01(r.y);
      v[2] = a + (b - a) * u01(r.z);
      v[3] = a + (b - a) * u01(r.w);
    } else {
      v[0] = u01(r.x) <= a ? 1.f : 0.f;
      v[1] = u01(r.y) <= a ? 1.f : 0.f;
      v[2] = u01(r.z) <= a ? 1.f : 0.f;
      v[3] = u01(r.w) <= a ? 1.f : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = q * 4 + k;
      if (i < n) y[i] = from_f<T>(v[k]);
    }
  }
}

__global__ void rng_advance_kernel(int64_t *state, int64_t quads) {
  state[1] += quads;
}

template <int MODE>
int launch_rng(int dtype, void *y, int64_t n, float a, float b, int64_t *state,
               hipStream_t s) {
  if (n <= 0) return 0;
  const int64_t nquads = (n + 3) / 4;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((rng_kernel<T, MODE>), dim3(grid_for(nquads, 256)),
                       dim3(256), 0, s, (T *)y, n, a, b, state);
  });
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, s, state, nquads);
  return 0;
}

}  // namespace sa

using namespace sa;

int sa_rng_uniform(int dtype, void *y, int64_t n, float lo, float hi,
                   int64_t *state, hipStream_t s) {
  return launch_rng<RNG_UNIFORM>(dtype, y, n, lo, hi, state, s);
}
int sa_rng_normal(int dtype, void *y, int64_t n, float mean, float std,
                  int64_t *state, hipStream_t s) {
  return launch_rng<RNG_NORMAL>(dtype, y, n, mean, std, state, s);
}
int sa_rng_bernoulli(int dtype, void *y, int64_t n, float p, int64_t *state,
                     hipStream_t s) {
  return launch_rng<RNG_BERNOULLI>(dtype, y, n, p, 0.f, state, s);
}
