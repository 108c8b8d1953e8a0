// Peak-rate probes for v_mfma_f32_32x32x2_f32 on gfx950: MFMA only, and MFMA
// fed by LDS reads the way the convolution / product kernels feed it.
//   hipcc --offload-arch=gfx950 -O3 -w -o mfma_rate tools/probes/mfma_rate.hip && ./mfma_rate
// Measured on an MI355X: 134-156 TF/s in every variant (one wave per SIMD and
// one accumulator are enough; two LDS reads per MFMA cost nothing), i.e. the
// inner loops of the float32 kernels are not what keeps them at 35-45 % of
// the matrix peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;

template <int NACC, bool LDS, int READS>
__global__ void __launch_bounds__(256) probe(float *out, int iters) {
  __shared__ float sm[64 * 68 + 64 * 132];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 68 + 64 * 132; i += 256) sm[i] = 1.0f + i * 1e-6f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = lane * 0.001f, bv = 1.0f + lane * 0.002f;
  const float *us = sm + (lane & 31), *vs = sm + 64 * 68 + (lane & 31);
  const int lk = lane >> 5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 32; ks += 2) {
      float b[NACC];
      if (LDS) {
        av = us[(ks + lk) * 68];
#pragma unroll
        for (int a = 0; a < NACC; ++a) b[a] = a < READS ? vs[(ks + lk) * 132 + 32 * a] : bv;
      } else {
#pragma unroll
        for (int a = 0; a < NACC; ++a) b[a] = bv;
      }
#pragma unroll
      for (int a = 0; a < NACC; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[a], acc[a], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS, int READS>
void run(const char *name, int blocks) {
  float *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NACC, LDS, READS>), dim3(blocks), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NACC, LDS, READS>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
  printf("%-44s blocks %4d: %.3f ms  %.1f TF/s\n", name, blocks, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 1024}) {
    run<1, false, 0>("mfma only, 1 accumulator", blocks);
    run<2, false, 0>("mfma only, 2 accumulators", blocks);
    run<4, false, 0>("mfma only, 4 accumulators", blocks);
    run<2, true, 2>("LDS-fed (1 A + 2 B reads per 2 mfma)", blocks);
    run<4, true, 2>("LDS-fed (1 A + 2 B reads per 4 mfma)", blocks);
    run<1, true, 1>("LDS-fed (1 A + 1 B read per mfma)", blocks);
  }
  return 0;
}
