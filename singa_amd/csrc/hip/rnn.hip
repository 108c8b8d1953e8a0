// Fused recurrent cell kernels.  The gate pre-activations (batch, 4H), ordered
// i, f, o, g, come from two MFMA GEMMs (x W_x and h W_h); one point-wise
// kernel then produces c, h and the activated gates (saved for backward), and
// one kernel maps (dh, dc) back to the gate gradients.  This replaces ~20
// element-wise launches per time step of the composed cell and the pointwise
// stage of cudnnRNNForwardTraining (reference src/model/operation/rnn.cc).
#include "common.cuh"
#include "kernels.h"

namespace sa {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ void lstm_fwd_kernel(const T *__restrict__ gates,
                                const T *__restrict__ c_prev, T *__restrict__ c,
                                T *__restrict__ h, T *__restrict__ acts,
                                int64_t batch, int64_t H) {
  const int64_t total = batch * H;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += step) {
    const int64_t b = e / H, j = e - b * H;
    const T *gr = gates + b * 4 * H;
    const float i = sigm(to_f<T>(gr[j]));
    const float f = sigm(to_f<T>(gr[H + j]));
    const float o = sigm(to_f<T>(gr[2 * H + j]));
    const float g = tanhf(to_f<T>(gr[3 * H + j]));
    const float cn = f * to_f<T>(c_prev[e]) + i * g;
    c[e] = from_f<T>(cn);
    h[e] = from_f<T>(o * tanhf(cn));
    T *ar = acts + b * 4 * H;
    ar[j] = from_f<T>(i);
    ar[H + j] = from_f<T>(f);
    ar[2 * H + j] = from_f<T>(o);
    ar[3 * H + j] = from_f<T>(g);
  }
}

template <typename T>
__global__ void lstm_bwd_kernel(const T *__restrict__ dh,
                                const T *__restrict__ dc_next,
                                const T *__restrict__ acts,
                                const T *__restrict__ c_prev,
                                const T *__restrict__ c, T *__restrict__ dgates,
                                T *__restrict__ dc_prev, int64_t batch,
                                int64_t H) {
  const int64_t total = batch * H;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += step) {
    const int64_t b = e / H, j = e - b * H;
    const T *ar = acts + b * 4 * H;
    const float i = to_f<T>(ar[j]), f = to_f<T>(ar[H + j]);
    const float o = to_f<T>(ar[2 * H + j]), g = to_f<T>(ar[3 * H + j]);
    const float tc = tanhf(to_f<T>(c[e]));
    const float dhv = to_f<T>(dh[e]);
    const float dcv = to_f<T>(dc_next[e]) + dhv * o * (1.f - tc * tc);
    T *dg = dgates + b * 4 * H;
    dg[j] = from_f<T>(dcv * g * i * (1.f - i));
    dg[H + j] = from_f<T>(dcv * to_f<T>(c_prev[e]) * f * (1.f - f));
    dg[2 * H + j] = from_f<T>(dhv * tc * o * (1.f - o));
    dg[3 * H + j] = from_f<T>(dcv * i * (1.f - g * g));
    dc_prev[e] = from_f<T>(dcv * f);
  }
}

}  // namespace sa

using namespace sa;

int sa_lstm_cell_fwd(int dtype, const void *gates, const void *c_prev, void *c,
                     void *h, void *acts, int64_t batch, int64_t hidden,
                     hipStream_t s) {
  const int64_t total = batch * hidden;
  if (total <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((lstm_fwd_kernel<T>), dim3(grid_for(total, 256)),
                       dim3(256), 0, s, (const T *)gates, (const T *)c_prev,
                       (T *)c, (T *)h, (T *)acts, batch, hidden);
  });
  return 0;
}

int sa_lstm_cell_bwd(int dtype, const void *dh, const void *dc_next,
                     const void *acts, const void *c_prev, const void *c,
                     void *dgates, void *dc_prev, int64_t batch,
                     int64_t hidden, hipStream_t s) {
  const int64_t total = batch * hidden;
  if (total <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((lstm_bwd_kernel<T>), dim3(grid_for(total, 256)),
                       dim3(256), 0, s, (const T *)dh, (const T *)dc_next,
                       (const T *)acts, (const T *)c_prev, (const T *)c,
                       (T *)dgates, (T *)dc_prev, batch, hidden);
  });
  return 0;
}
