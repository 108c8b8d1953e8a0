// Tensor-level neural-network primitives shared by the C++ layers: NCHW
// convolution, pooling, batch normalisation.  On HipGPU they launch the
// implicit-GEMM MFMA convolution and the fused reduction kernels; on CppCPU
// they are direct loops.  Semantics follow the reference's operation handles
// (src/model/operation/{convolution,pooling,batchnorm}.cc).
#pragma once
#include "../core/tensor.h"

namespace singa {

struct ConvParams {
  int kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, group;
  int num_filters;
};
int ConvOutSize(int in, int kernel, int stride, int pad);
/// x (N,C,H,W), w (K, C/g, R, S) [+ b (K)] -> y (N,K,P,Q)
Tensor ConvForward(const Tensor &x, const Tensor &w, const Tensor *b,
                   const ConvParams &p);
Tensor ConvBackwardData(const Tensor &dy, const Tensor &w, const Shape &x_shape,
                        const ConvParams &p);
Tensor ConvBackwardWeight(const Tensor &dy, const Tensor &x,
                          const Shape &w_shape, const ConvParams &p);
Tensor ConvBackwardBias(const Tensor &dy);

struct PoolParams {
  int kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w;
  bool is_max;
};
Tensor PoolForward(const Tensor &x, const PoolParams &p);
Tensor PoolBackward(const Tensor &dy, const Tensor &x, const Tensor &y,
                    const PoolParams &p);

/// Training forward on (N,C,H,W) or (N,C): returns y, writes the batch mean
/// and inverse std, and updates running = (1-factor)*running + factor*batch
/// (unbiased variance), eps 1e-5.
Tensor BatchNormForwardTrain(const Tensor &x, const Tensor &scale,
                             const Tensor &bias, Tensor *running_mean,
                             Tensor *running_var, float factor, Tensor *mean,
                             Tensor *invstd);
Tensor BatchNormForwardInfer(const Tensor &x, const Tensor &scale,
                             const Tensor &bias, const Tensor &running_mean,
                             const Tensor &running_var);
/// -> dx; writes dscale, dbias
Tensor BatchNormBackward(const Tensor &dy, const Tensor &x, const Tensor &scale,
                         const Tensor &mean, const Tensor &invstd,
                         Tensor *dscale, Tensor *dbias);

}  // namespace singa
