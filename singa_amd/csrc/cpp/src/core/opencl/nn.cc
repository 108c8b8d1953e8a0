// Launchers of the OpenCL convolution / pooling kernels (nn.cl).
#include <algorithm>

#include "runtime.h"

namespace singa {
namespace ocl {

namespace {

// field-for-field images of the kernels' argument structs
struct ConvShape {
  cl_int N, C, H, W, K, R, S, P, Q, stride_h, stride_w, pad_h, pad_w, groups;
};
struct PoolShape {
  cl_int N, C, H, W, P, Q, kh, kw, sh, sw, ph, pw, is_max;
};

ConvShape Shape(const ConvGeom &g) {
  return ConvShape{g.N, g.C, g.H, g.W, g.K, g.R, g.S, g.P, g.Q, g.stride_h,
                   g.stride_w, g.pad_h, g.pad_w, g.groups};
}

PoolShape Shape(const PoolGeom &g) {
  return PoolShape{g.N, g.C, g.H, g.W, g.P, g.Q, g.kh, g.kw, g.sh, g.sw,
                   g.ph, g.pw, g.is_max};
}

constexpr size_t kGroup = 256;

size_t Items(Runtime *rt, int64_t n) {
  size_t groups = static_cast<size_t>((n + kGroup - 1) / kGroup);
  const size_t cap = static_cast<size_t>(std::max(1, rt->compute_units)) * 16;
  return std::max<size_t>(1, std::min(groups, cap)) * kGroup;
}

size_t Groups(Runtime *rt, int64_t n) {
  const size_t cap = static_cast<size_t>(std::max(1, rt->compute_units)) * 16;
  return std::max<size_t>(1, std::min<size_t>(n, cap)) * kGroup;
}

}  // namespace

int conv_fwd(Runtime *rt, const ConvGeom *g, const void *x, const void *w,
             const void *bias, void *y) {
  cl_kernel k = rt->Kernel("k_conv_fwd");
  rt->Arg(k, 0, Shape(*g));
  rt->ArgPtr(k, 1, x);
  rt->ArgPtr(k, 2, w);
  rt->ArgPtr(k, 3, bias ? bias : w);   // never read without has_bias
  rt->Arg(k, 4, static_cast<cl_int>(bias != nullptr));
  rt->ArgPtr(k, 5, y);
  rt->Launch(k, Items(rt, (int64_t)g->N * g->K * g->P * g->Q), kGroup);
  return 0;
}

int conv_bwd_data(Runtime *rt, const ConvGeom *g, const void *dy,
                  const void *w, void *dx) {
  cl_kernel k = rt->Kernel("k_conv_bwd_data");
  rt->Arg(k, 0, Shape(*g));
  rt->ArgPtr(k, 1, dy);
  rt->ArgPtr(k, 2, w);
  rt->ArgPtr(k, 3, dx);
  rt->Launch(k, Items(rt, (int64_t)g->N * g->C * g->H * g->W), kGroup);
  return 0;
}

int conv_bwd_weight(Runtime *rt, const ConvGeom *g, const void *dy,
                    const void *x, void *dw) {
  cl_kernel k = rt->Kernel("k_conv_bwd_weight");
  rt->Arg(k, 0, Shape(*g));
  rt->ArgPtr(k, 1, dy);
  rt->ArgPtr(k, 2, x);
  rt->ArgPtr(k, 3, dw);
  const int64_t taps = (int64_t)g->K * (g->C / g->groups) * g->R * g->S;
  rt->Launch(k, Groups(rt, taps), kGroup);
  return 0;
}

int conv_bwd_bias(Runtime *rt, const void *dy, void *db, int N, int K,
                  int PQ) {
  cl_kernel k = rt->Kernel("k_conv_bwd_bias");
  rt->ArgPtr(k, 0, dy);
  rt->ArgPtr(k, 1, db);
  rt->Arg(k, 2, static_cast<cl_int>(N));
  rt->Arg(k, 3, static_cast<cl_int>(K));
  rt->Arg(k, 4, static_cast<cl_int>(PQ));
  rt->Launch(k, Groups(rt, K), kGroup);
  return 0;
}

int pool_fwd(Runtime *rt, const PoolGeom *g, const void *x, void *y) {
  cl_kernel k = rt->Kernel("k_pool_fwd");
  rt->Arg(k, 0, Shape(*g));
  rt->ArgPtr(k, 1, x);
  rt->ArgPtr(k, 2, y);
  rt->Launch(k, Items(rt, (int64_t)g->N * g->C * g->P * g->Q), kGroup);
  return 0;
}

int pool_bwd(Runtime *rt, const PoolGeom *g, const void *dy, const void *x,
             const void *y, void *dx) {
  cl_kernel k = rt->Kernel("k_pool_bwd");
  rt->Arg(k, 0, Shape(*g));
  rt->ArgPtr(k, 1, dy);
  rt->ArgPtr(k, 2, x);
  rt->ArgPtr(k, 3, y);
  rt->ArgPtr(k, 4, dx);
  rt->Launch(k, Items(rt, (int64_t)g->N * g->C * g->H * g->W), kGroup);
  return 0;
}

int bn_fwd_train(Runtime *rt, const void *x, const float *scale,
                 const float *bias, float *rmean, float *rvar, float factor,
                 float eps, void *y, float *mean, float *invstd, int N, int C,
                 int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  cl_kernel k = rt->Kernel("k_bn_fwd_train");
  int i = 0;
  rt->ArgPtr(k, i++, x);
  rt->ArgPtr(k, i++, scale);
  rt->ArgPtr(k, i++, bias);
  rt->ArgPtr(k, i++, rmean);
  rt->ArgPtr(k, i++, rvar);
  rt->Arg(k, i++, factor);
  rt->Arg(k, i++, eps);
  rt->ArgPtr(k, i++, y);
  rt->ArgPtr(k, i++, mean);
  rt->ArgPtr(k, i++, invstd);
  rt->Arg(k, i++, static_cast<cl_int>(N));
  rt->Arg(k, i++, static_cast<cl_int>(C));
  rt->Arg(k, i++, static_cast<cl_int>(HW));
  rt->Launch(k, Groups(rt, C), kGroup);
  return 0;
}

int bn_fwd_infer(Runtime *rt, const void *x, const float *scale,
                 const float *bias, const float *rmean, const float *rvar,
                 float eps, void *y, int N, int C, int HW) {
  const int64_t total = (int64_t)N * C * HW;
  if (total <= 0) return 0;
  cl_kernel k = rt->Kernel("k_bn_fwd_infer");
  int i = 0;
  rt->ArgPtr(k, i++, x);
  rt->ArgPtr(k, i++, scale);
  rt->ArgPtr(k, i++, bias);
  rt->ArgPtr(k, i++, rmean);
  rt->ArgPtr(k, i++, rvar);
  rt->Arg(k, i++, eps);
  rt->ArgPtr(k, i++, y);
  rt->Arg(k, i++, static_cast<cl_long>(total));
  rt->Arg(k, i++, static_cast<cl_int>(C));
  rt->Arg(k, i++, static_cast<cl_int>(HW));
  rt->Launch(k, Items(rt, total), kGroup);
  return 0;
}

int bn_bwd(Runtime *rt, const void *dy, const void *x, const float *scale,
           const float *mean, const float *invstd, void *dx, float *dscale,
           float *dbias, int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  cl_kernel k = rt->Kernel("k_bn_bwd");
  int i = 0;
  rt->ArgPtr(k, i++, dy);
  rt->ArgPtr(k, i++, x);
  rt->ArgPtr(k, i++, scale);
  rt->ArgPtr(k, i++, mean);
  rt->ArgPtr(k, i++, invstd);
  rt->ArgPtr(k, i++, dx);
  rt->ArgPtr(k, i++, dscale);
  rt->ArgPtr(k, i++, dbias);
  rt->Arg(k, i++, static_cast<cl_int>(N));
  rt->Arg(k, i++, static_cast<cl_int>(C));
  rt->Arg(k, i++, static_cast<cl_int>(HW));
  rt->Launch(k, Groups(rt, C), kGroup);
  return 0;
}

}  // namespace ocl
}  // namespace singa
