#include "singa/model/nn_ops.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "../../../hip/kernels.h"
#include "../core/opencl/runtime.h"

namespace singa {

namespace {
inline hipStream_t S(Context *ctx) { return static_cast<hipStream_t>(ctx->stream); }
inline ocl::Runtime *CL(Context *ctx) { return static_cast<ocl::Runtime *>(ctx->cl); }
#define SA_OK(expr)                                                     \
  do {                                                                  \
    CHECK_EQ((expr), 0) << "kernel launcher rejected arguments";        \
  } while (0)
// the OpenCL device computes in float32 only
#define CL_F32(t) CHECK_EQ((t), kFloat32) << "the OpenCL device computes in float32"
constexpr float kBnEps = 1e-5f;

ConvGeom MakeGeom(const Shape &x, const ConvParams &p) {
  ConvGeom g{};
  g.N = x[0]; g.C = x[1]; g.H = x[2]; g.W = x[3];
  g.K = p.num_filters; g.R = p.kernel_h; g.S = p.kernel_w;
  g.stride_h = p.stride_h; g.stride_w = p.stride_w;
  g.pad_h = p.pad_h; g.pad_w = p.pad_w;
  g.groups = p.group;
  g.P = ConvOutSize(g.H, g.R, g.stride_h, g.pad_h);
  g.Q = ConvOutSize(g.W, g.S, g.stride_w, g.pad_w);
  CHECK_GT(g.P, 0);
  CHECK_GT(g.Q, 0);
  CHECK_EQ(g.C % g.groups, 0);
  CHECK_EQ(g.K % g.groups, 0);
  return g;
}

// visit every (output element, filter tap, input element) triple
template <typename F>
void ConvLoop(const ConvGeom &g, F &&f) {
  const int Cg = g.C / g.groups, Kg = g.K / g.groups;
  for (int n = 0; n < g.N; ++n)
    for (int k = 0; k < g.K; ++k) {
      const int grp = k / Kg;
      for (int p = 0; p < g.P; ++p)
        for (int q = 0; q < g.Q; ++q) {
          const size_t yo = ((size_t(n) * g.K + k) * g.P + p) * g.Q + q;
          for (int c = 0; c < Cg; ++c)
            for (int r = 0; r < g.R; ++r) {
              const int h = p * g.stride_h - g.pad_h + r;
              if (h < 0 || h >= g.H) continue;
              for (int s = 0; s < g.S; ++s) {
                const int w = q * g.stride_w - g.pad_w + s;
                if (w < 0 || w >= g.W) continue;
                const size_t xo =
                    ((size_t(n) * g.C + grp * Cg + c) * g.H + h) * g.W + w;
                const size_t wo = ((size_t(k) * Cg + c) * g.R + r) * g.S + s;
                f(yo, xo, wo);
              }
            }
        }
    }
}
}  // namespace

int ConvOutSize(int in, int kernel, int stride, int pad) {
  return (in + 2 * pad - kernel) / stride + 1;
}

namespace {
// Tile edge of the Winograd route the HIP device takes for this layer, 0 for
// the direct kernels: float32 3x3 / stride 1 / pad 1 layers with enough
// channels and positions for the (m+2)^2 batched products to fill the chip
// (measured on ResNet-50 shapes at batch 32: F(4x4, 3x3) wins from 64
// channels at 56 x 56 to 512 channels at 7 x 7).  SINGA_CONV_WINOGRAD=0
// switches it off.
int WinogradTile(const ConvGeom &g, DataType t) {
  static const bool off = [] {
    const char *e = std::getenv("SINGA_CONV_WINOGRAD");
    return e && e[0] == '0';
  }();
  if (off || t != kFloat32 || g.groups != 1 || g.R != 3 || g.S != 3 ||
      g.stride_h != 1 || g.stride_w != 1 || g.pad_h != 1 || g.pad_w != 1)
    return 0;
  if (g.C < 32 || g.K < 32 || int64_t(g.N) * g.H * g.W < 512) return 0;
  return 4;
}
}  // namespace

Tensor ConvForward(const Tensor &x, const Tensor &w, const Tensor *b,
                   const ConvParams &p) {
  CHECK_EQ(x.nDim(), 4u);
  ConvGeom g = MakeGeom(x.shape(), p);
  Tensor y(Shape{size_t(g.N), size_t(g.K), size_t(g.P), size_t(g.Q)},
           x.device(), x.data_type());
  Block *xb = x.block(), *wb = w.block(), *yb = y.block();
  Block *bb = b ? b->block() : nullptr;
  std::vector<Block *> reads{xb, wb};
  if (bb) reads.push_back(bb);
  auto dev = x.device();
  DataType t = x.data_type();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *xp = static_cast<const float *>(xb->data());
      const float *wp = static_cast<const float *>(wb->data());
      float *yp = static_cast<float *>(yb->mutable_data());
      const size_t pq = size_t(g.P) * g.Q;
      for (int n = 0; n < g.N; ++n)
        for (int k = 0; k < g.K; ++k) {
          float bv = bb ? static_cast<const float *>(bb->data())[k] : 0.f;
          std::fill_n(yp + (size_t(n) * g.K + k) * pq, pq, bv);
        }
      ConvLoop(g, [&](size_t yo, size_t xo, size_t wo) {
        yp[yo] += xp[xo] * wp[wo];
      });
    }, reads, {yb}, "ConvForward");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx)) {
        CL_F32(t);
        SA_OK(ocl::conv_fwd(CL(ctx), &g, xb->data(), wb->data(),
                            bb ? bb->data() : nullptr, yb->mutable_data()));
      } else if (const int m = WinogradTile(g, t)) {
        Tensor ws(Shape{size_t(sa_conv3x3_winograd_workspace(
                      m, g.N, g.C, g.K, g.H, g.W))}, dev, t);
        SA_OK(sa_conv3x3_winograd(t, m, 0, xb->data(), wb->data(),
                                  bb ? bb->data() : nullptr, nullptr,
                                  yb->mutable_data(), g.N, g.C, g.K, g.H, g.W,
                                  ws.block()->mutable_data(), S(ctx)));
      } else {
        SA_OK(sa_conv_fwd(t, &g, xb->data(), wb->data(),
                          bb ? bb->data() : nullptr, yb->mutable_data(),
                          S(ctx)));
      }
    }, reads, {yb}, "ConvForward");
  }
  return y;
}

Tensor ConvBackwardData(const Tensor &dy, const Tensor &w, const Shape &x_shape,
                        const ConvParams &p) {
  ConvGeom g = MakeGeom(x_shape, p);
  Tensor dx(x_shape, dy.device(), dy.data_type());
  Block *gb = dy.block(), *wb = w.block(), *xb = dx.block();
  auto dev = dy.device();
  DataType t = dy.data_type();
  const size_t nx = dx.Size();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *gp = static_cast<const float *>(gb->data());
      const float *wp = static_cast<const float *>(wb->data());
      float *xp = static_cast<float *>(xb->mutable_data());
      std::fill_n(xp, nx, 0.f);
      ConvLoop(g, [&](size_t yo, size_t xo, size_t wo) {
        xp[xo] += gp[yo] * wp[wo];
      });
    }, {gb, wb}, {xb}, "ConvBackwardData");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx)) {
        CL_F32(t);
        SA_OK(ocl::conv_bwd_data(CL(ctx), &g, gb->data(), wb->data(),
                                 xb->mutable_data()));
      } else if (const int m = WinogradTile(g, t)) {
        Tensor ws(Shape{size_t(sa_conv3x3_winograd_workspace(
                      m, g.N, g.C, g.K, g.H, g.W))}, dev, t);
        SA_OK(sa_conv3x3_winograd(t, m, 1, gb->data(), wb->data(), nullptr,
                                  nullptr, xb->mutable_data(), g.N, g.C, g.K,
                                  g.H, g.W, ws.block()->mutable_data(),
                                  S(ctx)));
      } else {
        SA_OK(sa_conv_bwd_data(t, &g, gb->data(), wb->data(),
                               xb->mutable_data(), S(ctx)));
      }
    }, {gb, wb}, {xb}, "ConvBackwardData");
  }
  return dx;
}

Tensor ConvBackwardWeight(const Tensor &dy, const Tensor &x,
                          const Shape &w_shape, const ConvParams &p) {
  ConvGeom g = MakeGeom(x.shape(), p);
  Tensor dw(w_shape, dy.device(), dy.data_type());
  Block *gb = dy.block(), *xb = x.block(), *wb = dw.block();
  auto dev = dy.device();
  DataType t = dy.data_type();
  const size_t nw = dw.Size();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *gp = static_cast<const float *>(gb->data());
      const float *xp = static_cast<const float *>(xb->data());
      float *wp = static_cast<float *>(wb->mutable_data());
      std::fill_n(wp, nw, 0.f);
      ConvLoop(g, [&](size_t yo, size_t xo, size_t wo) {
        wp[wo] += gp[yo] * xp[xo];
      });
    }, {gb, xb}, {wb}, "ConvBackwardWeight");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx)) {
        CL_F32(t);
        SA_OK(ocl::conv_bwd_weight(CL(ctx), &g, gb->data(), xb->data(),
                                   wb->mutable_data()));
      } else {
        SA_OK(sa_conv_bwd_weight(t, &g, gb->data(), xb->data(),
                                 wb->mutable_data(), S(ctx)));
      }
    }, {gb, xb}, {wb}, "ConvBackwardWeight");
  }
  return dw;
}

Tensor ConvBackwardBias(const Tensor &dy) {
  CHECK_EQ(dy.nDim(), 4u);
  const int N = dy.shape(0), K = dy.shape(1), PQ = dy.shape(2) * dy.shape(3);
  Tensor db(Shape{size_t(K)}, dy.device(), dy.data_type());
  Block *gb = dy.block(), *bb = db.block();
  auto dev = dy.device();
  DataType t = dy.data_type();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *gp = static_cast<const float *>(gb->data());
      float *bp = static_cast<float *>(bb->mutable_data());
      for (int k = 0; k < K; ++k) {
        double acc = 0;
        for (int n = 0; n < N; ++n)
          for (int i = 0; i < PQ; ++i) acc += gp[(size_t(n) * K + k) * PQ + i];
        bp[k] = static_cast<float>(acc);
      }
    }, {gb}, {bb}, "ConvBackwardBias");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx)) {
        CL_F32(t);
        SA_OK(ocl::conv_bwd_bias(CL(ctx), gb->data(), bb->mutable_data(), N, K,
                                 PQ));
      } else {
        SA_OK(sa_conv_bwd_bias(t, gb->data(), bb->mutable_data(), N, K, PQ,
                               S(ctx)));
      }
    }, {gb}, {bb}, "ConvBackwardBias");
  }
  return db;
}

// ---- pooling --------------------------------------------------------------------
namespace {
PoolGeom MakePool(const Shape &x, const PoolParams &p) {
  PoolGeom g;
  g.N = x[0]; g.C = x[1]; g.H = x[2]; g.W = x[3];
  g.kh = p.kernel_h; g.kw = p.kernel_w; g.sh = p.stride_h; g.sw = p.stride_w;
  g.ph = p.pad_h; g.pw = p.pad_w; g.is_max = p.is_max ? 1 : 0;
  g.P = ConvOutSize(g.H, g.kh, g.sh, g.ph);
  g.Q = ConvOutSize(g.W, g.kw, g.sw, g.pw);
  CHECK_GT(g.P, 0);
  CHECK_GT(g.Q, 0);
  return g;
}
}  // namespace

Tensor PoolForward(const Tensor &x, const PoolParams &p) {
  CHECK_EQ(x.nDim(), 4u);
  PoolGeom g = MakePool(x.shape(), p);
  Tensor y(Shape{size_t(g.N), size_t(g.C), size_t(g.P), size_t(g.Q)},
           x.device(), x.data_type());
  Block *xb = x.block(), *yb = y.block();
  auto dev = x.device();
  DataType t = x.data_type();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *xp = static_cast<const float *>(xb->data());
      float *yp = static_cast<float *>(yb->mutable_data());
      for (int nc = 0; nc < g.N * g.C; ++nc)
        for (int oh = 0; oh < g.P; ++oh)
          for (int ow = 0; ow < g.Q; ++ow) {
            float acc = g.is_max ? -std::numeric_limits<float>::max() : 0.f;
            int cnt = 0;
            for (int r = 0; r < g.kh; ++r) {
              int h = oh * g.sh - g.ph + r;
              if (h < 0 || h >= g.H) continue;
              for (int s = 0; s < g.kw; ++s) {
                int w = ow * g.sw - g.pw + s;
                if (w < 0 || w >= g.W) continue;
                float v = xp[(size_t(nc) * g.H + h) * g.W + w];
                acc = g.is_max ? std::max(acc, v) : acc + v;
                ++cnt;
              }
            }
            yp[(size_t(nc) * g.P + oh) * g.Q + ow] =
                g.is_max ? acc : acc / std::max(cnt, 1);
          }
    }, {xb}, {yb}, "PoolForward");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx)) {
        CL_F32(t);
        SA_OK(ocl::pool_fwd(CL(ctx), &g, xb->data(), yb->mutable_data()));
      } else {
        SA_OK(sa_pool_fwd(t, &g, xb->data(), yb->mutable_data(), S(ctx)));
      }
    }, {xb}, {yb}, "PoolForward");
  }
  return y;
}

Tensor PoolBackward(const Tensor &dy, const Tensor &x, const Tensor &y,
                    const PoolParams &p) {
  PoolGeom g = MakePool(x.shape(), p);
  Tensor dx(x.shape(), x.device(), x.data_type());
  Block *gb = dy.block(), *xb = x.block(), *yb = y.block(), *db = dx.block();
  auto dev = x.device();
  DataType t = x.data_type();
  const size_t nx = dx.Size();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *gp = static_cast<const float *>(gb->data());
      const float *xp = static_cast<const float *>(xb->data());
      const float *yp = static_cast<const float *>(yb->data());
      float *dp = static_cast<float *>(db->mutable_data());
      std::fill_n(dp, nx, 0.f);
      for (int nc = 0; nc < g.N * g.C; ++nc)
        for (int oh = 0; oh < g.P; ++oh)
          for (int ow = 0; ow < g.Q; ++ow) {
            const size_t yo = (size_t(nc) * g.P + oh) * g.Q + ow;
            int cnt = 0;
            if (!g.is_max)
              for (int r = 0; r < g.kh; ++r) {
                int h = oh * g.sh - g.ph + r;
                if (h < 0 || h >= g.H) continue;
                for (int s = 0; s < g.kw; ++s) {
                  int w = ow * g.sw - g.pw + s;
                  if (w >= 0 && w < g.W) ++cnt;
                }
              }
            bool done = false;
            for (int r = 0; r < g.kh && !done; ++r) {
              int h = oh * g.sh - g.ph + r;
              if (h < 0 || h >= g.H) continue;
              for (int s = 0; s < g.kw; ++s) {
                int w = ow * g.sw - g.pw + s;
                if (w < 0 || w >= g.W) continue;
                const size_t xo = (size_t(nc) * g.H + h) * g.W + w;
                if (g.is_max) {
                  if (xp[xo] == yp[yo]) {  // first maximum takes the gradient
                    dp[xo] += gp[yo];
                    done = true;
                    break;
                  }
                } else {
                  dp[xo] += gp[yo] / cnt;
                }
              }
            }
          }
    }, {gb, xb, yb}, {db}, "PoolBackward");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx)) {
        CL_F32(t);
        SA_OK(ocl::pool_bwd(CL(ctx), &g, gb->data(), xb->data(), yb->data(),
                            db->mutable_data()));
      } else {
        SA_OK(sa_pool_bwd(t, &g, gb->data(), xb->data(), yb->data(),
                          db->mutable_data(), S(ctx)));
      }
    }, {gb, xb, yb}, {db}, "PoolBackward");
  }
  return dx;
}

// ---- batch norm -------------------------------------------------------------------
namespace {
void BnDims(const Tensor &x, int *N, int *C, int *HW) {
  CHECK(x.nDim() == 4u || x.nDim() == 2u);
  *N = x.shape(0);
  *C = x.shape(1);
  *HW = x.nDim() == 4 ? x.shape(2) * x.shape(3) : 1;
}
}  // namespace

Tensor BatchNormForwardTrain(const Tensor &x, const Tensor &scale,
                             const Tensor &bias, Tensor *running_mean,
                             Tensor *running_var, float factor, Tensor *mean,
                             Tensor *invstd) {
  int N, C, HW;
  BnDims(x, &N, &C, &HW);
  Tensor y(x.shape(), x.device(), x.data_type());
  mean->ResetLike(scale);
  invstd->ResetLike(scale);
  Block *xb = x.block(), *sb = scale.block(), *bb = bias.block(),
        *rm = running_mean->block(), *rv = running_var->block(),
        *yb = y.block(), *mb = mean->block(), *ib = invstd->block();
  auto dev = x.device();
  DataType t = x.data_type();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *xp = static_cast<const float *>(xb->data());
      const float *sp = static_cast<const float *>(sb->data());
      const float *bp = static_cast<const float *>(bb->data());
      float *rmp = static_cast<float *>(rm->mutable_data());
      float *rvp = static_cast<float *>(rv->mutable_data());
      float *yp = static_cast<float *>(yb->mutable_data());
      float *mp = static_cast<float *>(mb->mutable_data());
      float *ip = static_cast<float *>(ib->mutable_data());
      const double cnt = double(N) * HW;
      for (int c = 0; c < C; ++c) {
        double s = 0, ss = 0;
        for (int n = 0; n < N; ++n)
          for (int i = 0; i < HW; ++i) {
            double v = xp[(size_t(n) * C + c) * HW + i];
            s += v; ss += v * v;
          }
        double mu = s / cnt, var = std::max(ss / cnt - mu * mu, 0.0);
        mp[c] = float(mu);
        ip[c] = float(1.0 / std::sqrt(var + kBnEps));
        double unbiased = cnt > 1 ? var * cnt / (cnt - 1) : var;
        rmp[c] = (1 - factor) * rmp[c] + factor * float(mu);
        rvp[c] = (1 - factor) * rvp[c] + factor * float(unbiased);
        for (int n = 0; n < N; ++n)
          for (int i = 0; i < HW; ++i) {
            size_t o = (size_t(n) * C + c) * HW + i;
            yp[o] = (xp[o] - mp[c]) * ip[c] * sp[c] + bp[c];
          }
      }
    }, {xb, sb, bb, rm, rv}, {yb, mb, ib, rm, rv}, "BatchNormForwardTrain");
  } else {
    if (dev->lang() == kOpencl) {
      CL_F32(t);
      dev->Exec([=](Context *ctx) {
        SA_OK(ocl::bn_fwd_train(
            CL(ctx), xb->data(), static_cast<const float *>(sb->data()),
            static_cast<const float *>(bb->data()),
            static_cast<float *>(rm->mutable_data()),
            static_cast<float *>(rv->mutable_data()), factor, kBnEps,
            yb->mutable_data(), static_cast<float *>(mb->mutable_data()),
            static_cast<float *>(ib->mutable_data()), N, C, HW));
      }, {xb, sb, bb, rm, rv}, {yb, mb, ib, rm, rv}, "BatchNormForwardTrain");
      return y;
    }
    Tensor work(Shape{size_t(std::max<int64_t>(sa_bn_workspace(N, C, HW), 1))},
                dev, kFloat32);
    Block *wk = work.block();
    dev->Exec([=](Context *ctx) {
      SA_OK(sa_bn_fwd_train(
          t, xb->data(), static_cast<const float *>(sb->data()),
          static_cast<const float *>(bb->data()),
          static_cast<float *>(rm->mutable_data()),
          static_cast<float *>(rv->mutable_data()), factor, kBnEps,
          yb->mutable_data(), static_cast<float *>(mb->mutable_data()),
          static_cast<float *>(ib->mutable_data()), N, C, HW,
          static_cast<float *>(wk->mutable_data()), S(ctx)));
    }, {xb, sb, bb, rm, rv}, {yb, mb, ib, rm, rv, wk}, "BatchNormForwardTrain");
  }
  return y;
}

Tensor BatchNormForwardInfer(const Tensor &x, const Tensor &scale,
                             const Tensor &bias, const Tensor &running_mean,
                             const Tensor &running_var) {
  int N, C, HW;
  BnDims(x, &N, &C, &HW);
  Tensor y(x.shape(), x.device(), x.data_type());
  Block *xb = x.block(), *sb = scale.block(), *bb = bias.block(),
        *rm = running_mean.block(), *rv = running_var.block(), *yb = y.block();
  auto dev = x.device();
  DataType t = x.data_type();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *xp = static_cast<const float *>(xb->data());
      const float *sp = static_cast<const float *>(sb->data());
      const float *bp = static_cast<const float *>(bb->data());
      const float *rmp = static_cast<const float *>(rm->data());
      const float *rvp = static_cast<const float *>(rv->data());
      float *yp = static_cast<float *>(yb->mutable_data());
      for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
          float inv = 1.f / std::sqrt(rvp[c] + kBnEps);
          for (int i = 0; i < HW; ++i) {
            size_t o = (size_t(n) * C + c) * HW + i;
            yp[o] = (xp[o] - rmp[c]) * inv * sp[c] + bp[c];
          }
        }
    }, {xb, sb, bb, rm, rv}, {yb}, "BatchNormForwardInfer");
  } else {
    if (dev->lang() == kOpencl) {
      CL_F32(t);
      dev->Exec([=](Context *ctx) {
        SA_OK(ocl::bn_fwd_infer(CL(ctx), xb->data(),
                                static_cast<const float *>(sb->data()),
                                static_cast<const float *>(bb->data()),
                                static_cast<const float *>(rm->data()),
                                static_cast<const float *>(rv->data()), kBnEps,
                                yb->mutable_data(), N, C, HW));
      }, {xb, sb, bb, rm, rv}, {yb}, "BatchNormForwardInfer");
      return y;
    }
    dev->Exec([=](Context *ctx) {
      SA_OK(sa_bn_fwd_infer(t, xb->data(),
                            static_cast<const float *>(sb->data()),
                            static_cast<const float *>(bb->data()),
                            static_cast<const float *>(rm->data()),
                            static_cast<const float *>(rv->data()), kBnEps,
                            yb->mutable_data(), N, C, HW, S(ctx)));
    }, {xb, sb, bb, rm, rv}, {yb}, "BatchNormForwardInfer");
  }
  return y;
}

Tensor BatchNormBackward(const Tensor &dy, const Tensor &x, const Tensor &scale,
                         const Tensor &mean, const Tensor &invstd,
                         Tensor *dscale, Tensor *dbias) {
  int N, C, HW;
  BnDims(x, &N, &C, &HW);
  Tensor dx(x.shape(), x.device(), x.data_type());
  dscale->ResetLike(scale);
  dbias->ResetLike(scale);
  Block *gb = dy.block(), *xb = x.block(), *sb = scale.block(),
        *mb = mean.block(), *ib = invstd.block(), *db = dx.block(),
        *dsb = dscale->block(), *dbb = dbias->block();
  auto dev = x.device();
  DataType t = x.data_type();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *gp = static_cast<const float *>(gb->data());
      const float *xp = static_cast<const float *>(xb->data());
      const float *sp = static_cast<const float *>(sb->data());
      const float *mp = static_cast<const float *>(mb->data());
      const float *ip = static_cast<const float *>(ib->data());
      float *dp = static_cast<float *>(db->mutable_data());
      float *dsp = static_cast<float *>(dsb->mutable_data());
      float *dbp = static_cast<float *>(dbb->mutable_data());
      const double cnt = double(N) * HW;
      for (int c = 0; c < C; ++c) {
        double sg = 0, sgx = 0;
        for (int n = 0; n < N; ++n)
          for (int i = 0; i < HW; ++i) {
            size_t o = (size_t(n) * C + c) * HW + i;
            double xh = (xp[o] - mp[c]) * ip[c];
            sg += gp[o]; sgx += gp[o] * xh;
          }
        dbp[c] = float(sg);
        dsp[c] = float(sgx);
        for (int n = 0; n < N; ++n)
          for (int i = 0; i < HW; ++i) {
            size_t o = (size_t(n) * C + c) * HW + i;
            double xh = (xp[o] - mp[c]) * ip[c];
            dp[o] = float(sp[c] * ip[c] * (gp[o] - sg / cnt - xh * sgx / cnt));
          }
      }
    }, {gb, xb, sb, mb, ib}, {db, dsb, dbb}, "BatchNormBackward");
  } else {
    if (dev->lang() == kOpencl) {
      CL_F32(t);
      dev->Exec([=](Context *ctx) {
        SA_OK(ocl::bn_bwd(CL(ctx), gb->data(), xb->data(),
                          static_cast<const float *>(sb->data()),
                          static_cast<const float *>(mb->data()),
                          static_cast<const float *>(ib->data()),
                          db->mutable_data(),
                          static_cast<float *>(dsb->mutable_data()),
                          static_cast<float *>(dbb->mutable_data()), N, C, HW));
      }, {gb, xb, sb, mb, ib}, {db, dsb, dbb}, "BatchNormBackward");
      return dx;
    }
    Tensor work(Shape{size_t(std::max<int64_t>(sa_bn_workspace(N, C, HW), 1))},
                dev, kFloat32);
    Block *wk = work.block();
    dev->Exec([=](Context *ctx) {
      SA_OK(sa_bn_bwd(t, gb->data(), xb->data(),
                      static_cast<const float *>(sb->data()),
                      static_cast<const float *>(mb->data()),
                      static_cast<const float *>(ib->data()),
                      db->mutable_data(),
                      static_cast<float *>(dsb->mutable_data()),
                      static_cast<float *>(dbb->mutable_data()), N, C, HW,
                      static_cast<float *>(wk->mutable_data()), S(ctx)));
    }, {gb, xb, sb, mb, ib}, {db, dsb, dbb, wk}, "BatchNormBackward");
  }
  return dx;
}

}  // namespace singa
