// pybind11 surface of the HIP kernels.  Device buffers arrive as integer
// addresses (torch ``data_ptr()``), the stream as the integer handle of the
// current torch HIP stream.  Every wrapper checks the launcher's status code
// and raises, so an unsupported dtype can never silently do nothing.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"

namespace py = pybind11;
using sa::Desc;

namespace {

inline void *P(uintptr_t p) { return reinterpret_cast<void *>(p); }
inline const void *CP(uintptr_t p) { return reinterpret_cast<const void *>(p); }
inline hipStream_t S(uintptr_t s) { return reinterpret_cast<hipStream_t>(s); }

void check(int rc, const char *what) {
  if (rc != 0)
    throw std::runtime_error(std::string(what) + ": launcher returned " +
                             std::to_string(rc) +
                             " (-1 dtype, -2 alignment/size, -3 needs float32)");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

Desc make_desc(const std::vector<int64_t> &shape,
               const std::vector<int64_t> &s0, const std::vector<int64_t> &s1,
               const std::vector<int64_t> &s2) {
  Desc d;
  if (shape.size() > 8) throw std::runtime_error("more than 8 dimensions");
  d.ndim = static_cast<int>(shape.size());
  for (int i = 0; i < 8; ++i) {
    d.shape[i] = i < d.ndim ? shape[i] : 1;
    d.s0[i] = i < (int)s0.size() ? s0[i] : 0;
    d.s1[i] = i < (int)s1.size() ? s1[i] : 0;
    d.s2[i] = i < (int)s2.size() ? s2[i] : 0;
  }
  return d;
}

}  // namespace

PYBIND11_MODULE(_hip, m) {
  m.doc() = "singa_amd HIP kernels for gfx950";
  m.attr("arch") = "gfx950";

  // ---- element-wise -------------------------------------------------------
  m.def("set_gemm_f32_tile", [](int t) { sa_set_gemm_f32_tile(t); });
  m.def("set_gemm_h16_tile", [](int t) { sa_set_gemm_h16_tile(t); });
  m.def("get_gemm_f32_tile", []() { return sa_get_gemm_f32_tile(); });
  m.def("set_deterministic", [](int on) { sa_set_deterministic(on); },
        "split reductions combine their partial sums in a fixed order");
  m.def("get_deterministic", []() { return sa_get_deterministic(); });
  m.def("unary_contig", [](int op, int dt_in, int dt_out, uintptr_t x,
                           uintptr_t y, int64_t n, float a, float b,
                           uintptr_t s) {
    check(sa_unary(op, dt_in, dt_out, CP(x), P(y), nullptr, n, a, b, S(s)),
          "unary");
  });
  m.def("unary_strided", [](int op, int dt_in, int dt_out, uintptr_t x,
                            uintptr_t y, std::vector<int64_t> shape,
                            std::vector<int64_t> sx, std::vector<int64_t> sy,
                            int64_t n, float a, float b, uintptr_t s) {
    Desc d = make_desc(shape, sx, {}, sy);
    check(sa_unary(op, dt_in, dt_out, CP(x), P(y), &d, n, a, b, S(s)),
          "unary_strided");
  });
  m.def("binary_contig", [](int op, int dt, uintptr_t x, uintptr_t y,
                            uintptr_t z, int64_t n, uintptr_t s) {
    check(sa_binary(op, dt, CP(x), CP(y), P(z), nullptr, n, S(s)), "binary");
  });
  m.def("binary_strided", [](int op, int dt, uintptr_t x, uintptr_t y,
                             uintptr_t z, std::vector<int64_t> shape,
                             std::vector<int64_t> sx, std::vector<int64_t> sy,
                             std::vector<int64_t> sz, int64_t n, uintptr_t s) {
    Desc d = make_desc(shape, sx, sy, sz);
    check(sa_binary(op, dt, CP(x), CP(y), P(z), &d, n, S(s)),
          "binary_strided");
  });
  m.def("fill", [](int dt, uintptr_t y, int64_t n, float v, uintptr_t s) {
    check(sa_fill(dt, P(y), n, v, S(s)), "fill");
  });
  m.def("axpy", [](int dt, uintptr_t alpha_ptr, float alpha, uintptr_t x,
                   uintptr_t y, int64_t n, uintptr_t s) {
    check(sa_axpy(dt, (const float *)CP(alpha_ptr), alpha, CP(x), P(y), n,
                  S(s)),
          "axpy");
  });

  // ---- reductions ---------------------------------------------------------
  m.def("reduce_rows", [](int op, int dt, uintptr_t x, uintptr_t out,
                          int64_t rows, int64_t cols, uintptr_t s) {
    check(sa_reduce_rows(op, dt, CP(x), P(out), rows, cols, S(s)),
          "reduce_rows");
  });
  m.def("reduce_cols", [](int dt, uintptr_t x, uintptr_t out, int64_t rows,
                          int64_t cols, uintptr_t s) {
    check(sa_reduce_cols(dt, CP(x), P(out), rows, cols, S(s)), "reduce_cols");
  });
  m.def("reduce_cols_fin", [](int dt, uintptr_t x, uintptr_t out, int out_typed,
                              int64_t rows, int64_t cols, uintptr_t sums,
                              int64_t sums_cap, uintptr_t tickets,
                              int64_t tickets_cap, uintptr_t s) {
    const int rc = sa_reduce_cols_fin(
        dt, CP(x), P(out), out_typed, rows, cols,
        reinterpret_cast<float *>(sums), sums_cap,
        reinterpret_cast<unsigned *>(tickets), tickets_cap, S(s));
    if (rc == -3) return false;   // shape beyond the scratch: caller falls back
    check(rc, "reduce_cols_fin");
    return true;
  });
  m.def("reduce_all", [](int op, int dt, uintptr_t x, uintptr_t out, int64_t n,
                         uintptr_t s) {
    check(sa_reduce_all(op, dt, CP(x), P(out), n, S(s)), "reduce_all");
  });

  // ---- softmax / losses -----------------------------------------------------
  m.def("softmax_fwd", [](int dt, uintptr_t x, uintptr_t y, int64_t rows,
                          int64_t cols, uintptr_t s) {
    check(sa_softmax_fwd(dt, CP(x), P(y), rows, cols, S(s)), "softmax_fwd");
  });
  m.def("softmax_bwd", [](int dt, uintptr_t dy, uintptr_t y, uintptr_t dx,
                          int64_t rows, int64_t cols, uintptr_t s) {
    check(sa_softmax_bwd(dt, CP(dy), CP(y), P(dx), rows, cols, S(s)),
          "softmax_bwd");
  });
  m.def("cross_entropy_fwd", [](int dt, uintptr_t p, uintptr_t t, int t_is_int,
                                int t_dt, uintptr_t loss, int64_t rows,
                                int64_t cols, uintptr_t s) {
    check(sa_cross_entropy_fwd(dt, CP(p), CP(t), t_is_int, t_dt, P(loss), rows,
                               cols, S(s)),
          "cross_entropy_fwd");
  });
  m.def("softmax_ce_bwd", [](int dt, uintptr_t p, uintptr_t t, int t_is_int,
                             int t_dt, int64_t rows, int64_t cols,
                             uintptr_t s) {
    check(sa_softmax_ce_bwd(dt, P(p), CP(t), t_is_int, t_dt, rows, cols, S(s)),
          "softmax_ce_bwd");
  });
  m.def("softmax_ce_fused_fwd", [](int dt, uintptr_t x, uintptr_t t,
                                   int t_is_int, int t_dt, uintptr_t p,
                                   uintptr_t loss, int64_t rows, int64_t cols,
                                   uintptr_t s) {
    check(sa_softmax_ce_fused_fwd(dt, CP(x), CP(t), t_is_int, t_dt, P(p),
                                  P(loss), rows, cols, S(s)),
          "softmax_ce_fused_fwd");
  });
  m.def("softmax_ce_fused_bwd", [](int dt, uintptr_t p, uintptr_t t,
                                   int t_is_int, int t_dt, uintptr_t dx,
                                   int64_t rows, int64_t cols, uintptr_t s) {
    check(sa_softmax_ce_fused_bwd(dt, CP(p), CP(t), t_is_int, t_dt, P(dx), rows,
                                  cols, S(s)),
          "softmax_ce_fused_bwd");
  });

  // ---- random -----------------------------------------------------------------
  m.def("rng_uniform", [](int dt, uintptr_t y, int64_t n, float lo, float hi,
                          uintptr_t state, uintptr_t s) {
    check(sa_rng_uniform(dt, P(y), n, lo, hi, (int64_t *)P(state), S(s)),
          "rng_uniform");
  });
  m.def("rng_normal", [](int dt, uintptr_t y, int64_t n, float mean, float sd,
                         uintptr_t state, uintptr_t s) {
    check(sa_rng_normal(dt, P(y), n, mean, sd, (int64_t *)P(state), S(s)),
          "rng_normal");
  });
  m.def("rng_bernoulli", [](int dt, uintptr_t y, int64_t n, float p,
                            uintptr_t state, uintptr_t s) {
    check(sa_rng_bernoulli(dt, P(y), n, p, (int64_t *)P(state), S(s)),
          "rng_bernoulli");
  });

  // ---- gemm ---------------------------------------------------------------------
  m.def("gemm", [](int dt, int64_t M, int64_t N, int64_t K, float alpha,
                   uintptr_t A, int64_t a_rs, int64_t a_cs, int64_t a_bs,
                   uintptr_t B, int64_t b_rs, int64_t b_cs, int64_t b_bs,
                   float beta, uintptr_t C, int64_t ldc, int64_t c_bs,
                   int64_t batch, uintptr_t bias, int act, uintptr_t s) {
    check(sa_gemm(dt, M, N, K, alpha, CP(A), a_rs, a_cs, a_bs, CP(B), b_rs,
                  b_cs, b_bs, beta, P(C), ldc, c_bs, batch, CP(bias), act,
                  S(s)),
          "gemm");
  });

  m.def("gemm_f32_slices", [](int64_t M, int64_t N, int64_t kc, int64_t k_total,
                              uintptr_t A, int64_t a_rs, int64_t a_cs,
                              uintptr_t B, int64_t b_rs, int64_t b_cs,
                              uintptr_t part, int64_t splits, uintptr_t s) {
    check(sa_gemm_f32_slices(M, N, kc, k_total, CP(A), a_rs, a_cs, CP(B), b_rs,
                             b_cs, (float *)P(part), splits, S(s)),
          "gemm_f32_slices");
  });

  // returns the status instead of raising: -3 = layout not eligible
  m.def("gemm_h16_partial", [](int dt, int64_t M, int64_t N, int64_t K,
                               uintptr_t A, int64_t a_rs, int64_t a_cs,
                               int64_t a_bs, uintptr_t B, int64_t b_rs,
                               int64_t b_cs, int64_t b_bs, uintptr_t part,
                               int64_t batch, uintptr_t s, int64_t k_total) {
    return sa_gemm_h16_partial(dt, M, N, K, CP(A), a_rs, a_cs, a_bs, CP(B),
                               b_rs, b_cs, b_bs, (float *)P(part), batch,
                               k_total, S(s));
  }, py::arg("dt"), py::arg("M"), py::arg("N"), py::arg("K"), py::arg("A"),
     py::arg("a_rs"), py::arg("a_cs"), py::arg("a_bs"), py::arg("B"),
     py::arg("b_rs"), py::arg("b_cs"), py::arg("b_bs"), py::arg("part"),
     py::arg("batch"), py::arg("s"), py::arg("k_total") = 0);
  m.def("gemm_splitk_reduce", [](int dt, uintptr_t part, int splits, int64_t M,
                                 int64_t N, float alpha, float beta,
                                 uintptr_t C, int64_t ldc, uintptr_t bias,
                                 int act, uintptr_t s) {
    check(sa_gemm_splitk_reduce(dt, (const float *)CP(part), splits, M, N,
                                alpha, beta, P(C), ldc, CP(bias), act, S(s)),
          "gemm_splitk_reduce");
  });

  m.def("copy_bytes", [](uintptr_t dst, uintptr_t src, int64_t bytes,
                         uintptr_t s) {
    check(sa_copy_bytes(P(dst), CP(src), bytes, S(s)), "copy_bytes");
  });
  m.def("zero_bytes", [](uintptr_t dst, int64_t bytes, uintptr_t s) {
    check(sa_zero_bytes(P(dst), bytes, S(s)), "zero_bytes");
  });

  // ---- layer norm / gelu ------------------------------------------------------
  m.def("layernorm_fwd", [](int dt, uintptr_t x, uintptr_t gamma, uintptr_t beta,
                            uintptr_t y, uintptr_t mean, uintptr_t rstd,
                            int64_t rows, int D, float eps, uintptr_t s) {
    check(sa_layernorm_fwd(dt, CP(x), CP(gamma), CP(beta), P(y),
                           (float *)P(mean), (float *)P(rstd), rows, D, eps,
                           S(s)),
          "layernorm_fwd");
  });
  m.def("layernorm_bwd", [](int dt, uintptr_t dy, uintptr_t x, uintptr_t gamma,
                            uintptr_t mean, uintptr_t rstd, uintptr_t dx,
                            uintptr_t dgamma, uintptr_t dbeta, int64_t rows,
                            int D, uintptr_t s) {
    check(sa_layernorm_bwd(dt, CP(dy), CP(x), CP(gamma),
                           (const float *)CP(mean), (const float *)CP(rstd),
                           P(dx), (float *)P(dgamma), (float *)P(dbeta), rows,
                           D, S(s)),
          "layernorm_bwd");
  });
  // parameter gradients in one launch; false: scratch too small
  m.def("layernorm_bwd_param", [](int dt, uintptr_t dy, uintptr_t x,
                                  uintptr_t mean, uintptr_t rstd,
                                  uintptr_t dgamma, uintptr_t dbeta,
                                  int out_typed, uintptr_t sums,
                                  int64_t sums_cap, uintptr_t tickets,
                                  int64_t tickets_cap, int64_t rows, int D,
                                  uintptr_t s) {
    const int rc = sa_layernorm_bwd_param(
        dt, CP(dy), CP(x), (const float *)CP(mean), (const float *)CP(rstd),
        P(dgamma), P(dbeta), out_typed, (float *)P(sums), sums_cap,
        (unsigned *)P(tickets), tickets_cap, rows, D, S(s));
    if (rc == -3) return false;
    check(rc, "layernorm_bwd_param");
    return true;
  });
  m.def("gelu_fwd", [](int dt, uintptr_t x, uintptr_t y, int64_t n,
                       uintptr_t s) {
    check(sa_gelu_fwd(dt, CP(x), P(y), n, S(s)), "gelu_fwd");
  });
  m.def("gelu_bwd", [](int dt, uintptr_t dy, uintptr_t x, uintptr_t dx,
                       int64_t n, uintptr_t s) {
    check(sa_gelu_bwd(dt, CP(dy), CP(x), P(dx), n, S(s)), "gelu_bwd");
  });

  // ---- convolution -----------------------------------------------------------------
  py::class_<ConvGeom>(m, "ConvGeom")
      .def(py::init([](int N, int C, int H, int W, int K, int R, int Sw, int P_,
                       int Q, int sh, int sw, int ph, int pw, int groups) {
        ConvGeom g{N, C, H, W, K, R, Sw, P_, Q, sh, sw, ph, pw, groups};
        return g;
      }))
      .def_readwrite("N", &ConvGeom::N)
      .def_readwrite("tile", &ConvGeom::tile)
      .def_readwrite("split_target", &ConvGeom::split_target)
      .def("set_scratch", [](ConvGeom &g, uintptr_t p, int64_t elems) {
        g.scratch = reinterpret_cast<float *>(p);
        g.scratch_elems = elems;
      })
      .def("set_det_scratch", [](ConvGeom &g, uintptr_t p, int64_t elems) {
        g.det_scratch = reinterpret_cast<float *>(p);
        g.det_elems = elems;
      })
      .def("set_side_zero", [](ConvGeom &g, uintptr_t p, int64_t words) {
        g.side_zero = reinterpret_cast<uint32_t *>(p);
        g.side_zero_words = words;
      })
      .def_readwrite("out_zeroed", &ConvGeom::out_zeroed)
      .def("set_stats", [](ConvGeom &g, uintptr_t stats, uintptr_t shift) {
        g.stats = reinterpret_cast<float *>(stats);
        g.stats_shift = reinterpret_cast<const float *>(shift);
      })
      .def("set_prof", [](ConvGeom &g, uintptr_t p) {
        g.prof = reinterpret_cast<unsigned long long *>(p);
      });
  m.def("conv_fwd", [](int dt, const ConvGeom &g, uintptr_t x, uintptr_t w,
                       uintptr_t bias, uintptr_t y, uintptr_t s) {
    check(sa_conv_fwd(dt, &g, CP(x), CP(w), CP(bias), P(y), S(s)), "conv_fwd");
  });
  // returns false when the geometry needs the implicit-GEMM kernel
  m.def("conv1x1_h16", [](int dt, int mode, int64_t n_img, int64_t C,
                          int64_t K, int64_t HW, uintptr_t act, uintptr_t w,
                          uintptr_t addend, uintptr_t out, uintptr_t s) {
    const int rc = sa_conv1x1_h16(dt, mode, n_img, C, K, HW, CP(act), CP(w),
                                  CP(addend), P(out), S(s));
    if (rc == -3) return false;
    check(rc, "conv1x1_h16");
    return true;
  });
  m.def("conv3x3_h16", [](int dt, int mode, int n_img, int C, int K, int H,
                          int W, uintptr_t act, uintptr_t w, uintptr_t wt,
                          uintptr_t addend, uintptr_t out, uintptr_t s) {
    const int rc = sa_conv3x3_h16(dt, mode, n_img, C, K, H, W, CP(act), CP(w),
                                  P(wt), CP(addend), P(out), S(s));
    if (rc == -3) return false;
    check(rc, "conv3x3_h16");
    return true;
  });
  m.def("conv_flat_h16", [](int dt, int mode, int n_img, int C, int K, int H,
                            int W, int taps, uintptr_t act, uintptr_t w,
                            uintptr_t wt, uintptr_t addend, uintptr_t out,
                            uintptr_t s) {
    const int rc = sa_conv_flat_h16(dt, mode, n_img, C, K, H, W, taps, CP(act),
                                    CP(w), P(wt), CP(addend), P(out), S(s));
    if (rc == -3) return false;
    check(rc, "conv_flat_h16");
    return true;
  });
  m.def("subsample2", [](int dt, uintptr_t x, uintptr_t xs, int64_t planes,
                         int H, int W, uintptr_t s) {
    check(sa_subsample2(dt, CP(x), P(xs), planes, H, W, S(s)), "subsample2");
  });
  m.def("upsample2_add", [](int dt, uintptr_t dxs, uintptr_t addend,
                            uintptr_t dx, int64_t planes, int H, int W,
                            uintptr_t s) {
    check(sa_upsample2_add(dt, CP(dxs), CP(addend), P(dx), planes, H, W, S(s)),
          "upsample2_add");
  });
  m.def("conv_wgrad_h16_splits", &sa_conv_wgrad_h16_splits);
  m.def("conv_wgrad_h16", [](int dt, int n_img, int C, int K, int H, int W,
                             int taps, uintptr_t dy, uintptr_t x,
                             uintptr_t part, int splits, uintptr_t dw,
                             uintptr_t s) {
    const int rc = sa_conv_wgrad_h16(dt, n_img, C, K, H, W, taps, CP(dy), CP(x),
                                     reinterpret_cast<float *>(part), splits,
                                     P(dw), S(s));
    if (rc == -3) return false;
    check(rc, "conv_wgrad_h16");
    return true;
  });
  m.def("conv_bwd_data", [](int dt, const ConvGeom &g, uintptr_t dy,
                            uintptr_t w, uintptr_t dx, uintptr_t s) {
    check(sa_conv_bwd_data(dt, &g, CP(dy), CP(w), P(dx), S(s)),
          "conv_bwd_data");
  });
  m.def("conv_bwd_data_acc", [](int dt, const ConvGeom &g, uintptr_t dy,
                                uintptr_t w, uintptr_t addend, uintptr_t dx,
                                uintptr_t s) {
    check(sa_conv_bwd_data_acc(dt, &g, CP(dy), CP(w), CP(addend), P(dx), S(s)),
          "conv_bwd_data_acc");
  });
  m.def("conv_bwd_weight", [](int dt, const ConvGeom &g, uintptr_t dy,
                              uintptr_t x, uintptr_t dw, uintptr_t s) {
    check(sa_conv_bwd_weight(dt, &g, CP(dy), CP(x), P(dw), S(s)),
          "conv_bwd_weight");
  });
  m.def("conv_bwd_bias", [](int dt, uintptr_t dy, uintptr_t db, int N, int K,
                            int PQ, uintptr_t s) {
    check(sa_conv_bwd_bias(dt, CP(dy), P(db), N, K, PQ, S(s)),
          "conv_bwd_bias");
  });

  // ---- batch norm ---------------------------------------------------------------------
  m.def("bn_workspace", &sa_bn_workspace);
  m.def("bn_fwd_train", [](int dt, uintptr_t x, uintptr_t scale, uintptr_t bias,
                           uintptr_t rmean, uintptr_t rvar, float factor,
                           float eps, uintptr_t y, uintptr_t mean,
                           uintptr_t invstd, int N, int C, int HW,
                           uintptr_t work, uintptr_t s) {
    check(sa_bn_fwd_train(dt, CP(x), (const float *)CP(scale),
                          (const float *)CP(bias), (float *)P(rmean),
                          (float *)P(rvar), factor, eps, P(y), (float *)P(mean),
                          (float *)P(invstd), N, C, HW, (float *)P(work), S(s)),
          "bn_fwd_train");
  });
  m.def("bn_fwd_infer", [](int dt, uintptr_t x, uintptr_t scale, uintptr_t bias,
                           uintptr_t rmean, uintptr_t rvar, float eps,
                           uintptr_t y, int N, int C, int HW, uintptr_t s) {
    check(sa_bn_fwd_infer(dt, CP(x), (const float *)CP(scale),
                          (const float *)CP(bias), (const float *)CP(rmean),
                          (const float *)CP(rvar), eps, P(y), N, C, HW, S(s)),
          "bn_fwd_infer");
  });
  m.def("bn_bwd", [](int dt, uintptr_t dy, uintptr_t x, uintptr_t scale,
                     uintptr_t mean, uintptr_t invstd, uintptr_t dx,
                     uintptr_t dscale, uintptr_t dbias, int N, int C, int HW,
                     uintptr_t work, uintptr_t s) {
    check(sa_bn_bwd(dt, CP(dy), CP(x), (const float *)CP(scale),
                    (const float *)CP(mean), (const float *)CP(invstd), P(dx),
                    (float *)P(dscale), (float *)P(dbias), N, C, HW,
                    (float *)P(work), S(s)),
          "bn_bwd");
  });

  m.def("bn_fwd_train_act", [](int dt, uintptr_t x, uintptr_t scale,
                               uintptr_t bias, uintptr_t rmean, uintptr_t rvar,
                               float factor, float eps, uintptr_t y,
                               uintptr_t mean, uintptr_t invstd, int N, int C,
                               int HW, uintptr_t work, int act, uintptr_t res,
                               uintptr_t s) {
    check(sa_bn_fwd_train_act(dt, CP(x), (const float *)CP(scale),
                              (const float *)CP(bias), (float *)P(rmean),
                              (float *)P(rvar), factor, eps, P(y),
                              (float *)P(mean), (float *)P(invstd), N, C, HW,
                              (float *)P(work), act, CP(res), S(s)),
          "bn_fwd_train_act");
  });
  m.def("bn_fwd_train_sums", [](int dt, uintptr_t x, uintptr_t scale,
                                uintptr_t bias, uintptr_t rmean,
                                uintptr_t rvar, float factor, float eps,
                                uintptr_t y, uintptr_t mean, uintptr_t invstd,
                                int N, int C, int HW, uintptr_t stats, int act,
                                uintptr_t res, uintptr_t s) {
    check(sa_bn_fwd_train_sums(dt, CP(x), (const float *)CP(scale),
                               (const float *)CP(bias), (float *)P(rmean),
                               (float *)P(rvar), factor, eps, P(y),
                               (float *)P(mean), (float *)P(invstd), N, C, HW,
                               (float *)P(stats), act, CP(res), S(s)),
          "bn_fwd_train_sums");
  });
  m.attr("stat_copies") = kStatCopies;
  m.def("chan_sums", [](int dt, uintptr_t y, uintptr_t stats, uintptr_t shift,
                        int N, int K, int PQ, uintptr_t s) {
    check(sa_chan_sums(dt, CP(y), (float *)P(stats), (const float *)CP(shift),
                       N, K, PQ, S(s)),
          "chan_sums");
  });
  m.def("bn_fwd_infer_act", [](int dt, uintptr_t x, uintptr_t scale,
                               uintptr_t bias, uintptr_t rmean, uintptr_t rvar,
                               float eps, uintptr_t y, int N, int C, int HW,
                               int act, uintptr_t res, uintptr_t s) {
    check(sa_bn_fwd_infer_act(dt, CP(x), (const float *)CP(scale),
                              (const float *)CP(bias), (const float *)CP(rmean),
                              (const float *)CP(rvar), eps, P(y), N, C, HW, act,
                              CP(res), S(s)),
          "bn_fwd_infer_act");
  });
  m.def("bn_bwd_act", [](int dt, uintptr_t dy, uintptr_t x, uintptr_t y,
                         uintptr_t scale, uintptr_t bias, uintptr_t mean,
                         uintptr_t invstd, uintptr_t dx, uintptr_t dres,
                         uintptr_t dscale, uintptr_t dbias, int N, int C,
                         int HW, uintptr_t work, int act, uintptr_t s) {
    check(sa_bn_bwd_act(dt, CP(dy), CP(x), CP(y), (const float *)CP(scale),
                        (const float *)CP(bias), (const float *)CP(mean),
                        (const float *)CP(invstd), P(dx), P(dres),
                        (float *)P(dscale), (float *)P(dbias), N, C, HW,
                        (float *)P(work), act, S(s)),
          "bn_bwd_act");
  });

  // ---- pooling ----------------------------------------------------------------------------
  py::class_<PoolGeom>(m, "PoolGeom")
      .def(py::init([](int N, int C, int H, int W, int P_, int Q, int kh,
                       int kw, int sh, int sw, int ph, int pw, int is_max) {
        PoolGeom g{N, C, H, W, P_, Q, kh, kw, sh, sw, ph, pw, is_max};
        return g;
      }));
  m.def("pool_fwd", [](int dt, const PoolGeom &g, uintptr_t x, uintptr_t y,
                       uintptr_t s) {
    check(sa_pool_fwd(dt, &g, CP(x), P(y), S(s)), "pool_fwd");
  });
  m.def("pool_bwd", [](int dt, const PoolGeom &g, uintptr_t dy, uintptr_t x,
                       uintptr_t y, uintptr_t dx, uintptr_t s) {
    check(sa_pool_bwd(dt, &g, CP(dy), CP(x), CP(y), P(dx), S(s)), "pool_bwd");
  });

  // ---- optimiser ------------------------------------------------------------------------------
  m.def("sgd_update", [](int dt, uintptr_t p, uintptr_t g, uintptr_t buf,
                         uintptr_t lr, uintptr_t mom, uintptr_t damp,
                         uintptr_t wd, int nesterov, float scale, int64_t n,
                         uintptr_t s) {
    check(sa_sgd_update(dt, P(p), CP(g), P(buf), (const float *)CP(lr),
                        (const float *)CP(mom), (const float *)CP(damp),
                        (const float *)CP(wd), nesterov, scale, n, S(s)),
          "sgd_update");
  });

  m.def("sgd_update_multi", [](int dt, std::vector<uintptr_t> p,
                               std::vector<uintptr_t> g,
                               std::vector<uintptr_t> buf,
                               std::vector<uintptr_t> master,
                               std::vector<int64_t> n, uintptr_t lr,
                               uintptr_t mom, uintptr_t damp, uintptr_t wd,
                               int nesterov, float scale, uintptr_t s) {
    if (p.size() != g.size() || p.size() != n.size() ||
        (!buf.empty() && buf.size() != p.size()) ||
        (!master.empty() && master.size() != p.size()))
      throw std::invalid_argument("sgd_update_multi: list lengths differ");
    check(sa_sgd_update_multi(
              dt, (int)p.size(), reinterpret_cast<void *const *>(p.data()),
              reinterpret_cast<const void *const *>(g.data()),
              buf.empty() ? nullptr
                          : reinterpret_cast<void *const *>(buf.data()),
              master.empty()
                  ? nullptr
                  : reinterpret_cast<float *const *>(master.data()),
              n.data(), (const float *)CP(lr), (const float *)CP(mom),
              (const float *)CP(damp), (const float *)CP(wd), nesterov, scale,
              S(s)),
          "sgd_update_multi");
  });

  // ---- index --------------------------------------------------------------------------------------
  m.def("index_select", [](int dt, uintptr_t x, uintptr_t idx, uintptr_t out,
                           int64_t outer, int64_t n_in, int64_t n_out,
                           int64_t inner, uintptr_t s) {
    check(sa_index_select(dt, CP(x), (const int *)CP(idx), P(out), outer, n_in,
                          n_out, inner, S(s)),
          "index_select");
  });
  m.def("index_add", [](int dt, uintptr_t dy, uintptr_t idx, uintptr_t out,
                        int64_t outer, int64_t n_in, int64_t n_out,
                        int64_t inner, uintptr_t s) {
    check(sa_index_add(dt, CP(dy), (const int *)CP(idx), P(out), outer, n_in,
                       n_out, inner, S(s)),
          "index_add");
  });

  // ---- recurrent cells --------------------------------------------------------------------------------
  m.def("lstm_cell_fwd", [](int dt, uintptr_t gates, uintptr_t c_prev,
                            uintptr_t c, uintptr_t h, uintptr_t acts,
                            int64_t batch, int64_t hidden, uintptr_t s) {
    check(sa_lstm_cell_fwd(dt, CP(gates), CP(c_prev), P(c), P(h), P(acts),
                           batch, hidden, S(s)),
          "lstm_cell_fwd");
  });
  m.def("lstm_cell_bwd", [](int dt, uintptr_t dh, uintptr_t dc_next,
                            uintptr_t acts, uintptr_t c_prev, uintptr_t c,
                            uintptr_t dgates, uintptr_t dc_prev, int64_t batch,
                            int64_t hidden, uintptr_t s) {
    check(sa_lstm_cell_bwd(dt, CP(dh), CP(dc_next), CP(acts), CP(c_prev),
                           CP(c), P(dgates), P(dc_prev), batch, hidden, S(s)),
          "lstm_cell_bwd");
  });

  // ---- Winograd transforms -----------------------------------------------------------------------------
  m.def("wino_filter", [](int dt, int mm, uintptr_t w, uintptr_t U, int K,
                          int C, bool flip, uintptr_t s) {
    check(sa_wino_filter(dt, mm, CP(w), P(U), K, C, flip ? 1 : 0, S(s)),
          "wino_filter");
  });
  m.def("wino_input", [](int dt, int mm, uintptr_t x, uintptr_t V, int N,
                         int C, int H, int W, int slices, uintptr_t s) {
    check(sa_wino_input(dt, mm, CP(x), P(V), N, C, H, W, slices, S(s)),
          "wino_input");
  });
  m.def("wino_prepare", [](int dt, int mm, uintptr_t x, uintptr_t V, int N,
                           int CI, int H, int W, uintptr_t w, uintptr_t U,
                           int K, int C, bool flip, uintptr_t s) {
    check(sa_wino_prepare(dt, mm, CP(x), P(V), N, CI, H, W, CP(w), P(U), K, C,
                          flip ? 1 : 0, S(s)), "wino_prepare");
  });
  m.def("wino_prepare_grad", [](int dt, int mm, uintptr_t x, uintptr_t V,
                                uintptr_t dy, uintptr_t dM, int N, int C,
                                int K, int H, int W, int slices,
                                uintptr_t s) {
    check(sa_wino_prepare_grad(dt, mm, CP(x), P(V), CP(dy), P(dM), N, C, K, H,
                               W, slices, S(s)), "wino_prepare_grad");
  });
  m.def("wino_outgrad", [](int dt, int mm, uintptr_t dy, uintptr_t dM, int N,
                           int K, int H, int W, int slices, uintptr_t s) {
    check(sa_wino_outgrad(dt, mm, CP(dy), P(dM), N, K, H, W, slices, S(s)),
          "wino_outgrad");
  });
  m.def("wino_filtergrad", [](int dt, int mm, uintptr_t dU, bool du_float,
                              uintptr_t dw, int K, int C, int slices,
                              uintptr_t s) {
    check(sa_wino_filtergrad(dt, mm, CP(dU), du_float ? 1 : 0, P(dw), K, C,
                             slices, S(s)), "wino_filtergrad");
  });
  m.def("wino_output", [](int dt, int mm, uintptr_t M, uintptr_t bias,
                          uintptr_t addend, uintptr_t y, int N, int K, int H,
                          int W, uintptr_t s) {
    check(sa_wino_output(dt, mm, CP(M), CP(bias), CP(addend), P(y), N, K, H,
                         W, S(s)), "wino_output");
  });

  // ---- xGMI windows: direct collectives between the GPUs of a node ----------------------------------
  m.attr("xgmi_max_ranks") = sa::kXgmiMaxRanks;
  m.def("xgmi_signal_bytes", []() { return sa_xgmi_signal_bytes(); });
  m.def("xgmi_one_shot_max_bytes", []() { return sa_xgmi_one_shot_max_bytes(); });
  m.def("xgmi_alloc", [](int64_t bytes) {
    void *p = nullptr;
    check(sa_xgmi_alloc(bytes, &p), "xgmi_alloc");
    return reinterpret_cast<uintptr_t>(p);
  }, "uncached device memory, zero-filled");
  m.def("xgmi_free", [](uintptr_t p) { check(sa_xgmi_free(P(p)), "xgmi_free"); });
  m.def("xgmi_export", [](uintptr_t p) {
    char h[64];
    check(sa_xgmi_export(P(p), h), "xgmi_export");
    return py::bytes(h, 64);
  }, "IPC handle of an allocation (64 bytes)");
  m.def("xgmi_import", [](std::string handle) {
    if (handle.size() != 64)
      throw std::invalid_argument("xgmi_import: a handle has 64 bytes");
    void *p = nullptr;
    check(sa_xgmi_import(handle.data(), &p), "xgmi_import");
    return reinterpret_cast<uintptr_t>(p);
  });
  m.def("xgmi_close", [](uintptr_t p) { check(sa_xgmi_close(P(p)), "xgmi_close"); });
  m.def("xgmi_error_word", []() {
    void *h = nullptr, *d = nullptr;
    check(sa_xgmi_error_word(&h, &d), "xgmi_error_word");
    return py::make_tuple(reinterpret_cast<uintptr_t>(h),
                          reinterpret_cast<uintptr_t>(d));
  }, "(host address, device address) of a zeroed word in mapped host memory");
  m.def("xgmi_error_word_free", [](uintptr_t h) {
    check(sa_xgmi_error_word_free(P(h)), "xgmi_error_word_free");
  });
  m.def("xgmi_read_word", [](uintptr_t h) {
    return (int)*reinterpret_cast<volatile unsigned *>(h);
  });
  m.def("xgmi_clear_word", [](uintptr_t h) {
    *reinterpret_cast<volatile unsigned *>(h) = 0u;
  });
  auto make_peers = [](const std::vector<uintptr_t> &data,
                       const std::vector<uintptr_t> &flags, int rank,
                       uintptr_t error) {
    if (data.size() != flags.size() || data.empty() ||
        data.size() > (size_t)sa::kXgmiMaxRanks || rank < 0 ||
        rank >= (int)data.size())
      throw std::invalid_argument("xgmi: bad peer lists");
    sa::XgmiPeers pr{};
    for (size_t i = 0; i < data.size(); ++i) {
      pr.data[i] = reinterpret_cast<char *>(data[i]);
      pr.flag[i] = reinterpret_cast<unsigned *>(flags[i]);
    }
    pr.epoch = pr.flag[rank] + sa::kXgmiMaxBlocks * sa::kXgmiMaxRanks;
    pr.error = reinterpret_cast<unsigned *>(error);
    return pr;
  };
  m.def("xgmi_allreduce", [make_peers](int dt, std::vector<uintptr_t> data,
                                       std::vector<uintptr_t> flags, int rank,
                                       uintptr_t error, int64_t off,
                                       int64_t bytes, bool one_shot,
                                       double timeout_s, uintptr_t s) {
    sa::XgmiPeers pr = make_peers(data, flags, rank, error);
    check(sa_xgmi_allreduce(dt, &pr, rank, (int)data.size(), off, bytes,
                            one_shot ? 1 : 0, timeout_s, S(s)),
          "xgmi_allreduce");
  });
  m.def("xgmi_allgather", [make_peers](std::vector<uintptr_t> data,
                                       std::vector<uintptr_t> flags, int rank,
                                       uintptr_t error, int64_t off,
                                       int64_t gathered_off, int64_t bytes,
                                       double timeout_s, uintptr_t s) {
    sa::XgmiPeers pr = make_peers(data, flags, rank, error);
    check(sa_xgmi_allgather(&pr, rank, (int)data.size(), off, gathered_off,
                            bytes, timeout_s, S(s)),
          "xgmi_allgather");
  });

  m.def("xgmi_alltoall", [make_peers](std::vector<uintptr_t> data,
                                      std::vector<uintptr_t> flags, int rank,
                                      uintptr_t error, int64_t off,
                                      int64_t gathered_off,
                                      int64_t chunk_bytes, double timeout_s,
                                      uintptr_t s) {
    sa::XgmiPeers pr = make_peers(data, flags, rank, error);
    check(sa_xgmi_alltoall(&pr, rank, (int)data.size(), off, gathered_off,
                           chunk_bytes, timeout_s, S(s)),
          "xgmi_alltoall");
  });
  m.def("xgmi_reduce_scatter", [make_peers](int dt, std::vector<uintptr_t> data,
                                            std::vector<uintptr_t> flags,
                                            int rank, uintptr_t error,
                                            int64_t off, int64_t bytes,
                                            double timeout_s, uintptr_t s) {
    sa::XgmiPeers pr = make_peers(data, flags, rank, error);
    check(sa_xgmi_reduce_scatter(dt, &pr, rank, (int)data.size(), off, bytes,
                                 timeout_s, S(s)),
          "xgmi_reduce_scatter");
  });

  // ---- communication helpers ----------------------------------------------------------------------------
  m.def("pack", [](int src_dt, int dst_dt, std::vector<uintptr_t> ptrs,
                   std::vector<int64_t> counts, uintptr_t bucket, uintptr_t s) {
    if (ptrs.size() != counts.size())
      throw std::invalid_argument("pack: list lengths differ");
    check(sa_pack(src_dt, dst_dt, (int)ptrs.size(),
                  reinterpret_cast<void *const *>(ptrs.data()), counts.data(),
                  P(bucket), S(s)),
          "pack");
  });
  m.def("unpack", [](int src_dt, int dst_dt, std::vector<uintptr_t> ptrs,
                     std::vector<int64_t> counts, uintptr_t bucket, float scale,
                     uintptr_t s) {
    if (ptrs.size() != counts.size())
      throw std::invalid_argument("unpack: list lengths differ");
    check(sa_unpack(src_dt, dst_dt, (int)ptrs.size(),
                    reinterpret_cast<void *const *>(ptrs.data()),
                    counts.data(), CP(bucket), scale, S(s)),
          "unpack");
  });
  m.def("sparsify_threshold", [](uintptr_t x, float thr, uintptr_t vals,
                                 uintptr_t idx, uintptr_t count, int64_t n,
                                 int64_t cap, uintptr_t s) {
    check(sa_sparsify_threshold((const float *)CP(x), thr, (float *)P(vals),
                                (int *)P(idx), (int *)P(count), n, cap, S(s)),
          "sparsify_threshold");
  });
  m.def("topk_abs_workspace", &sa_topk_abs_workspace);
  m.def("topk_abs", [](uintptr_t x, int64_t n, int64_t k, uintptr_t vals,
                       uintptr_t idx, uintptr_t work, uintptr_t s) {
    check(sa_topk_abs((const float *)CP(x), n, k, (float *)P(vals),
                      (int *)P(idx), (unsigned *)P(work), S(s)),
          "topk_abs");
  });
  m.def("scatter_add_sparse", [](uintptr_t vals, uintptr_t idx, uintptr_t count,
                                 int64_t max_count, uintptr_t dense, int64_t n,
                                 uintptr_t s) {
    check(sa_scatter_add_sparse((const float *)CP(vals), (const int *)CP(idx),
                                (const int *)CP(count), max_count,
                                (float *)P(dense), n, S(s)),
          "scatter_add_sparse");
  });
  m.def("sparse_select", [](uintptr_t x, int64_t n, float thr, int topk,
                            int64_t k, uintptr_t payload, int64_t cap,
                            uintptr_t work, uintptr_t s) {
    check(sa_sparse_select((const float *)CP(x), n, thr, topk, k,
                           (int *)P(payload), cap, (unsigned *)P(work), S(s)),
          "sparse_select");
  });
  m.def("scatter_add_segments", [](uintptr_t payloads, int nseg, int64_t cap,
                                   float sign, uintptr_t dense, int64_t n,
                                   uintptr_t s) {
    check(sa_scatter_add_segments((const int *)CP(payloads), nseg, cap, sign,
                                  (float *)P(dense), n, S(s)),
          "scatter_add_segments");
  });

  m.def("adaptive_update_multi", [](int dt, int rule, std::vector<uintptr_t> p,
                                    std::vector<uintptr_t> g,
                                    std::vector<uintptr_t> s1,
                                    std::vector<uintptr_t> s2,
                                    std::vector<int64_t> n, uintptr_t lr,
                                    uintptr_t wd, uintptr_t eps, uintptr_t a,
                                    uintptr_t b, uintptr_t step, float scale,
                                    uintptr_t s) {
    if (p.size() != g.size() || p.size() != s1.size() ||
        p.size() != n.size() || (!s2.empty() && s2.size() != p.size()))
      throw std::invalid_argument("adaptive_update_multi: list lengths differ");
    check(sa_adaptive_update_multi(
              dt, rule, (int)p.size(),
              reinterpret_cast<void *const *>(p.data()),
              reinterpret_cast<const void *const *>(g.data()),
              reinterpret_cast<void *const *>(s1.data()),
              s2.empty() ? nullptr
                         : reinterpret_cast<void *const *>(s2.data()),
              n.data(), (const float *)CP(lr), (const float *)CP(wd),
              (const float *)CP(eps), (const float *)CP(a),
              (const float *)CP(b), (const float *)CP(step), scale, S(s)),
          "adaptive_update_multi");
  });

  // ---- recurrent cells and the persistent LSTM ----------------------------------------------------------
  m.def("gru_cell_fwd", [](int dt, uintptr_t gx, uintptr_t gh, uintptr_t brn,
                           uintptr_t hp, uintptr_t h, uintptr_t acts,
                           uintptr_t ghn, int64_t batch, int64_t hidden,
                           uintptr_t s) {
    check(sa_gru_cell_fwd(dt, CP(gx), CP(gh), CP(brn), CP(hp), P(h), P(acts),
                          P(ghn), batch, hidden, S(s)),
          "gru_cell_fwd");
  });
  m.def("gru_cell_bwd", [](int dt, uintptr_t dh, uintptr_t acts, uintptr_t ghn,
                           uintptr_t hp, uintptr_t dgx, uintptr_t dgh,
                           uintptr_t dhp, int64_t batch, int64_t hidden,
                           uintptr_t s) {
    check(sa_gru_cell_bwd(dt, CP(dh), CP(acts), CP(ghn), CP(hp), P(dgx), P(dgh),
                          P(dhp), batch, hidden, S(s)),
          "gru_cell_bwd");
  });
  m.def("rnn_act_bwd", [](int dt, int mode, uintptr_t dh, uintptr_t h,
                          uintptr_t d, int64_t total, uintptr_t s) {
    check(sa_rnn_act_bwd(dt, mode, CP(dh), CP(h), P(d), total, S(s)),
          "rnn_act_bwd");
  });
  m.def("lstm_seq_supported", [](int64_t batch, int64_t hidden) {
    return sa_lstm_seq_supported(batch, hidden) != 0;
  });
  m.def("lstm_seq_fwd", [](uintptr_t gx, uintptr_t R, uintptr_t h0,
                           uintptr_t c0, uintptr_t hseq, uintptr_t cseq,
                           uintptr_t acts, uintptr_t bar, int batch, int hidden,
                           int steps, int reverse, uintptr_t s) {
    LstmSeqArgs a{};
    a.gx = (const float *)CP(gx); a.R = (const float *)CP(R);
    a.h0 = (const float *)CP(h0); a.c0 = (const float *)CP(c0);
    a.hseq = (float *)P(hseq); a.cseq = (float *)P(cseq);
    a.acts = (float *)P(acts); a.bar = (unsigned *)P(bar);
    a.batch = batch; a.hidden = hidden; a.steps = steps; a.reverse = reverse;
    check(sa_lstm_seq(&a, 0, S(s)), "lstm_seq_fwd");
  });
  m.def("lstm_seq_bwd", [](uintptr_t R, uintptr_t c0, uintptr_t cseq,
                           uintptr_t acts, uintptr_t dy, int64_t dy_stride,
                           uintptr_t dhy, uintptr_t dcy, uintptr_t dgates,
                           uintptr_t dhx, uintptr_t dcx, uintptr_t slabs,
                           uintptr_t bar, int batch, int hidden, int steps,
                           int reverse, uintptr_t s) {
    LstmSeqArgs a{};
    a.R = (const float *)CP(R); a.c0 = (const float *)CP(c0);
    a.cseq = (float *)P(cseq); a.acts = (float *)P(acts);
    a.dy = (const float *)CP(dy); a.dy_stride = dy_stride;
    a.dhy = (const float *)CP(dhy); a.dcy = (const float *)CP(dcy);
    a.dgates = (float *)P(dgates); a.dhx = (float *)P(dhx);
    a.dcx = (float *)P(dcx); a.slabs = (float *)P(slabs);
    a.bar = (unsigned *)P(bar);
    a.batch = batch; a.hidden = hidden; a.steps = steps; a.reverse = reverse;
    check(sa_lstm_seq(&a, 1, S(s)), "lstm_seq_bwd");
  });

  // ---- fused attention ----------------------------------------------------------------------------------
  py::class_<AttnDesc>(m, "AttnDesc")
      .def(py::init([](uintptr_t q, uintptr_t k, uintptr_t v, uintptr_t mask,
                       std::vector<int64_t> qs, std::vector<int64_t> ks,
                       std::vector<int64_t> vs, std::vector<int64_t> ms, int B,
                       int H, int Sq, int Sk, int D, float scale) {
        if (qs.size() != 3 || ks.size() != 3 || vs.size() != 3 ||
            ms.size() != 4)
          throw std::invalid_argument("AttnDesc: stride list lengths");
        AttnDesc d{};
        d.q = CP(q); d.k = CP(k); d.v = CP(v); d.mask = CP(mask);
        for (int i = 0; i < 3; ++i) {
          d.q_stride[i] = qs[i]; d.k_stride[i] = ks[i]; d.v_stride[i] = vs[i];
        }
        for (int i = 0; i < 4; ++i) d.m_stride[i] = ms[i];
        d.B = B; d.H = H; d.Sq = Sq; d.Sk = Sk; d.D = D; d.scale = scale;
        return d;
      }));
  m.def("attention_fwd", [](int dt, const AttnDesc &d, uintptr_t out,
                            uintptr_t lse, uintptr_t s) {
    check(sa_attention_fwd(dt, &d, P(out), (float *)P(lse), S(s)),
          "attention_fwd");
  });
  m.def("attention_bwd", [](int dt, const AttnDesc &d, uintptr_t out,
                            uintptr_t dout, uintptr_t lse, uintptr_t delta,
                            uintptr_t dq, uintptr_t dk, uintptr_t dv,
                            uintptr_t s) {
    check(sa_attention_bwd(dt, &d, CP(out), CP(dout), (const float *)CP(lse),
                           (float *)P(delta), P(dq), P(dk), P(dv), S(s)),
          "attention_bwd");
  });

  // ---- device-side time stamps ---------------------------------------------------------------------------
  m.def("stamp_open", [](uintptr_t slots, int i, uintptr_t s) {
    check(sa_stamp_open((unsigned long long *)P(slots), i, S(s)), "stamp_open");
  });
  m.def("stamp_close", [](uintptr_t slots, int open, uintptr_t acc, int node,
                          uintptr_t gate, float sec_per_tick, uintptr_t s) {
    check(sa_stamp_close((const unsigned long long *)CP(slots), open,
                         (float *)P(acc), node, (const int *)CP(gate),
                         sec_per_tick, S(s)),
          "stamp_close");
  });
  m.def("wall_clock_hz", &sa_wall_clock_hz);
}
