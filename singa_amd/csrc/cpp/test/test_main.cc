// Self-checking test program of the C++ API (no external test framework).
//   singa_cpp_test            host device only
//   singa_cpp_test --gpu      every numeric test also on HipGPU 0, compared
//                             with the host result of the same operation
//   singa_cpp_test --opencl   the same on OpenclDevice 0
// Modelled on the reference's test/singa/test_*.cc suites (tensor math,
// layers, optimizers, loss, metric, snapshot, feed-forward net, endpoint).
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "singa/singa.h"
#include "singa/io/jpeg.h"

using namespace singa;

static int g_failed = 0, g_checks = 0;
static bool g_gpu = false;
static std::shared_ptr<Device> g_dev;

#define EXPECT(cond)                                                       \
  do {                                                                     \
    ++g_checks;                                                            \
    if (!(cond)) {                                                         \
      ++g_failed;                                                          \
      std::printf("  FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);      \
    }                                                                      \
  } while (0)
#define EXPECT_NEAR(a, b, tol)                                             \
  do {                                                                     \
    ++g_checks;                                                            \
    double a_ = (a), b_ = (b);                                             \
    if (!(std::fabs(a_ - b_) <= (tol) * (1.0 + std::fabs(b_)))) {          \
      ++g_failed;                                                          \
      std::printf("  FAILED %s:%d  %s = %g vs %s = %g\n", __FILE__,        \
                  __LINE__, #a, a_, #b, b_);                               \
    }                                                                      \
  } while (0)

static Tensor FromVec(const std::vector<float> &v, const Shape &s,
                      std::shared_ptr<Device> dev) {
  Tensor t(s, dev);
  t.CopyDataFromHostPtr(v.data(), v.size());
  return t;
}
static Tensor FromInts(const std::vector<int> &v, const Shape &s,
                       std::shared_ptr<Device> dev) {
  Tensor t(s, dev, kInt);
  t.CopyDataFromHostPtr(v.data(), v.size());
  return t;
}
static std::vector<float> ToVec(const Tensor &t) {
  return t.Contiguous().Clone(defaultDevice).ToVector<float>();
}
static std::vector<float> Rand(size_t n, unsigned seed, float scale = 1.f) {
  std::mt19937 g(seed);
  std::normal_distribution<float> d(0.f, scale);
  std::vector<float> v(n);
  for (auto &x : v) x = d(g);
  return v;
}
static void ExpectVecNear(const std::vector<float> &a,
                          const std::vector<float> &b, double tol,
                          const char *what) {
  ++g_checks;
  if (a.size() != b.size()) {
    ++g_failed;
    std::printf("  FAILED %s: size %zu vs %zu\n", what, a.size(), b.size());
    return;
  }
  double worst = 0;
  for (size_t i = 0; i < a.size(); ++i)
    worst = std::max(worst, std::fabs(double(a[i]) - b[i]) / (1.0 + std::fabs(b[i])));
  if (!(worst <= tol)) {
    ++g_failed;
    std::printf("  FAILED %s: max rel diff %g > %g\n", what, worst, tol);
  }
}

// ---- tests that run per device ------------------------------------------------------
static void TestTensorMath(std::shared_ptr<Device> dev) {
  auto a = Rand(12, 1), b = Rand(12, 2);
  Tensor A = FromVec(a, {3, 4}, dev), B = FromVec(b, {3, 4}, dev);
  auto sum = ToVec(A + B), prod = ToVec(A * B);
  for (int i = 0; i < 12; ++i) {
    EXPECT_NEAR(sum[i], a[i] + b[i], 1e-6);
    EXPECT_NEAR(prod[i], a[i] * b[i], 1e-6);
  }
  auto r = ToVec(ReLU(A));
  auto s = ToVec(Sigmoid(A));
  for (int i = 0; i < 12; ++i) {
    EXPECT_NEAR(r[i], std::max(a[i], 0.f), 1e-6);
    EXPECT_NEAR(s[i], 1.f / (1.f + std::exp(-a[i])), 1e-5);
  }
  auto sc = ToVec(A * 2.5f + 1.f);
  EXPECT_NEAR(sc[5], a[5] * 2.5f + 1.f, 1e-6);
  auto lt = ToVec(A < 0.f);
  EXPECT_NEAR(lt[3], a[3] < 0 ? 1.f : 0.f, 0);
  // GEMM incl. transposed views
  Tensor C = Mult(A, B.T());  // (3,4)x(4,3)
  auto c = ToVec(C);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float ref = 0;
      for (int k = 0; k < 4; ++k) ref += a[i * 4 + k] * b[j * 4 + k];
      EXPECT_NEAR(c[i * 3 + j], ref, 1e-5);
    }
  Tensor D = Mult(A.T(), B);  // (4,3)x(3,4)
  auto d = ToVec(D);
  float ref = 0;
  for (int k = 0; k < 3; ++k) ref += a[k * 4 + 1] * b[k * 4 + 2];
  EXPECT_NEAR(d[1 * 4 + 2], ref, 1e-5);
  // row / column helpers and reductions
  Tensor v = FromVec({1, 2, 3, 4}, {4}, dev);
  Tensor M = A.Clone();
  AddRow(v, &M);
  EXPECT_NEAR(ToVec(M)[6], a[6] + 3, 1e-6);
  Tensor col = FromVec({10, 20, 30}, {3}, dev);
  MultColumn(col, &M);
  EXPECT_NEAR(ToVec(M)[6], (a[6] + 3) * 20, 1e-5);
  Tensor sr(Shape{4}, dev), scol(Shape{3}, dev);
  SumRows(A, &sr);
  SumColumns(A, &scol);
  EXPECT_NEAR(ToVec(sr)[1], a[1] + a[5] + a[9], 1e-5);
  EXPECT_NEAR(ToVec(scol)[2], a[8] + a[9] + a[10] + a[11], 1e-5);
  float total = 0, l1 = 0, l2 = 0;
  for (float x : a) { total += x; l1 += std::fabs(x); l2 += x * x; }
  EXPECT_NEAR(Sum<float>(A), total, 1e-5);
  EXPECT_NEAR(A.L1(), l1 / 12, 1e-5);
  EXPECT_NEAR(A.L2(), std::sqrt(l2) / 12, 1e-5);
  // softmax rows sum to one; cross entropy
  Tensor P = SoftMax(A);
  auto p = ToVec(P);
  EXPECT_NEAR(p[0] + p[1] + p[2] + p[3], 1.0, 1e-5);
  Tensor t = FromInts({1, 0, 3}, {3}, dev);
  Tensor loss(Shape{3}, dev);
  ComputeCrossEntropy(P, t, &loss);
  EXPECT_NEAR(ToVec(loss)[2], -std::log(p[2 * 4 + 3]), 1e-5);
  SoftmaxCrossEntropyBwd(t, &P);
  EXPECT_NEAR(ToVec(P)[1], p[1] - 1.f, 1e-6);
  // layout
  Tensor cat = ConcatenateColumns({A, B});
  EXPECT(cat.shape(1) == 8);
  EXPECT_NEAR(ToVec(cat)[8 + 5], b[4 + 1], 1e-6);
  Tensor sl = SliceOn(cat, 2, 6, 1);
  EXPECT_NEAR(ToVec(sl)[0], a[2], 1e-6);
  EXPECT_NEAR(ToVec(sl)[2], b[0], 1e-6);
  Tensor rows = CopyRows(A, 1, 3);
  EXPECT_NEAR(ToVec(rows)[0], a[4], 1e-6);
  // casts and random
  Tensor I = A.Clone();
  I.SetValue(3.7f);
  I.ToType(kInt);
  EXPECT(I.Clone(defaultDevice).ToVector<int>()[0] == 3);
  Tensor u(Shape{4096}, dev);
  Uniform(0.f, 1.f, &u);
  float mean = Sum<float>(u) / 4096;
  EXPECT(mean > 0.45 && mean < 0.55);
  Gaussian(1.f, 2.f, &u);
  mean = Sum<float>(u) / 4096;
  EXPECT(mean > 0.8 && mean < 1.2);
  Bernoulli(0.3f, &u);
  mean = Sum<float>(u) / 4096;
  EXPECT(mean > 0.25 && mean < 0.35);
}

// Parts of the OpenCL device that the shared tests do not reach: the memory
// interface, every unary code, int tensors, strided / broadcast operands, a
// GEMM larger than one tile with alpha / beta and transposes, reductions
// longer than a work-group, convolution / pooling geometries with stride,
// padding and groups, all against the host device.
// The part of the reference's tensor function set beyond the layers' needs:
// the remaining element-wise functions, comparisons, views (transpose,
// broadcast), repeats and the axis forms of soft-max, against loops on the
// host.
static void TestTensorFunctions(std::shared_ptr<Device> dev) {
  const std::vector<float> x = {-1.75f, -0.5f, 0.f, 0.25f, 0.5f, 0.9f,
                                1.5f, 2.5f, -2.5f, 3.2f, -0.3f, 0.75f};
  Tensor X = FromVec(x, {3, 4}, dev);
  struct Case { const char *name; Tensor (*fn)(const Tensor &);
                double (*ref)(double); bool unit; };
  const Case cases[] = {
      {"Erf", Erf, [](double v) { return std::erf(v); }, false},
      {"Ceil", Ceil, [](double v) { return std::ceil(v); }, false},
      {"Floor", Floor, [](double v) { return std::floor(v); }, false},
      {"Round", Round, [](double v) { return std::round(v); }, false},
      {"RoundE", RoundE, [](double v) { return std::nearbyint(v); }, false},
      {"SoftPlus", SoftPlus, [](double v) { return std::log1p(std::exp(v)); },
       false},
      {"SoftSign", SoftSign, [](double v) { return v / (1 + std::fabs(v)); },
       false},
      {"Cos", Cos, [](double v) { return std::cos(v); }, false},
      {"Cosh", Cosh, [](double v) { return std::cosh(v); }, false},
      {"Sin", Sin, [](double v) { return std::sin(v); }, false},
      {"Sinh", Sinh, [](double v) { return std::sinh(v); }, false},
      {"Tan", Tan, [](double v) { return std::tan(v); }, false},
      {"Atan", Atan, [](double v) { return std::atan(v); }, false},
      {"Asinh", Asinh, [](double v) { return std::asinh(v); }, false},
      // domain [-1, 1]: the input is squashed first
      {"Acos", Acos, [](double v) { return std::acos(v); }, true},
      {"Asin", Asin, [](double v) { return std::asin(v); }, true},
      {"Atanh", Atanh, [](double v) { return std::atanh(v); }, true},
  };
  for (const Case &c : cases) {
    std::vector<float> in = x, want(x.size());
    if (c.unit)
      for (auto &v : in) v = 0.95f * v / (1.f + std::fabs(v));
    for (size_t i = 0; i < in.size(); ++i) want[i] = (float)c.ref(in[i]);
    Tensor I = FromVec(in, {3, 4}, dev);
    ExpectVecNear(ToVec(c.fn(I)), want, 2e-5, c.name);
  }
  {  // Acosh wants x >= 1
    std::vector<float> in = x, want(x.size());
    for (auto &v : in) v = 1.f + std::fabs(v);
    for (size_t i = 0; i < in.size(); ++i) want[i] = std::acosh(in[i]);
    ExpectVecNear(ToVec(Acosh(FromVec(in, {3, 4}, dev))), want, 2e-5, "Acosh");
  }
  // comparisons and powers
  auto y = Rand(12, 31);
  Tensor Y = FromVec(y, {3, 4}, dev);
  std::vector<float> le(12), ge(12), eq(12), eqs(12), pw(12);
  for (int i = 0; i < 12; ++i) {
    le[i] = x[i] <= y[i];
    ge[i] = x[i] >= y[i];
    eq[i] = x[i] == x[i];
    eqs[i] = x[i] == 0.5f;
    pw[i] = std::pow(std::fabs(x[i]) + 0.5f, y[i]);
  }
  ExpectVecNear(ToVec(X <= Y), le, 0, "<=");
  ExpectVecNear(ToVec(X >= Y), ge, 0, ">=");
  ExpectVecNear(ToVec(X == X), eq, 0, "==");
  ExpectVecNear(ToVec(X == 0.5f), eqs, 0, "== scalar");
  Tensor out(Shape{3, 4}, dev);
  LT(X, Y, &out);
  GT(Y, X, &out);
  LE(X, 0.25f, &out);
  EXPECT_NEAR(Sum<float>(out), 6.0, 0);
  ExpectVecNear(ToVec(Pow(Abs(X) + 0.5f, Y)), pw, 1e-5, "Pow(tensor, tensor)");
  Pow(X, 2.f, &out);
  ExpectVecNear(ToVec(out), ToVec(Square(X)), 1e-6, "Pow(x, 2, out)");
  Sub(X, 1.f, &out);
  EXPECT_NEAR(ToVec(out)[0], -2.75, 1e-6);
  Div(X, 2.f, &out);
  EXPECT_NEAR(ToVec(out)[6], 0.75, 1e-6);
  Div(3.f, Abs(X) + 1.f, &out);
  EXPECT_NEAR(ToVec(out)[0], 3.0 / 2.75, 1e-6);
  ReLUBackward(Y, X, &out);
  EXPECT_NEAR(ToVec(out)[0], 0.0, 0);
  EXPECT_NEAR(ToVec(out)[3], y[3], 0);
  // views: transposes, broadcast, Transform, conversions
  auto z = Rand(24, 32);
  Tensor Z = FromVec(z, {2, 3, 4}, dev);
  Tensor P = Transpose(Z, {2, 0, 1});
  EXPECT(P.shape() == (Shape{4, 2, 3}) && P.transpose());
  auto pv = ToVec(Transform(P));
  bool ok = true;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 2; ++b)
      for (int c = 0; c < 3; ++c)
        ok = ok && pv[(a * 2 + b) * 3 + c] == z[(b * 3 + c) * 4 + a];
  EXPECT(ok);
  Tensor R = Transpose(Z);
  EXPECT(R.shape() == (Shape{4, 3, 2}));
  EXPECT_NEAR(ToVec(R)[1], z[12], 0);     // (0, 0, 1) <- (1, 0, 0)
  Tensor row = FromVec({1.f, 2.f, 3.f, 4.f}, {1, 4}, dev);
  Tensor Bc = Broadcast(row, Shape{2, 3, 4});
  EXPECT(Bc.shape() == (Shape{2, 3, 4}) && Bc.stride()[0] == 0 &&
         Bc.stride()[1] == 0);
  auto bv = ToVec(Contiguous(Bc));
  EXPECT(bv.size() == 24 && bv[0] == 1.f && bv[7] == 4.f && bv[20] == 1.f);
  ExpectVecNear(ToVec(Z + Contiguous(Bc)),
                [&] { auto w = z; for (int i = 0; i < 24; ++i) w[i] += 1 + i % 4;
                      return w; }(), 1e-6, "broadcast add");
  Tensor col = FromVec({5.f, 6.f}, {2, 1}, dev);
  Tensor Bm = Broadcast(col, Shape{7, 7}, 2);    // matrix axes left alone
  EXPECT(Bm.shape() == (Shape{2, 1}));
  Tensor I = X.AsType(kInt);
  EXPECT(I.data_type() == kInt && X.data_type() == kFloat32);
  EXPECT(I.ToVector<int>()[7] == 2 && I.ToVector<int>()[0] == -1);
  I.ToType(kFloat32);
  EXPECT(I.data_type() == kFloat32);
  EXPECT_NEAR(ToVec(I)[9], 3.0, 0);
  // repeats
  Tensor M = FromVec({1, 2, 3, 4, 5, 6}, {2, 3}, dev);
  ExpectVecNear(ToVec(M.Repeat({2}, 0)),
                {1, 2, 3, 1, 2, 3, 4, 5, 6, 4, 5, 6}, 0, "Repeat rows");
  ExpectVecNear(ToVec(M.Repeat({1, 0, 2}, 1)), {1, 3, 3, 4, 6, 6}, 0,
                "Repeat columns, one count per slice");
  Tensor flat = M.Repeat({2}, Noaxis);
  EXPECT(flat.shape() == (Shape{12}));
  ExpectVecNear(ToVec(flat), {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6}, 0,
                "Repeat flattened");
  // sums, scaled row / column adds, device-side axpy
  EXPECT_NEAR(ToVec(SumAll(M))[0], 21.0, 1e-6);
  Tensor M2 = M.Clone();
  AddColumn(2.f, 0.5f, FromVec({10, 20}, {2}, dev), &M2);
  ExpectVecNear(ToVec(M2), {20.5f, 21.f, 21.5f, 42.f, 42.5f, 43.f}, 1e-6,
                "AddColumn(alpha, beta)");
  M2 = M.Clone();
  AddRow(-1.f, 1.f, FromVec({1, 2, 3}, {3}, dev), &M2);
  ExpectVecNear(ToVec(M2), {0, 0, 0, 3, 3, 3}, 1e-6, "AddRow(alpha, beta)");
  M2 = M.Clone();
  Axpy(FromVec({-2.f}, {1}, dev), M, &M2);
  ExpectVecNear(ToVec(M2), {-1, -2, -3, -4, -5, -6}, 1e-6, "Axpy(tensor)");
  // soft-max over trailing axes and its gradient
  Tensor S = SoftMax(Z, 1);                   // 2 rows of 12
  auto sv = ToVec(S);
  double s0 = 0, s1 = 0;
  for (int i = 0; i < 12; ++i) { s0 += sv[i]; s1 += sv[12 + i]; }
  EXPECT_NEAR(s0, 1.0, 1e-5);
  EXPECT_NEAR(s1, 1.0, 1e-5);
  EXPECT(S.shape() == Z.shape());
  Tensor G = SoftMaxBackward(Z, 1, S);
  auto gv = ToVec(G);
  double dot = 0;
  for (int i = 0; i < 12; ++i) dot += z[i] * sv[i];
  EXPECT_NEAR(gv[5], sv[5] * (z[5] - dot), 1e-5);
  // losses returning their result, slices and concatenations
  Tensor prob = SoftMax(FromVec(Rand(8, 33), {2, 4}, dev));
  Tensor lab = FromInts({3, 1}, {2}, dev);
  auto lv = ToVec(CrossEntropyFwd(prob, lab));
  auto pr = ToVec(prob);
  EXPECT_NEAR(lv[0], -std::log(pr[3]), 1e-5);
  EXPECT_NEAR(lv[1], -std::log(pr[5]), 1e-5);
  auto gb = ToVec(SoftmaxCrossEntropyBwd(prob, lab));
  EXPECT_NEAR(gb[3], pr[3] - 1, 1e-6);
  EXPECT_NEAR(gb[4], pr[4], 1e-6);
  EXPECT_NEAR(ToVec(prob)[3], pr[3], 0);      // the argument is left alone
  ExpectVecNear(ToVec(SliceColumns(M, 1, 3)), {2, 3, 5, 6}, 0, "SliceColumns");
  ExpectVecNear(ToVec(ConcatRows({M, M})),
                {1, 2, 3, 4, 5, 6, 1, 2, 3, 4, 5, 6}, 0, "ConcatRows");
  ExpectVecNear(ToVec(ConcatColumns({M, M})),
                {1, 2, 3, 1, 2, 3, 4, 5, 6, 4, 5, 6}, 0, "ConcatColumns");
  EXPECT(Resize(M, Shape{3, 2}).shape() == (Shape{3, 2}));
  EXPECT(VecToStr(std::vector<int>{1, 2, 3}) == "[1, 2, 3]");
  EXPECT(CreateConstraint("l2") != nullptr && CreateRegularizer("l2") != nullptr);
}

static void TestOpenclSpecific(std::shared_ptr<Device> cl) {
  auto *ocl_dev = static_cast<OpenclDevice *>(cl.get());
  std::printf("  %s\n", ocl_dev->Description().c_str());
  EXPECT(OpenclDevice::GetNumDevices() >= 1);
  EXPECT(cl->lang() == kOpencl);
  // memory: lazy blocks, accounting, offsets, device-to-device copies
  const size_t before = cl->GetAllocatedMem();
  {
    Tensor t(Shape{1000}, cl);
    auto v = Rand(1000, 77);
    t.CopyDataFromHostPtr(v.data(), 1000);
    EXPECT(cl->GetAllocatedMem() >= before + 4000);
    Tensor u = t.Clone();
    ExpectVecNear(ToVec(u), v, 0, "clone");
    Tensor part = CopyRows(Reshape(t, Shape{10, 100}), 3, 5);
    EXPECT_NEAR(ToVec(part)[7], v[307], 0);
    Tensor back = u.Clone(defaultDevice);
    EXPECT_NEAR(back.ToVector<float>()[999], v[999], 0);
    Tensor again = back.Clone(cl);
    EXPECT_NEAR(ToVec(again)[500], v[500], 0);
  }
  EXPECT(cl->GetAllocatedMem() == before);
  // unary functions against the host device
  auto pos = Rand(300, 5);
  for (auto &x : pos) x = std::fabs(x) + 0.1f;
  auto any = Rand(300, 6);
  Tensor P = FromVec(pos, {300}, cl), Ph = FromVec(pos, {300}, defaultDevice);
  Tensor A = FromVec(any, {300}, cl), Ah = FromVec(any, {300}, defaultDevice);
  ExpectVecNear(ToVec(Exp(A)), ToVec(Exp(Ah)), 1e-5, "exp");
  ExpectVecNear(ToVec(Log(P)), ToVec(Log(Ph)), 1e-5, "log");
  ExpectVecNear(ToVec(Sqrt(P)), ToVec(Sqrt(Ph)), 1e-5, "sqrt");
  ExpectVecNear(ToVec(Tanh(A)), ToVec(Tanh(Ah)), 1e-5, "tanh");
  ExpectVecNear(ToVec(Abs(A)), ToVec(Abs(Ah)), 0, "abs");
  ExpectVecNear(ToVec(Sign(A)), ToVec(Sign(Ah)), 0, "sign");
  ExpectVecNear(ToVec(Square(A)), ToVec(Square(Ah)), 1e-6, "square");
  ExpectVecNear(ToVec(Pow(P, 1.7f)), ToVec(Pow(Ph, 1.7f)), 1e-4, "pow");
  ExpectVecNear(ToVec(A / P), ToVec(Ah / Ph), 1e-5, "div");
  ExpectVecNear(ToVec(A - P), ToVec(Ah - Ph), 1e-6, "sub");
  ExpectVecNear(ToVec(A > P), ToVec(Ah > Ph), 0, "gt");
  // transposed operand made contiguous; reductions over 1000 columns
  auto big = Rand(7 * 1000, 8);
  Tensor B = FromVec(big, {7, 1000}, cl), Bh = FromVec(big, {7, 1000},
                                                       defaultDevice);
  ExpectVecNear(ToVec(B.T()), ToVec(Bh.T()), 0, "transpose");
  Tensor rs(Shape{7}, cl), rsh(Shape{7}, defaultDevice);
  SumColumns(B, &rs);
  SumColumns(Bh, &rsh);
  ExpectVecNear(ToVec(rs), ToVec(rsh), 1e-4, "row sums");
  Tensor cs(Shape{1000}, cl), csh(Shape{1000}, defaultDevice);
  SumRows(B, &cs);
  SumRows(Bh, &csh);
  ExpectVecNear(ToVec(cs), ToVec(csh), 1e-5, "column sums");
  EXPECT_NEAR(Sum<float>(B), Sum<float>(Bh), 1e-3);
  ExpectVecNear(ToVec(SoftMax(B)), ToVec(SoftMax(Bh)), 1e-5, "softmax");
  ExpectVecNear(ToVec(RowMax(B)), ToVec(RowMax(Bh)), 0, "row max");
  // GEMM over several tiles, with transposes and accumulation
  auto ma = Rand(130 * 90, 9), mb = Rand(90 * 70, 10), mc = Rand(130 * 70, 11);
  auto gemm_on = [&](std::shared_ptr<Device> d) {
    Tensor X = FromVec(ma, {130, 90}, d), Y = FromVec(mb, {90, 70}, d);
    Tensor Z = FromVec(mc, {130, 70}, d);
    Mult<float>(0.5f, X, Y, 2.f, &Z);
    Tensor Xt = FromVec(ma, {90, 130}, d);
    Tensor W = Mult(Xt.T(), Y);
    auto z = ToVec(Z), w = ToVec(W);
    z.insert(z.end(), w.begin(), w.end());
    return z;
  };
  ExpectVecNear(gemm_on(cl), gemm_on(defaultDevice), 1e-4, "gemm");
  // convolution / pooling geometries
  struct Geo { int C, K, R, stride, pad, groups, H; };
  for (Geo ge : {Geo{3, 8, 3, 1, 1, 1, 9}, Geo{4, 6, 5, 2, 2, 2, 11},
                 Geo{8, 8, 1, 1, 0, 1, 7}, Geo{6, 4, 3, 2, 0, 1, 10}}) {
    ConvParams cp;
    cp.num_filters = ge.K; cp.kernel_h = cp.kernel_w = ge.R;
    cp.stride_h = cp.stride_w = ge.stride; cp.pad_h = cp.pad_w = ge.pad;
    cp.group = ge.groups;
    const Shape xs{2, size_t(ge.C), size_t(ge.H), size_t(ge.H)};
    const Shape ws{size_t(ge.K), size_t(ge.C / ge.groups), size_t(ge.R),
                   size_t(ge.R)};
    auto xv = Rand(Product(xs), 21), wv = Rand(Product(ws), 22),
         bv = Rand(ge.K, 23);
    std::vector<std::vector<float>> got[2];
    int which = 0;
    for (auto d : {cl, defaultDevice}) {
      Tensor x = FromVec(xv, xs, d), w = FromVec(wv, ws, d);
      Tensor b = FromVec(bv, {size_t(ge.K)}, d);
      Tensor y = ConvForward(x, w, &b, cp);
      Tensor dy = FromVec(Rand(y.Size(), 24), y.shape(), d);
      got[which].push_back(ToVec(y));
      got[which].push_back(ToVec(ConvBackwardData(dy, w, xs, cp)));
      got[which].push_back(ToVec(ConvBackwardWeight(dy, x, ws, cp)));
      got[which].push_back(ToVec(ConvBackwardBias(dy)));
      for (bool is_max : {true, false}) {
        PoolParams pp;
        pp.kernel_h = pp.kernel_w = 3; pp.stride_h = pp.stride_w = 2;
        pp.pad_h = pp.pad_w = 1; pp.is_max = is_max;
        Tensor py = PoolForward(x, pp);
        Tensor pdy = FromVec(Rand(py.Size(), 25), py.shape(), d);
        got[which].push_back(ToVec(py));
        got[which].push_back(ToVec(PoolBackward(pdy, x, py, pp)));
      }
      ++which;
    }
    for (size_t i = 0; i < got[0].size(); ++i)
      ExpectVecNear(got[0][i], got[1][i], 1e-4, "conv / pool pass");
  }
  // int tensors and casts
  Tensor I = FromInts({3, -4, 5, 6}, {4}, cl);
  Tensor F = I.Clone();
  F.ToType(kFloat32);
  EXPECT_NEAR(ToVec(F)[1], -4.f, 0);
  // the generator: reseeding repeats the stream
  Tensor r1(Shape{512}, cl), r2(Shape{512}, cl);
  cl->SetRandSeed(11);
  Gaussian(0.f, 1.f, &r1);
  cl->SetRandSeed(11);
  Gaussian(0.f, 1.f, &r2);
  ExpectVecNear(ToVec(r1), ToVec(r2), 0, "reseeded stream");
  Uniform(0.f, 1.f, &r2);
  float differs = 0;
  auto a1 = ToVec(r1), a2 = ToVec(r2);
  for (size_t i = 0; i < a1.size(); ++i) differs += a1[i] != a2[i];
  EXPECT(differs > 500);
}

struct LayerResult {
  std::vector<float> y, dx;
  std::vector<std::vector<float>> dp;
};

static LayerResult RunLayer(const std::string &type, const LayerConf &base,
                            const Shape &sample, size_t batch,
                            std::shared_ptr<Device> dev, unsigned seed) {
  LayerConf conf = base;
  conf.type = type;
  conf.name = "l";
  auto layer = CreateLayer(type);
  layer->Setup(sample, conf);
  layer->ToDevice(dev);
  unsigned s = seed;
  for (auto p : layer->param_values()) {
    // BatchNorm's running variance must stay positive
    bool positive = layer->layer_type() == "BatchNorm";
    auto v = Rand(p.Size(), ++s, 0.3f);
    if (positive) for (auto &x : v) x = std::fabs(x) + 0.5f;
    p.CopyDataFromHostPtr(v.data(), v.size());
  }
  Shape xs{batch};
  for (size_t d : sample) xs.push_back(d);
  Tensor x = FromVec(Rand(Product(xs), seed + 100), xs, dev);
  Tensor y = layer->Forward(kTrain, x);
  Tensor dy = FromVec(Rand(y.Size(), seed + 200), y.shape(), dev);
  auto ret = layer->Backward(kTrain, dy);
  LayerResult r;
  r.y = ToVec(y);
  r.dx = ToVec(ret.first);
  for (auto &g : ret.second)
    r.dp.push_back(g.empty() ? std::vector<float>() : ToVec(g));
  return r;
}

// finite-difference check of dx on the host for a scalar objective sum(y*dy)
static void GradCheck(const std::string &type, const LayerConf &base,
                      const Shape &sample, size_t batch) {
  LayerConf conf = base;
  conf.type = type;
  conf.name = "g";
  auto layer = CreateLayer(type);
  layer->Setup(sample, conf);
  unsigned s = 7;
  for (auto p : layer->param_values()) {
    auto v = Rand(p.Size(), ++s, 0.3f);
    if (layer->layer_type() == "BatchNorm") for (auto &x : v) x = std::fabs(x) + 0.5f;
    p.CopyDataFromHostPtr(v.data(), v.size());
  }
  Shape xs{batch};
  for (size_t d : sample) xs.push_back(d);
  auto xv = Rand(Product(xs), 11);
  Tensor x = FromVec(xv, xs, defaultDevice);
  Tensor y = layer->Forward(kTrain, x);
  auto dyv = Rand(y.Size(), 12);
  Tensor dy = FromVec(dyv, y.shape(), defaultDevice);
  auto dx = ToVec(layer->Backward(kTrain, dy).first);
  auto objective = [&](const std::vector<float> &in) {
    Tensor xi = FromVec(in, xs, defaultDevice);
    auto yy = ToVec(layer->Forward(kEval | kTrain, xi));
    layer->Backward(kTrain, dy);  // pop the cache
    double acc = 0;
    for (size_t i = 0; i < yy.size(); ++i) acc += double(yy[i]) * dyv[i];
    return acc;
  };
  const float eps = 1e-2f;
  for (size_t i = 0; i < xv.size(); i += std::max<size_t>(1, xv.size() / 7)) {
    auto xp = xv, xm = xv;
    xp[i] += eps;
    xm[i] -= eps;
    double num = (objective(xp) - objective(xm)) / (2 * eps);
    ++g_checks;
    if (std::fabs(num - dx[i]) > 2e-2 * (1 + std::fabs(num))) {
      ++g_failed;
      std::printf("  FAILED gradcheck %s at %zu: numeric %g vs analytic %g\n",
                  type.c_str(), i, num, dx[i]);
    }
  }
}

struct LayerCase {
  std::string type;
  LayerConf conf;
  Shape sample;
  size_t batch;
};

static std::vector<LayerCase> LayerCases() {
  std::vector<LayerCase> cases;
  LayerConf c;
  c.dense_conf.num_output = 5;
  cases.push_back({"singacpp_dense", c, {7}, 4});
  c = LayerConf();
  c.dense_conf.num_output = 3;
  c.dense_conf.transpose = true;
  cases.push_back({"singacpp_dense", c, {6}, 2});
  c = LayerConf();
  cases.push_back({"singacpp_relu", c, {3, 4, 4}, 2});
  cases.push_back({"singacpp_sigmoid", c, {10}, 3});
  cases.push_back({"singacpp_tanh", c, {10}, 3});
  c.relu_conf.negative_slope = 0.1f;
  cases.push_back({"singacpp_relu", c, {10}, 3});
  c = LayerConf();
  c.convolution_conf.num_output = 4;
  c.convolution_conf.set_kernel(3);
  c.convolution_conf.set_pad(1);
  cases.push_back({"singacpp_convolution", c, {3, 8, 8}, 2});
  c.convolution_conf.set_stride(2);
  c.convolution_conf.bias_term = false;
  cases.push_back({"singacpp_convolution", c, {4, 9, 9}, 2});
  c = LayerConf();
  c.convolution_conf.num_output = 6;
  c.convolution_conf.set_kernel(1);
  c.convolution_conf.group = 2;
  cases.push_back({"singacpp_convolution", c, {4, 5, 5}, 2});
  // enough channels and positions for the HIP device's Winograd route
  c = LayerConf();
  c.convolution_conf.num_output = 32;
  c.convolution_conf.set_kernel(3);
  c.convolution_conf.set_pad(1);
  cases.push_back({"singacpp_convolution", c, {32, 12, 11}, 4});
  c = LayerConf();
  c.pooling_conf.set_kernel(2);
  c.pooling_conf.set_stride(2);
  cases.push_back({"singacpp_pooling", c, {3, 8, 8}, 2});
  c.pooling_conf.pool = PoolingConf::AVE;
  c.pooling_conf.set_kernel(3);
  c.pooling_conf.set_pad(1);
  cases.push_back({"singacpp_pooling", c, {2, 7, 7}, 2});
  c = LayerConf();
  cases.push_back({"singacpp_batchnorm", c, {3, 4, 4}, 4});
  cases.push_back({"singacpp_lrn", c, {8, 3, 3}, 2});
  cases.push_back({"singacpp_softmax", c, {6}, 3});
  cases.push_back({"singacpp_flatten", c, {2, 3, 4}, 2});
  cases.push_back({"singacpp_prelu", c, {3, 2, 2}, 2});
  return cases;
}

static void TestLayersGradient() {
  for (auto &lc : LayerCases()) {
    if (lc.type == "singacpp_flatten") continue;
    GradCheck(lc.type, lc.conf, lc.sample, lc.batch);
  }
}

static void TestLayersGpuVsCpu() {
  unsigned seed = 40;
  for (auto &lc : LayerCases()) {
    ++seed;
    LayerResult c = RunLayer(lc.type, lc.conf, lc.sample, lc.batch,
                             defaultDevice, seed);
    LayerResult g = RunLayer(lc.type, lc.conf, lc.sample, lc.batch, g_dev, seed);
    ExpectVecNear(g.y, c.y, 2e-4, (lc.type + " forward").c_str());
    ExpectVecNear(g.dx, c.dx, 2e-4, (lc.type + " dx").c_str());
    EXPECT(g.dp.size() == c.dp.size());
    for (size_t i = 0; i < std::min(g.dp.size(), c.dp.size()); ++i)
      ExpectVecNear(g.dp[i], c.dp[i], 5e-4, (lc.type + " dparam").c_str());
  }
}

static void TestDropoutAndMulti(std::shared_ptr<Device> dev) {
  LayerConf c;
  c.type = "singacpp_dropout";
  c.dropout_conf.dropout_ratio = 0.4f;
  auto drop = CreateLayer(c.type);
  drop->Setup(Shape{1000}, c);
  Tensor x(Shape{8, 1000}, dev);
  x.SetValue(1.f);
  Tensor y = drop->Forward(kTrain, x);
  float kept = Sum<float>(y) / 8000;  // E[y] = 1
  EXPECT(kept > 0.93 && kept < 1.07);
  EXPECT_NEAR(Sum<float>(drop->Forward(kEval, x)), 8000, 1e-6);
  Tensor dx = drop->Backward(kTrain, x).first;
  EXPECT_NEAR(Sum<float>(dx), Sum<float>(y), 1e-5);

  LayerConf cc;
  cc.type = "singacpp_concat";
  cc.concat_conf.axis = 1;
  auto concat = CreateLayer(cc.type);
  concat->Setup(std::vector<Shape>{{3}, {2}}, cc);
  Tensor a = FromVec(Rand(6, 1), {2, 3}, dev), b = FromVec(Rand(4, 2), {2, 2}, dev);
  auto out = concat->Forward(kTrain, std::vector<Tensor>{a, b});
  EXPECT(out[0].shape(1) == 5);
  auto back = concat->Backward(kTrain, out).first;
  ExpectVecNear(ToVec(back[1]), ToVec(b), 1e-6, "concat backward");

  LayerConf sc;
  sc.type = "singacpp_slice";
  sc.slice_conf.axis = 1;
  sc.slice_conf.slice_point = {2};
  auto slice = CreateLayer(sc.type);
  slice->Setup(Shape{5}, sc);
  auto parts = slice->Forward(kTrain, std::vector<Tensor>{out[0]});
  EXPECT(parts.size() == 2 && parts[1].shape(1) == 3);
  auto joined = slice->Backward(kTrain, parts).first;
  ExpectVecNear(ToVec(joined[0]), ToVec(out[0]), 1e-6, "slice backward");

  LayerConf pc;
  pc.type = "singacpp_split";
  pc.split_conf.output_size = 3;
  auto split = CreateLayer(pc.type);
  split->Setup(Shape{3}, pc);
  auto reps = split->Forward(kTrain, std::vector<Tensor>{a});
  EXPECT(reps.size() == 3);
  auto g = split->Backward(kTrain, reps).first;
  EXPECT_NEAR(ToVec(g[0])[0], 3 * ToVec(a)[0], 1e-5);

  LayerConf mc;
  mc.type = "singacpp_merge";
  auto merge = CreateLayer(mc.type);
  merge->Setup(Shape{3}, mc);
  auto m = merge->Forward(kTrain, std::vector<Tensor>{a, a});
  EXPECT_NEAR(ToVec(m[0])[2], 2 * ToVec(a)[2], 1e-5);
  EXPECT(merge->Backward(kTrain, m).first.size() == 2);
}

struct RnnResult {
  double objective;
  std::vector<std::vector<float>> dx, dp;
};

// objective = sum_t <y_t, w_t> + <hy, w_T> (+ <cy, w_T+1>)
static RnnResult RunRnn(const std::string &mode, std::shared_ptr<Device> dev,
                        const std::vector<std::vector<float>> &xs,
                        const std::vector<std::vector<float>> &params,
                        bool backward) {
  const size_t T = xs.size(), B = 3, F = 4, H = 5;
  LayerConf conf;
  conf.name = "rnn";
  conf.type = "singacpp_rnn";
  conf.rnn_conf.rnn_mode = mode;
  conf.rnn_conf.hidden_size = H;
  auto layer = CreateLayer(conf.type);
  layer->Setup(Shape{F}, conf);
  layer->ToDevice(dev);
  auto pv = layer->param_values();
  for (size_t i = 0; i < pv.size(); ++i)
    pv[i].CopyDataFromHostPtr(params[i].data(), params[i].size());
  bool cell = mode == "lstm";
  std::vector<Tensor> in;
  for (size_t t = 0; t < T; ++t) in.push_back(FromVec(xs[t], {B, F}, dev));
  in.push_back(FromVec(Rand(B * H, 71, 0.5f), {B, H}, dev));
  if (cell) in.push_back(FromVec(Rand(B * H, 72, 0.5f), {B, H}, dev));
  auto out = layer->Forward(kTrain, in);
  RnnResult r;
  r.objective = 0;
  std::vector<Tensor> grads;
  for (size_t k = 0; k < out.size(); ++k) {
    auto w = Rand(B * H, 80 + k);
    auto y = ToVec(out[k]);
    for (size_t i = 0; i < y.size(); ++i) r.objective += double(y[i]) * w[i];
    grads.push_back(FromVec(w, {B, H}, dev));
  }
  if (backward) {
    auto ret = layer->Backward(kTrain, grads);
    for (auto &g : ret.first) r.dx.push_back(ToVec(g));
    for (auto &g : ret.second)
    r.dp.push_back(g.empty() ? std::vector<float>() : ToVec(g));
  } else {
    layer->Backward(kTrain, grads);
  }
  return r;
}

static void TestRnn() {
  const size_t T = 3, B = 3, F = 4, H = 5;
  for (std::string mode : {"tanh", "relu", "lstm", "gru"}) {
    size_t G = mode == "lstm" ? 4 : (mode == "gru" ? 3 : 1);
    std::vector<std::vector<float>> xs, params;
    for (size_t t = 0; t < T; ++t) xs.push_back(Rand(B * F, 60 + t));
    params.push_back(Rand(F * G * H, 90, 0.4f));
    params.push_back(Rand(H * G * H, 91, 0.4f));
    params.push_back(Rand(G * H, 92, 0.2f));
    if (mode == "gru") params.push_back(Rand(G * H, 93, 0.2f));
    RnnResult ref = RunRnn(mode, defaultDevice, xs, params, true);
    EXPECT(ref.dx.size() == T + (mode == "lstm" ? 2 : 1));
    const float eps = 1e-2f;
    // inputs
    for (size_t t = 0; t < T; ++t)
      for (size_t i : {size_t(0), size_t(5), size_t(11)}) {
        auto p = xs, m = xs;
        p[t][i] += eps;
        m[t][i] -= eps;
        double num = (RunRnn(mode, defaultDevice, p, params, false).objective -
                      RunRnn(mode, defaultDevice, m, params, false).objective) /
                     (2 * eps);
        EXPECT_NEAR(ref.dx[t][i], num, 3e-2);
      }
    // parameters
    for (size_t k = 0; k < params.size(); ++k)
      for (size_t i : {size_t(0), params[k].size() / 2, params[k].size() - 1}) {
        auto p = params, m = params;
        p[k][i] += eps;
        m[k][i] -= eps;
        double num = (RunRnn(mode, defaultDevice, xs, p, false).objective -
                      RunRnn(mode, defaultDevice, xs, m, false).objective) /
                     (2 * eps);
        EXPECT_NEAR(ref.dp[k][i], num, 3e-2);
      }
    if (g_gpu) {
      RnnResult gpu = RunRnn(mode, g_dev, xs, params, true);
      EXPECT_NEAR(gpu.objective, ref.objective, 1e-4);
      for (size_t k = 0; k < ref.dx.size(); ++k)
        ExpectVecNear(gpu.dx[k], ref.dx[k], 2e-4, (mode + " rnn dx").c_str());
      for (size_t k = 0; k < ref.dp.size(); ++k)
        ExpectVecNear(gpu.dp[k], ref.dp[k], 5e-4, (mode + " rnn dparam").c_str());
    }
  }
}

static void TestOptimizers(std::shared_ptr<Device> dev) {
  OptimizerConf conf;
  conf.momentum = 0.9f;
  SGD sgd;
  sgd.Setup(conf);
  Tensor w = FromVec({0.1f, 0.2f, 0.3f, 0.4f}, {4}, dev);
  Tensor g = FromVec({0.1f, 0.1f, 0.1f, 0.1f}, {4}, dev);
  sgd.Apply(0, 0.1f, "w", g, w);
  EXPECT_NEAR(ToVec(w)[0], 0.1f - 0.01f, 1e-6);
  sgd.Apply(1, 0.1f, "w", g, w);  // v = 0.9*0.01 + 0.01
  EXPECT_NEAR(ToVec(w)[0], 0.09f - 0.019f, 1e-6);

  Nesterov nes;
  nes.Setup(conf);
  Tensor w2 = FromVec({0.1f}, {1}, dev), g2 = FromVec({0.1f}, {1}, dev);
  nes.Apply(0, 0.1f, "w", g2, w2);  // v'=0.01, w -= 1.9*0.01
  EXPECT_NEAR(ToVec(w2)[0], 0.1f - 0.019f, 1e-6);

  AdaGrad ada;
  OptimizerConf ac;
  ac.delta = 1e-8f;
  ada.Setup(ac);
  Tensor w3 = FromVec({0.1f}, {1}, dev), g3 = FromVec({0.01f}, {1}, dev);
  ada.Apply(0, 0.1f, "w", g3, w3);  // g/sqrt(g^2) = 1
  EXPECT_NEAR(ToVec(w3)[0], 0.0f, 1e-4);

  RMSProp rms;
  OptimizerConf rc;
  rc.rho = 0.9f;
  rc.delta = 1e-8f;
  rms.Setup(rc);
  Tensor w4 = FromVec({0.1f}, {1}, dev), g4 = FromVec({0.01f}, {1}, dev);
  rms.Apply(0, 0.1f, "w", g4, w4);  // h = 0.1*1e-4, upd = 0.01/sqrt(1e-5)
  EXPECT_NEAR(ToVec(w4)[0], 0.1f - 0.1f * 0.01f / std::sqrt(1e-5f + 1e-8f), 1e-4);

  // regularizer + constraint + lr multiplier through the name-based Apply
  OptimizerConf oc;
  oc.has_regularizer = true;
  oc.regularizer.coefficient = 0.5f;
  SGD reg;
  reg.Setup(oc);
  reg.SetLearningRateGenerator([](int) { return 0.1f; });
  ParamSpec spec;
  spec.name = "w";
  spec.lr_mult = 2.f;
  reg.Register("w", spec);
  Tensor w5 = FromVec({1.f}, {1}, dev), g5 = FromVec({0.f}, {1}, dev);
  reg.Apply(0, "w", g5, w5);  // grad = 0.5*w, lr = 0.2
  EXPECT_NEAR(ToVec(w5)[0], 1.f - 0.2f * 0.5f, 1e-6);
  Constraint clip("l2", 1.f);
  Tensor big = FromVec({3.f, 4.f}, {2}, dev), dummy = big.Clone();
  clip.Apply(0, big, dummy);
  EXPECT_NEAR(ToVec(big)[1], 0.8f, 1e-5);
}

static void TestLossMetric(std::shared_ptr<Device> dev) {
  Tensor p = FromVec({1, 2, 3, 1, 0, -1}, {2, 3}, dev);
  Tensor t = FromVec({1, 1, 1, 0, 0, 0}, {2, 3}, dev);
  MSE mse;
  Tensor l = mse.Forward(kTrain, p, t);
  EXPECT_NEAR(ToVec(l)[0], 0.5f * (0 + 1 + 4), 1e-6);
  Tensor g = mse.Backward();
  EXPECT_NEAR(ToVec(g)[2], (3 - 1) / 2.f, 1e-6);
  SoftmaxCrossEntropy ce;
  Tensor lab = FromInts({2, 0}, {2}, dev);
  float loss = ce.Evaluate(kTrain, p, lab);
  double z0 = std::exp(1.) + std::exp(2.) + std::exp(3.);
  double z1 = std::exp(1.) + std::exp(0.) + std::exp(-1.);
  EXPECT_NEAR(loss, 0.5 * (-std::log(std::exp(3.) / z0) - std::log(std::exp(1.) / z1)), 1e-5);
  Tensor cg = ce.Backward();
  EXPECT_NEAR(ToVec(cg)[2], std::exp(3.) / z0 - 1, 1e-5);
  Accuracy acc;
  EXPECT_NEAR(acc.Evaluate(p, lab), 1.0, 1e-6);
  Tensor lab2 = FromInts({0, 1}, {2}, dev);
  EXPECT_NEAR(acc.Evaluate(p, lab2), 0.0, 1e-6);
  Accuracy top2(2);
  EXPECT_NEAR(top2.Evaluate(p, FromInts({1, 1}, {2}, dev)), 1.0, 1e-6);
}

static FeedForwardNet MakeMlp() {
  FeedForwardNet net;
  Shape in{2};
  LayerConf d1;
  d1.name = "fc1";
  d1.type = "singacpp_dense";
  d1.dense_conf.num_output = 16;
  ParamSpec w;
  w.name = "fc1_w";
  w.filler.type = "gaussian";
  w.filler.std = 0.5f;
  ParamSpec b;
  b.name = "fc1_b";
  d1.param = {w, b};
  net.Add(d1, &in);
  LayerConf r;
  r.name = "relu1";
  r.type = "singacpp_relu";
  net.Add(r);
  LayerConf d2;
  d2.name = "fc2";
  d2.type = "singacpp_dense";
  d2.dense_conf.num_output = 2;
  w.name = "fc2_w";
  b.name = "fc2_b";
  d2.param = {w, b};
  net.Add(d2);
  return net;
}

static void MakeToy(std::vector<float> *x, std::vector<int> *y, size_t n) {
  std::mt19937 g(3);
  std::uniform_real_distribution<float> u(-1, 1);
  for (size_t i = 0; i < n; ++i) {
    float a = u(g), b = u(g);
    x->push_back(a);
    x->push_back(b);
    y->push_back(a * a + b * b < 0.5f ? 1 : 0);
  }
}

static void TestFeedForwardNet(std::shared_ptr<Device> dev, bool graph) {
  defaultDevice->SetRandSeed(1);
  dev->SetRandSeed(1);
  FeedForwardNet net = MakeMlp();
  net.set_verbose(false);
  SGD sgd;
  OptimizerConf oc;
  oc.momentum = 0.9f;
  sgd.Setup(oc);
  sgd.SetLearningRateGenerator([](int) { return 0.1f; });
  SoftmaxCrossEntropy loss;
  Accuracy acc;
  net.Compile(true, &sgd, &loss, &acc);
  net.ToDevice(dev);
  std::vector<float> xv;
  std::vector<int> yv;
  MakeToy(&xv, &yv, 512);
  Tensor x = FromVec(xv, {512, 2}, dev), y = FromInts(yv, {512}, dev);
  float first = net.TrainOnBatch(0, x, y).first;
  if (graph) {
    // record one training step, then replay it
    Tensor bx = CopyRows(x, 0, 128), by = CopyRows(y, 0, 128);
    dev->Sync();
    dev->EnableGraph(true);
    Tensor fea = net.Forward(kTrain, bx);
    Tensor l = loss.Forward(kTrain, fea, by);
    auto grads = net.Backward(kTrain, loss.Backward() * (1.f / 128));
    auto values = net.GetParamValues();
    auto names = net.GetParamNames();
    for (size_t k = 0; k < grads.size(); ++k)
      sgd.Apply(0, 0.1f, names[k], grads[k], values[k]);
    dev->EnableGraph(false);
    EXPECT(dev->graph()->nodes().size() > 10);
    for (int it = 0; it < 150; ++it) dev->RunGraph(it % 2 == 0);
    dev->Sync();
    dev->ResetGraph();
  } else {
    net.Train(64, 25, x, y);
  }
  auto ev = net.Evaluate(x, y, 128);
  float final_loss = Sum<float>(ev.first) / ev.first.Size();
  float final_acc = Sum<float>(ev.second) / ev.second.Size();
  EXPECT(final_loss < first);
  EXPECT(final_acc > (graph ? 0.7f : 0.9f));
  Tensor pred = net.Predict(x, 100);
  EXPECT(pred.shape(0) == 512 && pred.shape(1) == 2);
  // clone shares values, not storage
  FeedForwardNet twin = net.Clone(defaultDevice);
  auto a = ToVec(net.GetParamValues()[0]), b = ToVec(twin.GetParamValues()[0]);
  ExpectVecNear(a, b, 0, "clone copies parameters");
}

static void TestLocalUpdater() {
  // two replicas on the host, each sees half of every batch: the averaged
  // update must equal the single-model update on the whole batch
  SGD opt;
  opt.SetLearningRateGenerator([](int) { return 0.5f; });
  auto up = std::make_shared<LocalUpdater>(2, &opt);
  ParamSpec spec;
  spec.name = "w";
  up->Register("w", spec);
  Tensor w0 = FromVec({1.f, 2.f}, {2}, defaultDevice), w1 = w0.Clone();
  Tensor g0 = FromVec({0.2f, 0.4f}, {2}, defaultDevice);
  Tensor g1 = FromVec({0.6f, 0.0f}, {2}, defaultDevice);
  std::thread t0([&] { for (int s = 0; s < 3; ++s) up->Apply(s, "w", g0, w0); });
  std::thread t1([&] { for (int s = 0; s < 3; ++s) up->Apply(s, "w", g1, w1); });
  t0.join();
  t1.join();
  EXPECT_NEAR(ToVec(w0)[0], 1.f - 3 * 0.5f * 0.4f, 1e-5);
  EXPECT_NEAR(ToVec(w1)[1], 2.f - 3 * 0.5f * 0.2f, 1e-5);
  ExpectVecNear(ToVec(w0), ToVec(w1), 0, "replicas agree");
}

static void TestMemPoolAndDevice() {
  HostMemPool pool;
  void *a = nullptr, *b = nullptr;
  pool.Malloc(&a, 1000);
  pool.Free(a);
  pool.Malloc(&b, 900);  // same 1 KiB class: the chunk is reused
  EXPECT(a == b);
  pool.Free(b);
  EXPECT(HipMemPool::RoundUp(1) == 512);
  EXPECT(HipMemPool::RoundUp(513) == 1024);
  EXPECT(HipMemPool::RoundUp((1 << 20) + 1) == (2u << 20));
  EXPECT(HipMemPool::RoundUp((2 << 20) + 1) == (4u << 20));
  // lazy allocation
  Tensor t(Shape{100});
  EXPECT(!t.initialized());
  t.SetValue(1.f);
  EXPECT(t.initialized());
  if (g_gpu) {
    auto *hp = new HipMemPool();
    int before = hp->driver_calls();
    void *p = nullptr, *q = nullptr;
    hp->Malloc(&p, 4096);
    hp->Free(p);
    hp->Malloc(&q, 4000);
    EXPECT(p == q);
    EXPECT(hp->driver_calls() == size_t(before + 1));
    hp->Free(q);
    auto usage = hp->GetMemUsage();
    EXPECT(usage.second > (size_t(100) << 30));  // 288 GB part
    delete hp;
    EXPECT(Platform::GetNumGPUs() >= 1);
    std::string q0 = Platform::DeviceQuery(0);
    EXPECT(q0.find("gfx950") != std::string::npos);
  }
}

static void TestIo() {
  // snapshot round trip
  std::string prefix = "/tmp/singa_cpp_test_snapshot";
  {
    Snapshot w(prefix, Snapshot::kWrite);
    w.Write("a", FromVec({1, 2, 3, 4, 5, 6}, {2, 3}, defaultDevice));
    w.Write("b", FromInts({7, 8}, {2}, defaultDevice));
  }
  Snapshot r(prefix, Snapshot::kRead);
  auto all = r.Read();
  EXPECT(all.size() == 2);
  EXPECT(r.ReadShape("a") == (Shape{2, 3}));
  EXPECT_NEAR(ToVec(r.Read("a"))[4], 5, 0);
  EXPECT(r.Read("b").ToVector<int>()[1] == 8);
  // record files through the Reader / Writer interfaces
  {
    const std::string bin = "/tmp/singa_cpp_test_records.bin";
    io::BinFileWriter bw;
    io::Writer *w = &bw;
    EXPECT(w->Open(bin, io::kCreate));
    EXPECT(w->Write("k1", std::string("\0\1\2", 3)));
    EXPECT(w->Write("", "no key"));
    EXPECT(!w->Write("k3", ""));  // empty values are rejected
    w->Close();
    EXPECT(w->Open(bin, io::kAppend));
    EXPECT(w->Write("k4", std::string(70000, 'x')));
    w->Close();
    io::BinFileReader br;
    io::Reader *rd = &br;
    EXPECT(rd->Open(bin));
    EXPECT(rd->Count() == 3);
    std::string k, v;
    EXPECT(rd->Read(&k, &v) && k == "k1" && v == std::string("\0\1\2", 3));
    EXPECT(rd->Read(&k, &v) && k.empty() && v == "no key");
    EXPECT(rd->Read(&k, &v) && k == "k4" && v.size() == 70000);
    EXPECT(!rd->Read(&k, &v));
    rd->SeekToFirst();
    EXPECT(rd->Read(&k, &v) && k == "k1");
    rd->Close();
    const std::string txt = "/tmp/singa_cpp_test_records.txt";
    io::TextFileWriter tw;
    EXPECT(tw.Open(txt, io::kCreate));
    EXPECT(tw.Write("", "first") && tw.Write("", "second"));
    tw.Close();
    io::TextFileReader tr;
    EXPECT(tr.Open(txt));
    EXPECT(tr.Count() == 2);
    EXPECT(tr.Read(&k, &v) && k == "0" && v == "first");
    EXPECT(tr.Read(&k, &v) && k == "1" && v == "second");
    EXPECT(!tr.Read(&k, &v));
    tr.Close();
  }
  // LMDB environments: small values in the leaves, big values in overflow
  // runs, several tree levels, append mode, key order
  {
    const std::string dir = "/tmp/singa_cpp_test_lmdb";
    std::remove((dir + "/data.mdb").c_str());
    rmdir(dir.c_str());
    io::LMDBWriter lw;
    io::Writer *w = &lw;
    EXPECT(w->Open(dir, io::kCreate));
    char name[16];
    for (int i = 2999; i >= 0; --i) {  // written in reverse key order
      snprintf(name, sizeof(name), "%08d", i);
      EXPECT(w->Write(name, std::string(size_t(i % 97 + 1), char('a' + i % 26))));
    }
    EXPECT(w->Write("big", std::string(100000, 'B')));
    EXPECT(w->Write("00000007", "overwritten"));
    w->Close();
    io::LMDBReader lr;
    io::Reader *rd = &lr;
    EXPECT(rd->Open(dir));
    EXPECT(rd->Count() == 3001);
    std::string k, v, prev;
    int n = 0;
    bool ordered = true, values_ok = true;
    while (rd->Read(&k, &v)) {
      if (n > 0 && !(prev < k)) ordered = false;
      prev = k;
      if (k == "big") {
        values_ok = values_ok && v == std::string(100000, 'B');
      } else if (k == "00000007") {
        values_ok = values_ok && v == "overwritten";
      } else {
        int i = atoi(k.c_str());
        values_ok = values_ok &&
                    v == std::string(size_t(i % 97 + 1), char('a' + i % 26));
      }
      ++n;
    }
    EXPECT(n == 3001 && ordered && values_ok && k == "big");
    EXPECT(!lr.valid());
    rd->SeekToFirst();
    EXPECT(rd->Read(&k, &v) && k == "00000000" && v == "a");
    rd->Close();
    EXPECT(w->Open(dir, io::kAppend));
    EXPECT(w->Write("zzz", "last"));
    w->Close();
    EXPECT(rd->Open(dir));
    EXPECT(rd->Count() == 3002);
    rd->Close();
    io::LMDBReader missing;
    EXPECT(!missing.Open("/tmp/singa_cpp_test_lmdb_missing"));
  }
  // channel to a file, stopwatch
  {
    SetChannelDirectory("/tmp");
    Channel *ch = GetChannel("singa_cpp_test_channel");
    ch->SetDestFilePath("/tmp/singa_cpp_test_channel.log");
    std::remove("/tmp/singa_cpp_test_channel.log");
    ch->Send("loss = 1.5");
    ch->Send("loss = 1.25");
    io::TextFileReader tr;
    EXPECT(tr.Open("/tmp/singa_cpp_test_channel.log"));
    EXPECT(tr.Count() == 2);
    EXPECT(GetChannel("singa_cpp_test_channel") == ch);
    Timer t;
    EXPECT(t.Elapsed() >= 0.0 && t.Elapsed() < 1000.0);
  }
  // csv
  CSVEncoder enc;
  CSVDecoder dec;
  std::vector<Tensor> rec{FromVec({1.5f, -2.f, 3.25f}, {3}, defaultDevice),
                          FromInts({4}, {1}, defaultDevice)};
  std::string line = enc.Encode(rec);
  EXPECT(line == "4,1.5,-2,3.25");
  auto back = dec.Decode(line);
  EXPECT(back.size() == 2 && back[1].ToVector<int>()[0] == 4);
  EXPECT_NEAR(ToVec(back[0])[2], 3.25, 0);
  // image codec, CHW <-> PPM
  std::vector<float> px(3 * 4 * 5);
  for (size_t i = 0; i < px.size(); ++i) px[i] = float((i * 7) % 256);
  ImageEncoder ienc;
  EncoderConf ec;
  ec.image_dim_order = "CHW";
  ienc.Setup(ec);
  std::vector<Tensor> irec{FromVec(px, {3, 4, 5}, defaultDevice),
                           FromInts({9}, {1}, defaultDevice)};
  std::string ppm = ienc.Encode(irec);
  EXPECT(ppm.substr(0, 2) == "P6");
  ImageDecoder idec;
  auto img = idec.Decode(ppm);
  EXPECT(img.size() == 2 && img[1].ToVector<int>()[0] == 9);
  ExpectVecNear(ToVec(img[0]), px, 0, "image codec round trip");
  // transformer
  ImageTransformer tr;
  TransformerConf tc;
  tc.resize_height = 8;
  tc.resize_width = 10;
  tc.crop_height = 6;
  tc.crop_width = 6;
  tc.horizontal_mirror = true;
  tr.Setup(tc);
  Tensor out = tr.Apply(kTrain, img[0]);
  EXPECT(out.shape() == (Shape{3, 6, 6}));
  Tensor ev = tr.Apply(kEval, img[0]);
  Tensor manual = resize(img[0], 8, 10, "CHW");
  manual = crop(manual, 6, 6, 1, 2, "CHW");
  ExpectVecNear(ToVec(ev), ToVec(manual), 1e-6, "eval = resize + centre crop");
  Tensor flipped = mirror(img[0], true, false, "CHW");
  EXPECT_NEAR(ToVec(flipped)[0], px[4], 0);
  Tensor same = resize(img[0], 4, 5, "CHW");
  ExpectVecNear(ToVec(same), px, 1e-5, "identity resize");
  // varint
  char buf[16];
  for (uint64_t v : {0ull, 127ull, 128ull, 16383ull, 16384ull, 2097151ull,
                     2097152ull, 4294967295ull, 4294967296ull}) {
    uint64_t got = 0;
    size_t n = appendInteger(buf, v);
    EXPECT(readInteger(buf, got) == n && got == v);
  }
  Tokenizer tok("a b,c", " ,");
  std::string s1, s2, s3;
  tok >> s1 >> s2 >> s3;
  EXPECT(s1 == "a" && s2 == "b" && s3 == "c" && !tok.Valid());
}

static void TestEndPoint() {
  NetworkThread a(0), b(0);
  EndPoint *to_b = a.epf_->getEp("127.0.0.1", b.port());
  std::string big(300000, 'x');
  for (int i = 0; i < 5; ++i) {
    Message m;
    std::string meta = "m" + std::to_string(i);
    m.setMetadata(meta.data(), meta.size());
    m.setPayload(big.data(), big.size());
    EXPECT(to_b->send(&m) > 0);
  }
  EndPoint *from_a = b.epf_->waitAny(EP_TIMEOUT);
  EXPECT(from_a != nullptr);
  if (!from_a) return;
  EXPECT(from_a->peer() == "127.0.0.1:" + std::to_string(a.port()));
  for (int i = 0; i < 5; ++i) {
    Message *m = from_a->recv(EP_TIMEOUT);
    EXPECT(m != nullptr);
    if (!m) return;
    void *meta, *payload;
    size_t ms = m->getMetadata(&meta), ps = m->getPayload(&payload);
    EXPECT(std::string(static_cast<char *>(meta), ms) == "m" + std::to_string(i));
    EXPECT(ps == big.size());
    delete m;
  }
  EXPECT(to_b->drain());
  EXPECT(to_b->pending_acks() == 0);
  // reply over the same connection
  Message reply;
  reply.setPayload("pong", 4);
  EXPECT(from_a->send(&reply) > 0);
  Message *r = to_b->recv(EP_TIMEOUT);
  EXPECT(r != nullptr);
  if (r) {
    void *p;
    EXPECT(r->getPayload(&p) == 4);
    delete r;
  }
  // an unreachable peer fails after the retries instead of hanging
  EndPoint *nowhere = a.epf_->getEp("127.0.0.1", 1);
  Message lost;
  lost.setPayload("x", 1);
  EXPECT(nowhere->send(&lost) == -1);
}

static void Run(const char *name, const std::function<void()> &fn) {
  int before = g_failed;
  std::printf("[ RUN  ] %s\n", name);
  try {
    fn();
  } catch (const std::exception &e) {
    ++g_failed;
    std::printf("  EXCEPTION %s\n", e.what());
  }
  std::printf("[ %s ] %s\n", g_failed == before ? " OK " : "FAIL", name);
  std::fflush(stdout);
}

// Configuration records in protobuf's two encodings: hand-assembled wire bytes
// with the field numbers of the reference's model.proto, a hand-written text
// file with comments / legacy fields / fields the records do not carry, round
// trips, malformed input, and layers / optimizers set up from the strings.
static void TestConfProto() {
  {
    // LayerConf { name: "fc" type: "singacpp_dense"
    //             dense_conf { num_output: 7 bias_term: false } }
    //   1:"fc"  2:"singacpp_dense"  117:{1:7 2:0}
    std::string wire = std::string("\x0a\x02" "fc" "\x12\x0e" "singacpp_dense", 20);
    wire += std::string("\xaa\x07\x04\x08\x07\x10\x00", 7);
    LayerConf conf;
    EXPECT(ParseFromString(wire, &conf));
    EXPECT(conf.name == "fc" && conf.type == "singacpp_dense");
    EXPECT(conf.dense_conf.num_output == 7 && !conf.dense_conf.bias_term);
    EXPECT(SerializeAsString(conf) == wire);
    auto layer = CreateLayer("singacpp_dense");
    layer->Setup(Shape{5}, wire);
    EXPECT(layer->GetOutputSampleShape() == Shape{7});
    EXPECT(layer->name() == "fc");
    LayerConf back;
    EXPECT(ParseFromString(layer->ToProtoStr(), &back));
    EXPECT(back.dense_conf.num_output == 7 && back.name == "fc");
  }
  {
    const char *text =
        "# a convolution as a reference user writes it\n"
        "name: 'conv1'  type: \"singacpp_convolution\"\n"
        "param { name: \"conv1_w\" lr_mult: 1 decay_mult: 1\n"
        "        filler { type: 'gaussian' std: 1e-2 }\n"
        "        regularizer { coefficient: 5e-4 } }\n"
        "param < name: \"conv1_b\", lr_mult: 2.0f; decay_mult: 0 >\n"
        "convolution_conf {\n"
        "  num_output: 32 kernel_size: 5 stride: 1 pad: 2\n"
        "  weight_filler { type: \"xavier\" }   # not carried by the records\n"
        "  engine: CUDNN\n"
        "}\n"
        "bottom: \"data\" top: \"conv1\"\n";
    LayerConf conf;
    std::string err;
    std::vector<std::string> unknown;
    EXPECT(ParseFromText(text, &conf, &err, &unknown));
    EXPECT(err.empty());
    EXPECT(conf.name == "conv1" && conf.param.size() == 2);
    EXPECT(conf.param[0].filler.type == "gaussian");
    EXPECT_NEAR(conf.param[0].filler.std, 0.01, 1e-6);
    EXPECT(conf.param[0].has_regularizer && !conf.param[0].has_constraint);
    EXPECT_NEAR(conf.param[0].regularizer_coefficient, 5e-4, 1e-6);
    EXPECT_NEAR(conf.param[1].lr_mult, 2.0, 0);
    EXPECT_NEAR(conf.param[1].decay_mult, 0.0, 0);
    const auto &cc = conf.convolution_conf;
    EXPECT(cc.num_output == 32 && cc.kernel_h == 5 && cc.kernel_w == 5);
    EXPECT(cc.pad_h == 2 && cc.pad_w == 2 && cc.stride_h == 1);
    EXPECT(unknown.size() == 4 &&
           unknown[0] == "convolution_conf.weight_filler" &&
           unknown[1] == "convolution_conf.engine" && unknown[2] == "bottom");
    // text -> record -> text -> record and record -> wire -> record agree
    LayerConf again, wired;
    EXPECT(ParseFromText(DebugString(conf), &again, &err));
    EXPECT(DebugString(again) == DebugString(conf));
    EXPECT(ParseFromString(SerializeAsString(conf), &wired));
    EXPECT(DebugString(wired) == DebugString(conf));
    EXPECT(DebugString(conf).find("kernel_h: 5") != std::string::npos);
    auto layer = CreateLayer("singacpp_convolution");
    layer->Setup(Shape{3, 16, 16}, conf);
    EXPECT((layer->GetOutputSampleShape() == Shape{32, 16, 16}));
  }
  {
    OptimizerConf oc;
    std::string err;
    EXPECT(ParseFromText("type: \"rmsprop\" delta: 1e-6 regularizer { "
                         "type: \"l2\" coefficient: 0.25 }", &oc, &err));
    EXPECT(oc.type == "rmsprop" && oc.has_regularizer && !oc.has_constraint);
    EXPECT_NEAR(oc.rho, 0.95, 1e-6);  // the message's default
    EXPECT_NEAR(oc.regularizer.coefficient, 0.25, 0);
    SGD sgd;
    OptimizerConf sc;
    sc.momentum = 0.5f;
    sgd.Setup(SerializeAsString(sc));
    Accuracy acc;
    MetricConf mc;
    mc.top_k = 3;
    acc.Setup(SerializeAsString(mc));
    PoolingConf pc;
    LayerConf lc;
    lc.pooling_conf.pool = PoolingConf::AVE;
    lc.pooling_conf.set_kernel(3);
    lc.slice_conf.slice_point = {2, 5};
    lc.batchnorm_conf.factor = 0.1;
    lc.rnn_conf.rnn_mode = "lstm";
    LayerConf l2;
    EXPECT(ParseFromString(SerializeAsString(lc), &l2));
    EXPECT(l2.pooling_conf.pool == PoolingConf::AVE &&
           l2.pooling_conf.kernel_w == 3);
    EXPECT((l2.slice_conf.slice_point == std::vector<int>{2, 5}));
    EXPECT(l2.batchnorm_conf.factor == 0.1 && l2.rnn_conf.rnn_mode == "lstm");
    EXPECT(DebugString(lc).find("pool: AVE") != std::string::npos);
    // packed form of a repeated scalar
    SliceConf unused;
    (void)unused;
    LayerConf l3;
    EXPECT(ParseFromString(std::string("\xf2\x07\x04\x12\x02\x02\x05", 7), &l3));
    EXPECT((l3.slice_conf.slice_point == std::vector<int>{2, 5}));
  }
  {
    // malformed input is refused, never read past its end
    LayerConf conf;
    std::string err;
    EXPECT(!ParseFromString(std::string("\x0a\x05" "ab", 4), &conf));
    EXPECT(!ParseFromString(std::string("\x0a", 1), &conf));
    EXPECT(!ParseFromString(std::string("\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff", 11), &conf));
    EXPECT(!ParseFromString(std::string("\x0b\x00", 2), &conf));  // group
    EXPECT(!ParseFromText("name: \"unterminated", &conf, &err));
    EXPECT(err.find("line 1") == 0);
    EXPECT(!ParseFromText("name: \"a\"\nconvolution_conf { num_output: x }",
                          &conf, &err));
    EXPECT(err.find("line 2") == 0);
    EXPECT(!ParseFromText("dense_conf { num_output: 3", &conf, &err));
    EXPECT(!ParseFromText("dense_conf { num_output: -3 }", &conf, &err));
    std::string deep;
    for (int i = 0; i < 100; ++i) deep += "param { filler { ";
    EXPECT(!ParseFromText(deep, &conf, &err));
    std::string nest = "\x08\x01";
    for (int i = 0; i < 100; ++i) {
      std::string outer = "\x32";  // field 6, length-delimited
      outer.push_back((char)(nest.size() & 0x7f) | (nest.size() > 127 ? 0x80 : 0));
      if (nest.size() > 127) outer.push_back((char)(nest.size() >> 7));
      nest = outer + nest;
    }
    ParseFromString(nest, &conf);  // ends, whatever it returns
    EXPECT(true);
  }
  {
    // files
    const std::string path = "/tmp/singa_conf_proto_test.conf";
    LayerConf conf, back;
    conf.name = "relu\t\"1\"";
    conf.relu_conf.negative_slope = 0.1f;
    EXPECT(WriteProtoToTextFile(conf, path));
    EXPECT(ReadProtoFromTextFile(path, &back));
    EXPECT(back.name == conf.name &&
           back.relu_conf.negative_slope == conf.relu_conf.negative_slope);
    EXPECT(WriteProtoToBinaryFile(conf, path));
    LayerConf bin;
    EXPECT(ReadProtoFromBinaryFile(path, &bin) && bin.name == conf.name);
    std::remove(path.c_str());
    EXPECT(!ReadProtoFromTextFile(path, &back));
  }
}

// JPEG record codec: a smooth colour image survives quality-100 coding within
// a couple of grey levels, the label rides in the comment segment, grey
// images and odd sizes work, and a damaged stream is refused.
static void TestJpeg() {
  const size_t H = 37, W = 53;
  std::vector<float> px(H * W * 3);
  for (size_t h = 0; h < H; ++h)
    for (size_t w = 0; w < W; ++w) {
      px[(h * W + w) * 3 + 0] = 255.f * w / (W - 1);
      px[(h * W + w) * 3 + 1] = 255.f * h / (H - 1);
      px[(h * W + w) * 3 + 2] = 128.f + 100.f * std::sin(0.2f * (h + w));
    }
  JPGEncoder enc;
  JPGDecoder dec;
  EncoderConf ec;
  ec.type = "jpg";
  ec.image_dim_order = "HWC";
  enc.Setup(ec);
  DecoderConf dc;
  dc.image_dim_order = "HWC";
  dec.Setup(dc);
  std::vector<Tensor> rec{FromVec(px, {H, W, 3}, defaultDevice),
                          FromInts({7}, {1}, defaultDevice)};
  std::string bytes = enc.Encode(rec);
  EXPECT(bytes.size() > 100 && (unsigned char)bytes[0] == 0xff &&
         (unsigned char)bytes[1] == 0xd8);
  auto back = dec.Decode(bytes);
  EXPECT(back.size() == 2 && back[1].ToVector<int>()[0] == 7);
  EXPECT(back[0].shape() == (Shape{H, W, 3}));
  auto got = ToVec(back[0]);
  float worst = 0.f;
  for (size_t i = 0; i < px.size(); ++i)
    worst = std::max(worst, std::fabs(got[i] - std::round(px[i])));
  EXPECT(worst <= 3.f);
  // the reference's record format: an ImageRecord message around the file,
  // channel 0 stored as the file's blue plane; same tensors back
  {
    JPGEncoder penc;
    EncoderConf pc;  // type "jpg2proto"
    penc.Setup(pc);
    std::string wire = penc.Encode(rec);
    ImageRecord ir;
    EXPECT(ParseFromString(wire, &ir));
    EXPECT(ir.shape == (std::vector<int>{(int)H, (int)W, 3}) &&
           ir.label == std::vector<int>{7});
    EXPECT(ir.pixel.size() > 100 && (unsigned char)ir.pixel[0] == 0xff &&
           (unsigned char)ir.pixel[1] == 0xd8);
    std::vector<uint8_t> raw;
    int rw = 0, rh = 0, rc = 0;
    EXPECT(jpeg::Decode(ir.pixel, &raw, &rw, &rh, &rc) && rc == 3);
    // pixel (0, W-1): channel 0 (255) sits in the file's blue plane, channel
    // 2 (about 45) in its red plane
    EXPECT(raw[(W - 1) * 3 + 2] >= 250 &&
           std::abs((int)raw[(W - 1) * 3 + 0] - 45) <= 4);
    auto pback = dec.Decode(wire);
    EXPECT(pback.size() == 2 && pback[1].ToVector<int>()[0] == 7);
    auto pgot = ToVec(pback[0]);
    float pworst = 0.f;
    for (size_t i = 0; i < px.size(); ++i)
      pworst = std::max(pworst, std::fabs(pgot[i] - std::round(px[i])));
    EXPECT(pworst <= 3.f);
    // a record without a label, and one that is neither format
    std::vector<Tensor> bare{rec[0]};
    EXPECT(dec.Decode(penc.Encode(bare)).size() == 1);
    bool refused = false;
    try {
      dec.Decode(std::string("\x08\x01\x10\x02", 4));
    } catch (const std::exception &) {
      refused = true;
    }
    EXPECT(refused);
    EXPECT(CreateEncoder("jpg2proto") != nullptr &&
           CreateDecoder("proto2jpg") != nullptr);
  }
  // grey, CHW order, lower quality: smaller file, bounded error
  std::vector<float> grey(H * W);
  for (size_t i = 0; i < grey.size(); ++i) grey[i] = px[i * 3];
  JPGEncoder genc;
  EncoderConf gc;
  gc.type = "jpg";
  gc.image_dim_order = "CHW";
  genc.Setup(gc);
  genc.set_quality(75);
  std::vector<Tensor> grec{FromVec(grey, {H, W}, defaultDevice)};
  std::string gbytes = genc.Encode(grec);
  EXPECT(gbytes.size() < bytes.size());
  auto gback = JPGDecoder().Decode(gbytes);
  EXPECT(gback.size() == 1 && gback[0].shape() == (Shape{1, H, W}));
  auto gg = ToVec(gback[0]);
  worst = 0.f;
  for (size_t i = 0; i < grey.size(); ++i)
    worst = std::max(worst, std::fabs(gg[i] - grey[i]));
  EXPECT(worst <= 12.f);
  // factory names and damaged input
  EXPECT(CreateEncoder("jpg") != nullptr && CreateDecoder("JPEG") != nullptr);
  std::vector<uint8_t> o;
  int w = 0, h = 0, c = 0;
  std::string err;
  EXPECT(!jpeg::Decode(bytes.substr(0, bytes.size() / 2), &o, &w, &h, &c,
                       nullptr, &err));
  EXPECT(!err.empty());
  // malformed headers are rejected before any table or plane is touched
  auto find_marker = [&](const std::string &b, int marker) {
    for (size_t i = 2; i + 3 < b.size();) {
      if ((uint8_t)b[i] != 0xff) return std::string::npos;
      const int m = (uint8_t)b[i + 1];
      if (m == marker) return i;
      i += 2 + (((uint8_t)b[i + 2] << 8) | (uint8_t)b[i + 3]);
    }
    return std::string::npos;
  };
  const size_t sof = find_marker(bytes, 0xc0), sos = find_marker(bytes, 0xda);
  EXPECT(sof != std::string::npos && sos != std::string::npos);
  {
    std::string dup = bytes;          // two frame components with one id
    dup[sof + 4 + 6 + 3] = dup[sof + 4 + 6];
    err.clear();
    EXPECT(!jpeg::Decode(dup, &o, &w, &h, &c, nullptr, &err) && !err.empty());
  }
  {
    std::string huge = bytes;         // 65535 x 65535 pixels
    for (int k = 1; k <= 4; ++k) huge[sof + 4 + k] = (char)0xff;
    err.clear();
    EXPECT(!jpeg::Decode(huge, &o, &w, &h, &c, nullptr, &err) && !err.empty());
  }
  {
    std::string notab = bytes;        // the scan selects Huffman table 3
    notab[sos + 4 + 2] = (char)0x33;
    err.clear();
    EXPECT(!jpeg::Decode(notab, &o, &w, &h, &c, nullptr, &err) && !err.empty());
  }
  {
    std::string empty = bytes.substr(0, sos);   // scan header of length 2
    empty += std::string("\xff\xda\x00\x02", 4);
    err.clear();
    EXPECT(!jpeg::Decode(empty, &o, &w, &h, &c, nullptr, &err) && !err.empty());
  }
}

// `--jpeg-decode in.jpg out.ppm` / `--jpeg-encode in.ppm out.jpg [quality]` /
// `--jpeg-bench in.jpg repeats`: file converters and a decoder stopwatch used to check the codec against an independent
// implementation from the Python tests.
static int JpegTool(int argc, char **argv) {
  auto slurp = [](const char *path) {
    std::string s;
    if (FILE *f = std::fopen(path, "rb")) {
      char buf[65536];
      size_t n;
      while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, n);
      std::fclose(f);
    }
    return s;
  };
  auto spill = [](const char *path, const std::string &s) {
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    std::fwrite(s.data(), 1, s.size(), f);
    std::fclose(f);
    return true;
  };
  const std::string mode = argv[1];
  if (argc < 4) return 2;
  DecoderConf dc;
  dc.image_dim_order = "HWC";
  EncoderConf ec;
  ec.type = "jpg";
  ec.image_dim_order = "HWC";
  if (mode == "--jpeg-bench") {  // --jpeg-bench in.jpg <repeats>
    const std::string data = slurp(argv[2]);
    const int reps = std::max(1, std::atoi(argv[3]));
    std::vector<uint8_t> px;
    int w = 0, h = 0, c = 0;
    Timer t;
    for (int i = 0; i < reps; ++i)
      if (!jpeg::Decode(data, &px, &w, &h, &c)) return 1;
    std::printf("%dx%dx%d, %zu bytes: %.3f ms per decode\n", w, h, c,
                data.size(), t.Elapsed() / reps);
    return 0;
  }
  if (mode == "--jpeg-decode") {
    JPGDecoder dec;
    dec.Setup(dc);
    auto t = dec.Decode(slurp(argv[2]));
    ImageEncoder ppm;
    ppm.Setup(ec);
    return spill(argv[3], ppm.Encode(t)) ? 0 : 1;
  }
  ImageDecoder ppm;
  ppm.Setup(dc);
  auto t = ppm.Decode(slurp(argv[2]));
  JPGEncoder enc;
  if (mode == "--jpeg-record") ec.type = "jpg2proto";  // the reference's record
  enc.Setup(ec);
  if (argc > 4) enc.set_quality(std::atoi(argv[4]));
  return spill(argv[3], enc.Encode(t)) ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]).rfind("--jpeg-", 0) == 0) {
    // a file the codec rejects (a failed CHECK throws) is exit status 1; only a
    // fault inside the codec ends the tool by a signal
    try {
      return JpegTool(argc, argv);
    } catch (const std::exception &e) {
      std::fprintf(stderr, "rejected: %s\n", e.what());
      return 1;
    }
  }
  bool opencl = false;
  for (int i = 1; i < argc; ++i) {
    if (std::string(argv[i]) == "--gpu") g_gpu = true;
    if (std::string(argv[i]) == "--opencl") opencl = true;
  }
  if (opencl) {
    singa::LogToStderr();
    if (OpenclDevice::GetNumDevices() < 1) {
      std::printf("no OpenCL GPU device on this machine\n");
      return 2;
    }
    g_dev = std::make_shared<OpenclDevice>(0);
    Run("Opencl/device", [&] { TestOpenclSpecific(g_dev); });
    Run("TensorMath/opencl", [&] { TestTensorMath(g_dev); });
    Run("TensorFunctions/opencl", [&] { TestTensorFunctions(g_dev); });
    Run("LayersOpenclVsCpu", TestLayersGpuVsCpu);
    Run("Optimizers/opencl", [&] { TestOptimizers(g_dev); });
    Run("LossMetric/opencl", [&] { TestLossMetric(g_dev); });
    Run("DropoutAndMulti/opencl", [&] { TestDropoutAndMulti(g_dev); });
    Run("FeedForwardNet/opencl", [&] { TestFeedForwardNet(g_dev, false); });
    g_dev->Sync();
    std::printf("%d checks, %d failed\n", g_checks, g_failed);
    g_dev.reset();
    return g_failed ? 1 : 0;
  }
  singa::LogToStderr();
  auto host = defaultDevice;
  Run("MemPoolAndDevice", TestMemPoolAndDevice);
  Run("TensorMath/cpu", [&] { TestTensorMath(host); });
  Run("TensorFunctions/cpu", [&] { TestTensorFunctions(host); });
  Run("LayersGradient/cpu", TestLayersGradient);
  Run("DropoutAndMulti/cpu", [&] { TestDropoutAndMulti(host); });
  Run("Optimizers/cpu", [&] { TestOptimizers(host); });
  Run("LossMetric/cpu", [&] { TestLossMetric(host); });
  Run("FeedForwardNet/cpu", [&] { TestFeedForwardNet(host, false); });
  Run("FeedForwardNet/cpu-graph", [&] { TestFeedForwardNet(host, true); });
  Run("LocalUpdater", TestLocalUpdater);
  if (!g_gpu) Run("Rnn/cpu", TestRnn);
  Run("Io", TestIo);
  Run("ConfProto", TestConfProto);
  Run("Jpeg", TestJpeg);
  Run("EndPoint", TestEndPoint);
  if (g_gpu) {
    g_dev = Platform::CreateCudaGPUs(1)[0];
    Run("TensorMath/gpu", [&] { TestTensorMath(g_dev); });
    Run("TensorFunctions/gpu", [&] { TestTensorFunctions(g_dev); });
    Run("LayersGpuVsCpu", TestLayersGpuVsCpu);
    Run("Rnn/cpu+gpu", TestRnn);
    Run("DropoutAndMulti/gpu", [&] { TestDropoutAndMulti(g_dev); });
    Run("Optimizers/gpu", [&] { TestOptimizers(g_dev); });
    Run("LossMetric/gpu", [&] { TestLossMetric(g_dev); });
    Run("FeedForwardNet/gpu", [&] { TestFeedForwardNet(g_dev, false); });
    Run("FeedForwardNet/gpu-graph", [&] { TestFeedForwardNet(g_dev, true); });
    g_dev->Sync();
  }
  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
