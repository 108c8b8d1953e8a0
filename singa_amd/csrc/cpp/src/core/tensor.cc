#include "singa/core/tensor.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>

#include "../../../hip/kernels.h"
#include "../../../runtime/snapshot.h"
#include "opencl/runtime.h"

namespace singa {

namespace {

inline hipStream_t S(Context *ctx) { return static_cast<hipStream_t>(ctx->stream); }
#define SA_OK(expr)                                                     \
  do {                                                                  \
    CHECK_EQ((expr), 0) << "kernel launcher rejected arguments";        \
  } while (0)
// the OpenCL runtime of the executing device, or null on a HIP device
inline ocl::Runtime *CL(Context *ctx) { return static_cast<ocl::Runtime *>(ctx->cl); }

float cpu_unary(int op, float x, float a, float b) {
  using namespace sa;
  switch (op) {
    case U_COPY: return x;
    case U_ABS: return std::fabs(x);
    case U_EXP: return std::exp(x);
    case U_LOG: return std::log(x);
    case U_SQRT: return std::sqrt(x);
    case U_SQUARE: return x * x;
    case U_SIGN: return x > 0 ? 1.f : (x < 0 ? -1.f : 0.f);
    case U_CEIL: return std::ceil(x);
    case U_FLOOR: return std::floor(x);
    case U_ROUND: return std::round(x);          // halves away from zero
    case U_ROUNDE: return std::nearbyint(x);     // halves to even
    case U_ERF: return std::erf(x);
    case U_COS: return std::cos(x);
    case U_COSH: return std::cosh(x);
    case U_ACOS: return std::acos(x);
    case U_ACOSH: return std::acosh(x);
    case U_SIN: return std::sin(x);
    case U_SINH: return std::sinh(x);
    case U_ASIN: return std::asin(x);
    case U_ASINH: return std::asinh(x);
    case U_TAN: return std::tan(x);
    case U_ATAN: return std::atan(x);
    case U_ATANH: return std::atanh(x);
    case U_SOFTPLUS: return std::log1p(std::exp(x));
    case U_SOFTSIGN: return x / (1.f + std::fabs(x));
    case U_MAX_S: return std::max(x, a);
    case U_MIN_S: return std::min(x, a);
    case U_SIGMOID: return 1.f / (1.f + std::exp(-x));
    case U_TANH: return std::tanh(x);
    case U_RELU: return x > 0 ? x : 0.f;
    case U_CLAMP: return std::min(std::max(x, a), b);
    case U_ADD_S: return x + a;
    case U_SUB_S: return x - a;
    case U_MUL_S: return x * a;
    case U_DIV_S: return x / a;
    case U_POW_S: return std::pow(x, a);
    case U_RDIV_S: return a / x;
    case U_LT_S: return x < a ? 1.f : 0.f;
    case U_LE_S: return x <= a ? 1.f : 0.f;
    case U_GT_S: return x > a ? 1.f : 0.f;
    case U_GE_S: return x >= a ? 1.f : 0.f;
    case U_EQ_S: return x == a ? 1.f : 0.f;
    default: LOG(FATAL) << "unary op " << op << " not implemented on CppCPU";
  }
  return 0.f;
}

float cpu_binary(int op, float x, float y) {
  using namespace sa;
  switch (op) {
    case B_ADD: return x + y;
    case B_SUB: return x - y;
    case B_MUL: return x * y;
    case B_DIV: return x / y;
    case B_POW: return std::pow(x, y);
    case B_LT: return x < y ? 1.f : 0.f;
    case B_LE: return x <= y ? 1.f : 0.f;
    case B_GT: return x > y ? 1.f : 0.f;
    case B_GE: return x >= y ? 1.f : 0.f;
    case B_EQ: return x == y ? 1.f : 0.f;
    case B_MAX: return std::max(x, y);
    case B_MIN: return std::min(x, y);
    case B_RELU_BWD: return y > 0 ? x : 0.f;
    default: LOG(FATAL) << "binary op " << op << " not implemented";
  }
  return 0.f;
}

template <typename F>
void ForType(DataType t, F &&f) {
  if (t == kFloat32) f(static_cast<float *>(nullptr));
  else if (t == kInt) f(static_cast<int *>(nullptr));
  else LOG(FATAL) << "data type " << t << " is not supported on CppCPU";
}

// out = op(in) elementwise; in / out contiguous, possibly different dtypes
void Unary(int op, const Tensor &in, Tensor *out, float a = 0.f, float b = 0.f,
           const char *name = "Unary") {
  CHECK_EQ(in.Size(), out->Size());
  CHECK(!in.transpose()) << "make the operand Contiguous() first";
  const size_t n = in.Size();
  DataType ti = in.data_type(), to = out->data_type();
  Block *ib = in.block(), *ob = out->block();
  auto dev = in.device();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      ForType(ti, [&](auto *pi) {
        using TI = std::remove_pointer_t<decltype(pi)>;
        ForType(to, [&](auto *po) {
          using TO = std::remove_pointer_t<decltype(po)>;
          const TI *x = static_cast<const TI *>(ib->data());
          TO *y = static_cast<TO *>(ob->mutable_data());
          for (size_t i = 0; i < n; ++i)
            y[i] = static_cast<TO>(cpu_unary(op, static_cast<float>(x[i]), a, b));
        });
      });
    }, {ib}, {ob}, name);
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::unary(CL(ctx), op, ti, to, ib->data(), ob->mutable_data(),
                         nullptr, static_cast<int64_t>(n), a, b));
      else
        SA_OK(sa_unary(op, ti, to, ib->data(), ob->mutable_data(), nullptr,
                       static_cast<int64_t>(n), a, b, S(ctx)));
    }, {ib}, {ob}, name);
  }
}

// z = op(x, y) with a collapsed descriptor (nullptr = all contiguous)
void BinaryDesc(int op, const Tensor &x, const Tensor &y, Tensor *z,
                const sa::Desc *d, const char *name) {
  DataType t = x.data_type();
  CHECK_EQ(t, y.data_type());
  CHECK_EQ(t, z->data_type());
  const size_t n = z->Size();
  Block *xb = x.block(), *yb = y.block(), *zb = z->block();
  auto dev = x.device();
  sa::Desc desc;
  bool has_desc = d != nullptr;
  if (has_desc) desc = *d;
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      ForType(t, [&](auto *p) {
        using T = std::remove_pointer_t<decltype(p)>;
        const T *a = static_cast<const T *>(xb->data());
        const T *b = static_cast<const T *>(yb->data());
        T *c = static_cast<T *>(zb->mutable_data());
        for (size_t i = 0; i < n; ++i) {
          int64_t o0 = i, o1 = i, o2 = i;
          if (has_desc) {
            o0 = o1 = o2 = 0;
            int64_t lin = i;
            for (int k = desc.ndim - 1; k >= 0; --k) {
              int64_t idx = lin % desc.shape[k];
              lin /= desc.shape[k];
              o0 += idx * desc.s0[k];
              o1 += idx * desc.s1[k];
              o2 += idx * desc.s2[k];
            }
          }
          c[o2] = static_cast<T>(cpu_binary(op, static_cast<float>(a[o0]),
                                            static_cast<float>(b[o1])));
        }
      });
    }, {xb, yb}, {zb}, name);
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::binary(CL(ctx), op, t, xb->data(), yb->data(),
                          zb->mutable_data(), has_desc ? &desc : nullptr,
                          static_cast<int64_t>(n)));
      else
        SA_OK(sa_binary(op, t, xb->data(), yb->data(), zb->mutable_data(),
                        has_desc ? &desc : nullptr, static_cast<int64_t>(n),
                        S(ctx)));
    }, {xb, yb}, {zb}, name);
  }
}

void Binary(int op, const Tensor &x, const Tensor &y, Tensor *z,
            const char *name) {
  CHECK(x.shape() == y.shape()) << "element-wise operands differ in shape";
  CHECK(!x.transpose() && !y.transpose());
  BinaryDesc(op, x, y, z, nullptr, name);
}

// M = op(M, v) with v broadcast along rows (row=true: v has ncol entries)
void RowCol(int op, const Tensor &v, Tensor *M, bool row, const char *name) {
  CHECK_EQ(M->nDim(), 2u);
  const int64_t nr = M->shape(0), nc = M->shape(1);
  CHECK_EQ(v.Size(), static_cast<size_t>(row ? nc : nr));
  sa::Desc d;
  d.ndim = 2;
  d.shape[0] = nr; d.shape[1] = nc;
  d.s0[0] = nc; d.s0[1] = 1;
  d.s1[0] = row ? 0 : 1; d.s1[1] = row ? 1 : 0;
  d.s2[0] = nc; d.s2[1] = 1;
  BinaryDesc(op, *M, v, M, &d, name);
}

// strided copy described by (outer, len, inner) with separate pitches
void Copy3(const Tensor &src, size_t src_off, size_t src_pitch, Tensor *dst,
           size_t dst_off, size_t dst_pitch, size_t outer, size_t len,
           size_t inner, const char *name) {
  DataType t = src.data_type();
  CHECK_EQ(t, dst->data_type());
  Block *sb = src.block(), *db = dst->block();
  auto dev = src.device();
  const size_t es = SizeOf(t);
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const char *s = static_cast<const char *>(sb->data()) + src_off * es;
      char *d = static_cast<char *>(db->mutable_data()) + dst_off * es;
      for (size_t o = 0; o < outer; ++o)
        std::memcpy(d + o * dst_pitch * es, s + o * src_pitch * es,
                    len * inner * es);
    }, {sb}, {db}, name);
  } else {
    sa::Desc d;
    d.ndim = 2;
    d.shape[0] = outer; d.shape[1] = len * inner;
    d.s0[0] = src_pitch; d.s0[1] = 1;
    d.s1[0] = 0; d.s1[1] = 0;
    d.s2[0] = dst_pitch; d.s2[1] = 1;
    dev->Exec([=](Context *ctx) {
      const char *from = static_cast<const char *>(sb->data()) + src_off * es;
      char *to = static_cast<char *>(db->mutable_data()) + dst_off * es;
      const int64_t count = static_cast<int64_t>(outer * len * inner);
      if (CL(ctx))
        SA_OK(ocl::unary(CL(ctx), sa::U_COPY, t, t, from, to, &d, count, 0.f,
                         0.f));
      else
        SA_OK(sa_unary(sa::U_COPY, t, t, from, to, &d, count, 0.f, 0.f,
                       S(ctx)));
    }, {sb}, {db}, name);
  }
}

float ReduceAll(int op, const Tensor &in) {
  CHECK_EQ(in.data_type(), kFloat32);
  const size_t n = in.Size();
  auto dev = in.device();
  Block *ib = in.block();
  if (dev->lang() == kCpp) {
    dev->Sync();
    const float *x = static_cast<const float *>(ib->data());
    double acc = 0;
    for (size_t i = 0; i < n; ++i)
      acc += op == sa::R_SUM ? x[i]
                             : (op == sa::R_ASUM ? std::fabs(x[i]) : x[i] * x[i]);
    return static_cast<float>(acc);
  }
  CHECK(!dev->graph_enabled())
      << "host-visible reductions cannot be recorded in a graph";
  Tensor tmp(Shape{1}, dev, kFloat32);
  Block *tb = tmp.block();
  dev->Exec([=](Context *ctx) {
    if (CL(ctx))
      SA_OK(ocl::reduce_all(CL(ctx), op, kFloat32, ib->data(),
                            tb->mutable_data(), static_cast<int64_t>(n)));
    else
      SA_OK(sa_reduce_all(op, kFloat32, ib->data(), tb->mutable_data(),
                          static_cast<int64_t>(n), S(ctx)));
  }, {ib}, {tb}, "ReduceAll");
  float v = 0;
  tmp.GetValue(&v, 1);
  return v;
}

}  // namespace

// ---- Tensor -------------------------------------------------------------------
Tensor::~Tensor() { SetBlock(nullptr); }
Tensor::Tensor() : device_(defaultDevice) {}
Tensor::Tensor(const Shape &shape, DataType dtype)
    : Tensor(shape, defaultDevice, dtype) {}
Tensor::Tensor(const Shape &shape, std::shared_ptr<Device> dev, DataType dtype)
    : data_type_(dtype), device_(dev), shape_(shape) {
  ContiguousStrides();
  size_t size = Product(shape_) * SizeOf(data_type_);
  if (size) SetBlock(device_->NewBlock(size));
}
Tensor::Tensor(const Tensor &in)
    : data_type_(in.data_type_), device_(in.device_), shape_(in.shape_),
      stride_(in.stride_) {
  SetBlock(in.block_);
}
Tensor::Tensor(Tensor &&in)
    : data_type_(in.data_type_), device_(in.device_), block_(in.block_),
      shape_(std::move(in.shape_)), stride_(std::move(in.stride_)) {
  in.block_ = nullptr;
}
Tensor &Tensor::operator=(const Tensor &in) {
  if (this == &in) return *this;
  data_type_ = in.data_type_;
  device_ = in.device_;
  shape_ = in.shape_;
  stride_ = in.stride_;
  SetBlock(in.block_);
  return *this;
}
Tensor &Tensor::operator=(Tensor &&in) {
  if (this == &in) return *this;
  SetBlock(nullptr);
  data_type_ = in.data_type_;
  device_ = in.device_;
  shape_ = std::move(in.shape_);
  stride_ = std::move(in.stride_);
  block_ = in.block_;
  in.block_ = nullptr;
  return *this;
}

void Tensor::SetBlock(Block *b) {
  if (b) b->IncRefCount();
  if (block_ && block_->DecRefCount() == 0) device_->FreeBlock(block_);
  block_ = b;
}

void Tensor::ContiguousStrides() {
  stride_.assign(shape_.size(), 1);
  for (int i = static_cast<int>(shape_.size()) - 2; i >= 0; --i)
    stride_[i] = stride_[i + 1] * static_cast<int>(shape_[i + 1]);
}

// True for every view that is not row-major: permuted axes as well as
// broadcast (stride 0) ones.  Extents of 1 do not count.
bool Tensor::transpose() const {
  int64_t want = 1;
  for (int i = static_cast<int>(shape_.size()) - 1; i >= 0; --i) {
    if (shape_[i] != 1 && stride_[i] != want) return true;
    want *= static_cast<int64_t>(shape_[i]);
  }
  return false;
}

Tensor &Tensor::Reshape(const Shape &shape) {
  if (Product(shape) != Product(shape_)) {
    SetBlock(device_->NewBlock(Product(shape) * SizeOf(data_type_)));
  } else if (transpose()) {
    *this = Contiguous();
  }
  shape_ = shape;
  ContiguousStrides();
  return *this;
}
Tensor &Tensor::Resize(const Shape &shape) { return Reshape(shape); }

Tensor &Tensor::ResetLike(const Tensor &t) {
  if (block_ == nullptr || device_ != t.device_ || MemSize() != t.MemSize()) {
    device_ = t.device_;
    data_type_ = t.data_type_;
    SetBlock(t.MemSize() ? device_->NewBlock(t.MemSize()) : nullptr);
  }
  data_type_ = t.data_type_;
  shape_ = t.shape_;
  stride_ = t.stride_;
  return *this;
}

Tensor Tensor::AsType(DataType type) const {
  if (type == data_type_) return *this;
  Tensor out(shape_, device_, type);
  if (initialized()) Unary(sa::U_COPY, Contiguous(), &out, 0, 0, "CastType");
  return out;
}

Tensor &Tensor::ToType(DataType type) {
  if (type != data_type_) *this = AsType(type);
  return *this;
}

Tensor &Tensor::ToDevice(std::shared_ptr<Device> dev) {
  if (dev == device_) return *this;
  Tensor out(shape_, dev, data_type_);
  if (initialized()) {
    Tensor src = Contiguous();
    out.CopyData(src);
  }
  out.stride_ = out.stride_;
  *this = out;
  return *this;
}
Tensor &Tensor::ToHost() { return ToDevice(device_->host()); }

Tensor Tensor::Clone(std::shared_ptr<Device> device) const {
  if (!device) device = device_;
  Tensor t(shape_, device, data_type_);
  if (initialized()) t.CopyData(Contiguous());
  return t;
}

Tensor Tensor::T() const {
  CHECK_EQ(shape_.size(), 2u) << "T() is for matrices";
  Tensor t(*this);
  std::swap(t.shape_[0], t.shape_[1]);
  std::swap(t.stride_[0], t.stride_[1]);
  return t;
}

Tensor &Tensor::Transpose() {
  std::reverse(shape_.begin(), shape_.end());
  std::reverse(stride_.begin(), stride_.end());
  return *this;
}

Tensor &Tensor::Transpose(const std::vector<size_t> &axes) {
  CHECK_EQ(axes.size(), shape_.size()) << "one axis per dimension";
  Shape s(shape_.size());
  std::vector<int> st(shape_.size());
  std::vector<bool> seen(shape_.size(), false);
  for (size_t i = 0; i < axes.size(); ++i) {
    CHECK_LT(axes[i], shape_.size());
    CHECK(!seen[axes[i]]) << "axis " << axes[i] << " named twice";
    seen[axes[i]] = true;
    s[i] = shape_[axes[i]];
    st[i] = stride_[axes[i]];
  }
  shape_ = s;
  stride_ = st;
  return *this;
}

// Right-aligned broadcasting: an extent of 1 and every added leading axis
// get stride 0.  The last `ignore_last_dim` axes keep their own extents (the
// matrix axes of a batched product).
Tensor &Tensor::Broadcast(const Shape &shape, int ignore_last_dim) {
  const int m = static_cast<int>(shape_.size());
  const int n = static_cast<int>(shape.size());
  CHECK_GE(n, m) << "broadcasting never drops axes";
  CHECK_GE(ignore_last_dim, 0);
  Shape s(n);
  std::vector<int> st(n);
  for (int i = 0; i < n; ++i) {            // i counts from the right
    const int to = n - 1 - i, from = m - 1 - i;
    if (from < 0) {
      s[to] = shape[to];
      st[to] = 0;
    } else if (i < ignore_last_dim || shape_[from] == shape[to]) {
      s[to] = shape_[from];
      st[to] = stride_[from];
    } else {
      CHECK_EQ(shape_[from], 1u)
          << "extent " << shape_[from] << " does not broadcast to "
          << shape[to];
      s[to] = shape[to];
      st[to] = 0;
    }
  }
  shape_ = s;
  stride_ = st;
  return *this;
}

Tensor Tensor::Contiguous() const {
  if (!transpose()) {
    Tensor same(*this);
    same.ContiguousStrides();   // extents of 1 may carry any stride
    return same;
  }
  CHECK_LE(shape_.size(), 8u) << "views have at most 8 axes";
  Tensor out(shape_, device_, data_type_);
  if (Size() == 0) return out;
  sa::Desc d;
  d.ndim = static_cast<int>(shape_.size());
  for (int i = 0; i < d.ndim; ++i) {
    d.shape[i] = static_cast<int64_t>(shape_[i]);
    d.s0[i] = stride_[i];
    d.s1[i] = 0;
    d.s2[i] = out.stride_[i];
  }
  Block *ib = block_, *ob = out.block();
  DataType t = data_type_;
  size_t n = Size();
  if (device_->lang() == kCpp) {
    device_->Exec([=](Context *) {
      const char *x = static_cast<const char *>(ib->data());
      char *y = static_cast<char *>(ob->mutable_data());
      const size_t es = SizeOf(t);
      for (size_t i = 0; i < n; ++i) {
        int64_t lin = static_cast<int64_t>(i), from = 0;
        for (int k = d.ndim - 1; k >= 0; --k) {
          from += (lin % d.shape[k]) * d.s0[k];
          lin /= d.shape[k];
        }
        std::memcpy(y + i * es, x + from * es, es);
      }
    }, {ib}, {ob}, "Contiguous");
  } else {
    device_->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::unary(CL(ctx), sa::U_COPY, t, t, ib->data(),
                         ob->mutable_data(), &d, static_cast<int64_t>(n), 0,
                         0));
      else
        SA_OK(sa_unary(sa::U_COPY, t, t, ib->data(), ob->mutable_data(), &d,
                       static_cast<int64_t>(n), 0, 0, S(ctx)));
    }, {ib}, {ob}, "Contiguous");
  }
  return out;
}

// Elements (or slices along `axis`) repeated: repeats holds one count for all
// slices or one per slice; axis == Noaxis repeats the elements of the
// flattened tensor.  Reference: Tensor::Repeat, src/core/tensor/tensor.cc.
Tensor Tensor::Repeat(const std::vector<size_t> &repeats, int axis,
                      std::shared_ptr<Device> device) const {
  if (!device) device = device_;
  CHECK(!repeats.empty());
  Tensor src = Contiguous();
  Shape s;
  size_t total = 0;
  if (axis == Noaxis) {
    CHECK_EQ(repeats.size(), 1u) << "one count for the flattened tensor";
    total = repeats[0] * Size();
    s = Shape{total};
  } else {
    if (axis < 0) axis += static_cast<int>(nDim());
    CHECK(axis >= 0 && axis < static_cast<int>(nDim())) << "no such axis";
    CHECK(repeats.size() == 1 || repeats.size() == shape_[axis])
        << "one count, or one per slice";
    s = shape_;
    if (repeats.size() == 1) {
      s[axis] = shape_[axis] * repeats[0];
    } else {
      s[axis] = 0;
      for (size_t r : repeats) s[axis] += r;
    }
    total = Product(s);
  }
  Tensor out(s, device_, data_type_);
  if (total) RepeatDataToFrom(repeats.size() == 1, repeats, axis, &out, src,
                              total);
  if (device != device_) out.ToDevice(device);
  return out;
}

template <typename SType>
void Tensor::SetValue(SType x) {
  CHECK(block_ != nullptr);
  Block *b = block_;
  const size_t n = Size();
  DataType t = data_type_;
  const float v = static_cast<float>(x);
  if (device_->lang() == kCpp) {
    device_->Exec([=](Context *) {
      ForType(t, [&](auto *p) {
        using T = std::remove_pointer_t<decltype(p)>;
        T *y = static_cast<T *>(b->mutable_data());
        std::fill(y, y + n, static_cast<T>(v));
      });
    }, {}, {b}, "SetValue");
  } else {
    device_->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::fill(CL(ctx), t, b->mutable_data(), static_cast<int64_t>(n),
                        v));
      else
        SA_OK(sa_fill(t, b->mutable_data(), static_cast<int64_t>(n), v,
                      S(ctx)));
    }, {}, {b}, "SetValue");
  }
}
template void Tensor::SetValue<float>(float);
template void Tensor::SetValue<int>(int);
template void Tensor::SetValue<double>(double);

template <typename SType>
void Tensor::CopyDataFromHostPtr(const SType *src, size_t num, size_t offset) {
  CHECK_EQ(sizeof(SType), SizeOf(data_type_))
      << "host element size does not match the tensor's data type";
  CHECK(src != nullptr);
  device_->CopyDataFromHostPtr(block_, src, sizeof(SType) * num,
                               sizeof(SType) * offset);
}
template void Tensor::CopyDataFromHostPtr<float>(const float *, size_t, size_t);
template void Tensor::CopyDataFromHostPtr<int>(const int *, size_t, size_t);
template void Tensor::CopyDataFromHostPtr<unsigned char>(const unsigned char *,
                                                         size_t, size_t);

void Tensor::CopyData(const Tensor &src) {
  CHECK_EQ(Size(), src.Size());
  CHECK(block_ != nullptr);
  if (src.block()) CopyDataToFrom(this, src, Size(), 0, 0);
}

template <typename SType>
void Tensor::GetValue(SType *value, size_t num) const {
  CHECK_EQ(sizeof(SType), SizeOf(data_type_));
  CHECK(!transpose()) << "GetValue needs a contiguous tensor";
  device_->CopyDataToHostPtr(value, block_, sizeof(SType) * num, 0);
}
template void Tensor::GetValue<float>(float *, size_t) const;
template void Tensor::GetValue<int>(int *, size_t) const;

void Tensor::ToProto(::singa_amd::SnapshotTensor *proto) const {
  Tensor h = Contiguous().Clone(device_->host());
  proto->shape.assign(shape_.begin(), shape_.end());
  proto->stride.assign(h.stride_.begin(), h.stride_.end());
  proto->dtype = data_type_;
  h.device()->Sync();
  proto->data.assign(static_cast<const char *>(h.block()->data()), h.MemSize());
}

void Tensor::FromProto(const ::singa_amd::SnapshotTensor &proto) {
  Shape s(proto.shape.begin(), proto.shape.end());
  data_type_ = static_cast<DataType>(proto.dtype);
  shape_ = s;
  ContiguousStrides();
  SetBlock(device_->NewBlock(Product(s) * SizeOf(data_type_)));
  CHECK_EQ(proto.data.size(), MemSize());
  device_->CopyDataFromHostPtr(block_, proto.data.data(), proto.data.size(), 0);
}

#define SINGA_INPLACE(op, code, name)                    \
  Tensor &Tensor::operator op(const Tensor &in) {        \
    Binary(code, *this, in, this, name);                 \
    return *this;                                        \
  }
SINGA_INPLACE(+=, sa::B_ADD, "Add")
SINGA_INPLACE(-=, sa::B_SUB, "Sub")
SINGA_INPLACE(*=, sa::B_MUL, "EltwiseMult")
SINGA_INPLACE(/=, sa::B_DIV, "Div")
#undef SINGA_INPLACE

#define SINGA_INPLACE_S(op, code, name)                             \
  template <typename SType>                                         \
  Tensor &Tensor::operator op(SType x) {                            \
    Unary(code, *this, this, static_cast<float>(x), 0.f, name);     \
    return *this;                                                   \
  }                                                                 \
  template Tensor &Tensor::operator op<float>(float);               \
  template Tensor &Tensor::operator op<int>(int);                   \
  template Tensor &Tensor::operator op<double>(double);
SINGA_INPLACE_S(+=, sa::U_ADD_S, "AddScalar")
SINGA_INPLACE_S(-=, sa::U_SUB_S, "SubScalar")
SINGA_INPLACE_S(*=, sa::U_MUL_S, "MultScalar")
SINGA_INPLACE_S(/=, sa::U_DIV_S, "DivScalar")
#undef SINGA_INPLACE_S

float Tensor::L1() const { return ReduceAll(sa::R_ASUM, *this) / Size(); }
float Tensor::L2() const {
  return std::sqrt(ReduceAll(sa::R_SUMSQ, *this)) / Size();
}

// ---- free functions ------------------------------------------------------------
void CopyDataToFrom(Tensor *dst, const Tensor &src, size_t num,
                    size_t dst_offset, size_t src_offset) {
  CHECK_EQ(dst->data_type(), src.data_type());
  size_t w = SizeOf(src.data_type());
  CHECK_GE(src.MemSize(), (src_offset + num) * w);
  CHECK_GE(dst->MemSize(), (dst_offset + num) * w);
  dst->device()->CopyDataToFrom(dst->block(), src.block(), num * w,
                                dst_offset * w, src_offset * w);
}

Tensor Reshape(const Tensor &in, const Shape &s) {
  Tensor out(in);
  out.Reshape(s);
  return out;
}

#define SINGA_UNARY_DEF(fn, code)                                     \
  void fn(const Tensor &in, Tensor *out) {                            \
    Unary(code, in, out, 0.f, 0.f, #fn);                              \
  }                                                                   \
  Tensor fn(const Tensor &in) {                                       \
    Tensor out(in.shape(), in.device(), in.data_type());              \
    fn(in, &out);                                                     \
    return out;                                                       \
  }
SINGA_UNARY_DEF(Abs, sa::U_ABS)
SINGA_UNARY_DEF(Exp, sa::U_EXP)
SINGA_UNARY_DEF(Log, sa::U_LOG)
SINGA_UNARY_DEF(ReLU, sa::U_RELU)
SINGA_UNARY_DEF(Sigmoid, sa::U_SIGMOID)
SINGA_UNARY_DEF(Sign, sa::U_SIGN)
SINGA_UNARY_DEF(Sqrt, sa::U_SQRT)
SINGA_UNARY_DEF(Square, sa::U_SQUARE)
SINGA_UNARY_DEF(Tanh, sa::U_TANH)
SINGA_UNARY_DEF(Erf, sa::U_ERF)
SINGA_UNARY_DEF(Ceil, sa::U_CEIL)
SINGA_UNARY_DEF(Floor, sa::U_FLOOR)
SINGA_UNARY_DEF(Round, sa::U_ROUND)
SINGA_UNARY_DEF(RoundE, sa::U_ROUNDE)
SINGA_UNARY_DEF(SoftPlus, sa::U_SOFTPLUS)
SINGA_UNARY_DEF(SoftSign, sa::U_SOFTSIGN)
SINGA_UNARY_DEF(Cos, sa::U_COS)
SINGA_UNARY_DEF(Cosh, sa::U_COSH)
SINGA_UNARY_DEF(Acos, sa::U_ACOS)
SINGA_UNARY_DEF(Acosh, sa::U_ACOSH)
SINGA_UNARY_DEF(Sin, sa::U_SIN)
SINGA_UNARY_DEF(Sinh, sa::U_SINH)
SINGA_UNARY_DEF(Asin, sa::U_ASIN)
SINGA_UNARY_DEF(Asinh, sa::U_ASINH)
SINGA_UNARY_DEF(Tan, sa::U_TAN)
SINGA_UNARY_DEF(Atan, sa::U_ATAN)
SINGA_UNARY_DEF(Atanh, sa::U_ATANH)
#undef SINGA_UNARY_DEF

#define SINGA_BINARY_DEF(opname, fn, code)                            \
  void fn(const Tensor &a, const Tensor &b, Tensor *out) {            \
    Binary(code, a, b, out, #fn);                                     \
  }                                                                   \
  Tensor operator opname(const Tensor &a, const Tensor &b) {          \
    Tensor out(a.shape(), a.device(), a.data_type());                 \
    fn(a, b, &out);                                                   \
    return out;                                                       \
  }
SINGA_BINARY_DEF(+, Add, sa::B_ADD)
SINGA_BINARY_DEF(-, Sub, sa::B_SUB)
SINGA_BINARY_DEF(*, EltwiseMult, sa::B_MUL)
SINGA_BINARY_DEF(/, Div, sa::B_DIV)
#undef SINGA_BINARY_DEF

Tensor operator<(const Tensor &a, const Tensor &b) {
  Tensor out(a.shape(), a.device(), a.data_type());
  Binary(sa::B_LT, a, b, &out, "LT");
  return out;
}
Tensor operator>(const Tensor &a, const Tensor &b) {
  Tensor out(a.shape(), a.device(), a.data_type());
  Binary(sa::B_GT, a, b, &out, "GT");
  return out;
}

Tensor ReLUBackward(const Tensor &dy, const Tensor &x) {
  Tensor out(dy.shape(), dy.device(), dy.data_type());
  Binary(sa::B_RELU_BWD, dy, x, &out, "ReLUBackward");
  return out;
}

#define SINGA_SCALAR_DEF(sig, code, name)                             \
  template <typename SType>                                           \
  Tensor sig(const Tensor &in, SType x) {                             \
    Tensor out(in.shape(), in.device(), in.data_type());              \
    Unary(code, in, &out, static_cast<float>(x), 0.f, name);          \
    return out;                                                       \
  }                                                                   \
  template Tensor sig<float>(const Tensor &, float);                  \
  template Tensor sig<int>(const Tensor &, int);                      \
  template Tensor sig<double>(const Tensor &, double);
SINGA_SCALAR_DEF(operator+, sa::U_ADD_S, "AddScalar")
SINGA_SCALAR_DEF(operator-, sa::U_SUB_S, "SubScalar")
SINGA_SCALAR_DEF(operator*, sa::U_MUL_S, "MultScalar")
SINGA_SCALAR_DEF(operator/, sa::U_DIV_S, "DivScalar")
SINGA_SCALAR_DEF(operator<, sa::U_LT_S, "LTScalar")
SINGA_SCALAR_DEF(operator<=, sa::U_LE_S, "LEScalar")
SINGA_SCALAR_DEF(operator>, sa::U_GT_S, "GTScalar")
SINGA_SCALAR_DEF(operator>=, sa::U_GE_S, "GEScalar")
SINGA_SCALAR_DEF(Pow, sa::U_POW_S, "PowScalar")
#undef SINGA_SCALAR_DEF

template <typename SType>
void Add(const Tensor &in, SType x, Tensor *out) {
  Unary(sa::U_ADD_S, in, out, static_cast<float>(x), 0.f, "AddScalar");
}
template void Add<float>(const Tensor &, float, Tensor *);
template <typename SType>
void EltwiseMult(const Tensor &in, SType x, Tensor *out) {
  Unary(sa::U_MUL_S, in, out, static_cast<float>(x), 0.f, "MultScalar");
}
template void EltwiseMult<float>(const Tensor &, float, Tensor *);
template <typename SType>
Tensor Div(SType x, const Tensor &in) {
  Tensor out(in.shape(), in.device(), in.data_type());
  Unary(sa::U_RDIV_S, in, &out, static_cast<float>(x), 0.f, "RDivScalar");
  return out;
}
template Tensor Div<float>(float, const Tensor &);

template <typename SType>
SType Sum(const Tensor &in) {
  return static_cast<SType>(ReduceAll(sa::R_SUM, in));
}
template float Sum<float>(const Tensor &);

void SumRows(const Tensor &M, Tensor *v) {
  CHECK_EQ(M.nDim(), 2u);
  CHECK(!M.transpose());
  const int64_t nr = M.shape(0), nc = M.shape(1);
  CHECK_EQ(v->Size(), static_cast<size_t>(nc));
  Block *mb = M.block(), *vb = v->block();
  auto dev = M.device();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *m = static_cast<const float *>(mb->data());
      float *o = static_cast<float *>(vb->mutable_data());
      std::fill(o, o + nc, 0.f);
      for (int64_t r = 0; r < nr; ++r)
        for (int64_t c = 0; c < nc; ++c) o[c] += m[r * nc + c];
    }, {mb}, {vb}, "SumRows");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::reduce_cols(CL(ctx), kFloat32, mb->data(),
                               vb->mutable_data(), nr, nc));
      else
        SA_OK(sa_reduce_cols(kFloat32, mb->data(), vb->mutable_data(), nr, nc,
                             S(ctx)));
    }, {mb}, {vb}, "SumRows");
  }
}

static void ReduceRowsOp(int op, const Tensor &M, Tensor *v, const char *name) {
  CHECK_EQ(M.nDim(), 2u);
  CHECK(!M.transpose());
  const int64_t nr = M.shape(0), nc = M.shape(1);
  CHECK_EQ(v->Size(), static_cast<size_t>(nr));
  Block *mb = M.block(), *vb = v->block();
  auto dev = M.device();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *m = static_cast<const float *>(mb->data());
      float *o = static_cast<float *>(vb->mutable_data());
      for (int64_t r = 0; r < nr; ++r) {
        float acc = op == sa::R_MAX ? m[r * nc] : 0.f;
        for (int64_t c = 0; c < nc; ++c)
          acc = op == sa::R_MAX ? std::max(acc, m[r * nc + c])
                                : acc + m[r * nc + c];
        o[r] = acc;
      }
    }, {mb}, {vb}, name);
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::reduce_rows(CL(ctx), op, kFloat32, mb->data(),
                               vb->mutable_data(), nr, nc));
      else
        SA_OK(sa_reduce_rows(op, kFloat32, mb->data(), vb->mutable_data(), nr,
                             nc, S(ctx)));
    }, {mb}, {vb}, name);
  }
}

void SumColumns(const Tensor &M, Tensor *v) {
  ReduceRowsOp(sa::R_SUM, M, v, "SumColumns");
}

Tensor RowMax(const Tensor &in) {
  Tensor out(Shape{in.shape(0)}, in.device(), in.data_type());
  ReduceRowsOp(sa::R_MAX, in, &out, "RowMax");
  return out;
}

Tensor Sum(const Tensor &in, int axis) {
  CHECK_EQ(in.nDim(), 2u);
  if (axis == 0) {
    Tensor out(Shape{in.shape(1)}, in.device(), in.data_type());
    SumRows(in, &out);
    return out;
  }
  CHECK_EQ(axis, 1);
  Tensor out(Shape{in.shape(0)}, in.device(), in.data_type());
  SumColumns(in, &out);
  return out;
}

Tensor Average(const Tensor &in, int axis) {
  Tensor s = Sum(in, axis);
  return s / static_cast<float>(in.shape(axis));
}

void SoftMax(const Tensor &in, Tensor *out) {
  CHECK_LE(in.nDim(), 2u);
  const int64_t rows = in.nDim() == 2 ? in.shape(0) : 1;
  const int64_t cols = in.Size() / rows;
  Block *ib = in.block(), *ob = out->block();
  auto dev = in.device();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *x = static_cast<const float *>(ib->data());
      float *y = static_cast<float *>(ob->mutable_data());
      for (int64_t r = 0; r < rows; ++r) {
        float mx = x[r * cols];
        for (int64_t c = 1; c < cols; ++c) mx = std::max(mx, x[r * cols + c]);
        float sum = 0;
        for (int64_t c = 0; c < cols; ++c)
          sum += (y[r * cols + c] = std::exp(x[r * cols + c] - mx));
        for (int64_t c = 0; c < cols; ++c) y[r * cols + c] /= sum;
      }
    }, {ib}, {ob}, "SoftMax");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::softmax_fwd(CL(ctx), kFloat32, ib->data(),
                               ob->mutable_data(), rows, cols));
      else
        SA_OK(sa_softmax_fwd(kFloat32, ib->data(), ob->mutable_data(), rows,
                             cols, S(ctx)));
    }, {ib}, {ob}, "SoftMax");
  }
}

Tensor SoftMax(const Tensor &in) {
  Tensor out(in.shape(), in.device(), in.data_type());
  SoftMax(in, &out);
  return out;
}

Tensor SoftMaxBackward(const Tensor &dy, const Tensor &y) {
  const int64_t rows = y.nDim() == 2 ? y.shape(0) : 1;
  const int64_t cols = y.Size() / rows;
  Tensor dx(y.shape(), y.device(), y.data_type());
  Block *gb = dy.block(), *yb = y.block(), *xb = dx.block();
  auto dev = y.device();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *g = static_cast<const float *>(gb->data());
      const float *p = static_cast<const float *>(yb->data());
      float *o = static_cast<float *>(xb->mutable_data());
      for (int64_t r = 0; r < rows; ++r) {
        float dot = 0;
        for (int64_t c = 0; c < cols; ++c) dot += g[r * cols + c] * p[r * cols + c];
        for (int64_t c = 0; c < cols; ++c)
          o[r * cols + c] = p[r * cols + c] * (g[r * cols + c] - dot);
      }
    }, {gb, yb}, {xb}, "SoftMaxBackward");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::softmax_bwd(CL(ctx), kFloat32, gb->data(), yb->data(),
                               xb->mutable_data(), rows, cols));
      else
        SA_OK(sa_softmax_bwd(kFloat32, gb->data(), yb->data(),
                             xb->mutable_data(), rows, cols, S(ctx)));
    }, {gb, yb}, {xb}, "SoftMaxBackward");
  }
  return dx;
}

void AddColumn(const Tensor &v, Tensor *M) { RowCol(sa::B_ADD, v, M, false, "AddColumn"); }
void AddRow(const Tensor &v, Tensor *M) { RowCol(sa::B_ADD, v, M, true, "AddRow"); }
void SubColumn(const Tensor &v, Tensor *M) { RowCol(sa::B_SUB, v, M, false, "SubColumn"); }
void SubRow(const Tensor &v, Tensor *M) { RowCol(sa::B_SUB, v, M, true, "SubRow"); }
void MultColumn(const Tensor &v, Tensor *M) { RowCol(sa::B_MUL, v, M, false, "MultColumn"); }
void MultRow(const Tensor &v, Tensor *M) { RowCol(sa::B_MUL, v, M, true, "MultRow"); }
void DivColumn(const Tensor &v, Tensor *M) { RowCol(sa::B_DIV, v, M, false, "DivColumn"); }
void DivRow(const Tensor &v, Tensor *M) { RowCol(sa::B_DIV, v, M, true, "DivRow"); }

template <typename SType>
void Mult(SType alpha_, const Tensor &A, const Tensor &B, SType beta_,
          Tensor *C) {
  CHECK_EQ(A.nDim(), 2u);
  const float alpha = static_cast<float>(alpha_), beta = static_cast<float>(beta_);
  Tensor Bm = B;
  bool vec = B.nDim() == 1;
  if (vec) Bm = Reshape(B, Shape{B.Size(), 1});
  CHECK_EQ(Bm.nDim(), 2u);
  const int64_t M = A.shape(0), K = A.shape(1), N = Bm.shape(1);
  CHECK_EQ(static_cast<int64_t>(Bm.shape(0)), K) << "inner dimensions differ";
  CHECK_EQ(C->Size(), static_cast<size_t>(M * N));
  const int64_t ars = A.stride()[0], acs = A.stride()[1];
  const int64_t brs = Bm.stride()[0], bcs = Bm.stride()[1];
  Block *ab = A.block(), *bb = Bm.block(), *cb = C->block();
  auto dev = A.device();
  DataType t = A.data_type();
  std::vector<Block *> reads{ab, bb};
  if (beta != 0.f) reads.push_back(cb);
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *a = static_cast<const float *>(ab->data());
      const float *b = static_cast<const float *>(bb->data());
      float *c = static_cast<float *>(cb->mutable_data());
      for (int64_t i = 0; i < M; ++i) {
        float *ci = c + i * N;
        for (int64_t j = 0; j < N; ++j) ci[j] = beta == 0.f ? 0.f : beta * ci[j];
        for (int64_t k = 0; k < K; ++k) {
          const float av = alpha * a[i * ars + k * acs];
          const float *bk = b + k * brs;
          for (int64_t j = 0; j < N; ++j) ci[j] += av * bk[j * bcs];
        }
      }
    }, reads, {cb}, "Mult");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::gemm(CL(ctx), t, M, N, K, alpha, ab->data(), ars, acs,
                        bb->data(), brs, bcs, beta, cb->mutable_data(), N));
      else
        SA_OK(sa_gemm(t, M, N, K, alpha, ab->data(), ars, acs, 0, bb->data(),
                      brs, bcs, 0, beta, cb->mutable_data(), N, 0, 1, nullptr,
                      0, S(ctx)));
    }, reads, {cb}, "Mult");
  }
}
template void Mult<float>(float, const Tensor &, const Tensor &, float, Tensor *);

void Mult(const Tensor &A, const Tensor &B, Tensor *C) {
  Mult<float>(1.f, A, B, 0.f, C);
}
Tensor Mult(const Tensor &A, const Tensor &B) {
  Shape s{A.shape(0)};
  if (B.nDim() == 2) s.push_back(B.shape(1));
  Tensor C(s, A.device(), A.data_type());
  Mult(A, B, &C);
  return C;
}

template <typename SType>
void Axpy(SType alpha_, const Tensor &in, Tensor *out) {
  CHECK_EQ(in.Size(), out->Size());
  const float alpha = static_cast<float>(alpha_);
  const size_t n = in.Size();
  Block *ib = in.block(), *ob = out->block();
  auto dev = in.device();
  DataType t = in.data_type();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *x = static_cast<const float *>(ib->data());
      float *y = static_cast<float *>(ob->mutable_data());
      for (size_t i = 0; i < n; ++i) y[i] += alpha * x[i];
    }, {ib, ob}, {ob}, "Axpy");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::axpy(CL(ctx), t, alpha, ib->data(), ob->mutable_data(),
                        static_cast<int64_t>(n)));
      else
        SA_OK(sa_axpy(t, nullptr, alpha, ib->data(), ob->mutable_data(),
                      static_cast<int64_t>(n), S(ctx)));
    }, {ib, ob}, {ob}, "Axpy");
  }
}
template void Axpy<float>(float, const Tensor &, Tensor *);

// ---- random ---------------------------------------------------------------------
namespace {
template <typename Gen>
void CpuRandom(Tensor *out, const char *name, Gen gen) {
  Block *ob = out->block();
  const size_t n = out->Size();
  auto dev = out->device();
  auto *cpu = dynamic_cast<CppCPU *>(dev.get());
  CHECK(cpu != nullptr);
  dev->Exec([=](Context *) {
    float *y = static_cast<float *>(ob->mutable_data());
    auto *rng = cpu->rng();
    for (size_t i = 0; i < n; ++i) y[i] = gen(*rng);
  }, {}, {ob}, name);
}
}  // namespace

template <typename SType>
void Bernoulli(SType p_, Tensor *out) {
  const float p = static_cast<float>(p_);
  if (out->device()->lang() == kCpp) {
    CpuRandom(out, "Bernoulli", [p](std::mt19937_64 &g) {
      return std::uniform_real_distribution<float>(0.f, 1.f)(g) < p ? 1.f : 0.f;
    });
    return;
  }
  Block *ob = out->block();
  const size_t n = out->Size();
  DataType t = out->data_type();
  out->device()->Exec([=](Context *ctx) {
    if (CL(ctx))
      SA_OK(ocl::rng_bernoulli(CL(ctx), t, ob->mutable_data(),
                               static_cast<int64_t>(n), p));
    else
      SA_OK(sa_rng_bernoulli(t, ob->mutable_data(), static_cast<int64_t>(n), p,
                             ctx->rng_state, S(ctx)));
  }, {}, {ob}, "Bernoulli");
}
template void Bernoulli<float>(float, Tensor *);

template <typename SType>
void Gaussian(SType mean_, SType std_, Tensor *out) {
  const float mean = static_cast<float>(mean_), sd = static_cast<float>(std_);
  if (out->device()->lang() == kCpp) {
    CpuRandom(out, "Gaussian", [mean, sd](std::mt19937_64 &g) {
      return std::normal_distribution<float>(mean, sd)(g);
    });
    return;
  }
  Block *ob = out->block();
  const size_t n = out->Size();
  DataType t = out->data_type();
  out->device()->Exec([=](Context *ctx) {
    if (CL(ctx))
      SA_OK(ocl::rng_normal(CL(ctx), t, ob->mutable_data(),
                            static_cast<int64_t>(n), mean, sd));
    else
      SA_OK(sa_rng_normal(t, ob->mutable_data(), static_cast<int64_t>(n), mean,
                          sd, ctx->rng_state, S(ctx)));
  }, {}, {ob}, "Gaussian");
}
template void Gaussian<float>(float, float, Tensor *);

template <typename SType>
void Uniform(SType low_, SType high_, Tensor *out) {
  const float lo = static_cast<float>(low_), hi = static_cast<float>(high_);
  if (out->device()->lang() == kCpp) {
    CpuRandom(out, "Uniform", [lo, hi](std::mt19937_64 &g) {
      return std::uniform_real_distribution<float>(lo, hi)(g);
    });
    return;
  }
  Block *ob = out->block();
  const size_t n = out->Size();
  DataType t = out->data_type();
  out->device()->Exec([=](Context *ctx) {
    if (CL(ctx))
      SA_OK(ocl::rng_uniform(CL(ctx), t, ob->mutable_data(),
                             static_cast<int64_t>(n), lo, hi));
    else
      SA_OK(sa_rng_uniform(t, ob->mutable_data(), static_cast<int64_t>(n), lo,
                           hi, ctx->rng_state, S(ctx)));
  }, {}, {ob}, "Uniform");
}
template void Uniform<float>(float, float, Tensor *);

// ---- loss helpers -----------------------------------------------------------------
void ComputeCrossEntropy(const Tensor &p, const Tensor &t, Tensor *loss) {
  CHECK_LE(p.nDim(), 2u);
  CHECK_EQ(t.data_type(), kInt) << "labels are int32 class indices";
  const int64_t rows = p.nDim() == 2 ? p.shape(0) : 1;
  const int64_t cols = p.Size() / rows;
  CHECK_EQ(t.Size(), static_cast<size_t>(rows));
  CHECK_EQ(loss->Size(), static_cast<size_t>(rows));
  Block *pb = p.block(), *tb = t.block(), *lb = loss->block();
  auto dev = p.device();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      const float *pp = static_cast<const float *>(pb->data());
      const int *tt = static_cast<const int *>(tb->data());
      float *l = static_cast<float *>(lb->mutable_data());
      for (int64_t r = 0; r < rows; ++r) {
        CHECK_LT(tt[r], cols);
        l[r] = -std::log(std::max(pp[r * cols + tt[r]], 1.17549435e-38f));
      }
    }, {pb, tb}, {lb}, "ComputeCrossEntropy");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::cross_entropy_fwd(CL(ctx), pb->data(), tb->data(),
                                     lb->mutable_data(), rows, cols));
      else
        SA_OK(sa_cross_entropy_fwd(kFloat32, pb->data(), tb->data(), 1, kInt,
                                   lb->mutable_data(), rows, cols, S(ctx)));
    }, {pb, tb}, {lb}, "ComputeCrossEntropy");
  }
}

void SoftmaxCrossEntropyBwd(const Tensor &t, Tensor *p) {
  const int64_t rows = p->nDim() == 2 ? p->shape(0) : 1;
  const int64_t cols = p->Size() / rows;
  CHECK_EQ(t.data_type(), kInt);
  Block *pb = p->block(), *tb = t.block();
  auto dev = p->device();
  if (dev->lang() == kCpp) {
    dev->Exec([=](Context *) {
      float *pp = static_cast<float *>(pb->mutable_data());
      const int *tt = static_cast<const int *>(tb->data());
      for (int64_t r = 0; r < rows; ++r) pp[r * cols + tt[r]] -= 1.f;
    }, {pb, tb}, {pb}, "SoftmaxCrossEntropyBwd");
  } else {
    dev->Exec([=](Context *ctx) {
      if (CL(ctx))
        SA_OK(ocl::softmax_ce_bwd(CL(ctx), pb->mutable_data(), tb->data(),
                                  rows, cols));
      else
        SA_OK(sa_softmax_ce_bwd(kFloat32, pb->mutable_data(), tb->data(), 1,
                                kInt, rows, cols, S(ctx)));
    }, {pb, tb}, {pb}, "SoftmaxCrossEntropyBwd");
  }
}

// ---- layout ----------------------------------------------------------------------
Tensor ConcatOn(const std::vector<Tensor> &in, int axis) {
  CHECK(!in.empty());
  Shape s = in[0].shape();
  CHECK_LT(static_cast<size_t>(axis), s.size());
  size_t total = 0;
  for (const auto &t : in) {
    CHECK_EQ(t.nDim(), s.size());
    total += t.shape(axis);
  }
  size_t outer = 1, inner = 1;
  for (int i = 0; i < axis; ++i) outer *= s[i];
  for (size_t i = axis + 1; i < s.size(); ++i) inner *= s[i];
  s[axis] = total;
  Tensor out(s, in[0].device(), in[0].data_type());
  size_t off = 0;
  for (const auto &t : in) {
    Tensor c = t.Contiguous();
    size_t len = t.shape(axis);
    Copy3(c, 0, len * inner, &out, off * inner, total * inner, outer, len,
          inner, "ConcatOn");
    off += len;
  }
  return out;
}

Tensor SliceOn(const Tensor &in, size_t start, size_t end, int axis) {
  Shape s = in.shape();
  CHECK_LT(start, end);
  CHECK_LE(end, s[axis]);
  size_t outer = 1, inner = 1;
  for (int i = 0; i < axis; ++i) outer *= s[i];
  for (size_t i = axis + 1; i < s.size(); ++i) inner *= s[i];
  size_t len = end - start, full = s[axis];
  s[axis] = len;
  Tensor out(s, in.device(), in.data_type());
  Tensor c = in.Contiguous();
  Copy3(c, start * inner, full * inner, &out, 0, len * inner, outer, len,
        inner, "SliceOn");
  return out;
}

Tensor ConcatenateRows(const std::vector<Tensor> &in) { return ConcatOn(in, 0); }
Tensor ConcatenateColumns(const std::vector<Tensor> &in) { return ConcatOn(in, 1); }
Tensor CopyRows(const Tensor &in, size_t start, size_t end) {
  return SliceOn(in, start, end, 0);
}
Tensor SliceRows(const Tensor &in, size_t start, size_t end) {
  return SliceOn(in, start, end, 0);
}
Tensor CopyColumns(const Tensor &in, size_t start, size_t end) {
  return SliceOn(in, start, end, 1);
}

Tensor SliceColumns(const Tensor &in, size_t start, size_t end) {
  return SliceOn(in, start, end, 1);
}
Tensor ConcatRows(const std::vector<Tensor> &in) { return ConcatOn(in, 0); }
Tensor ConcatColumns(const std::vector<Tensor> &in) { return ConcatOn(in, 1); }

// ---- views and copies as free functions ----------------------------------------------
Tensor Contiguous(const Tensor &in) { return in.Contiguous(); }
Tensor Transform(const Tensor &in) { return in.Contiguous(); }
void Transform(const Tensor &in, Tensor *out) {
  CHECK_EQ(in.Size(), out->Size());
  out->CopyData(in.Contiguous());
}
Tensor Resize(const Tensor &in, const Shape &s) {
  Tensor out(in);
  out.Resize(s);
  return out;
}
Tensor Transpose(const Tensor &in) {
  Tensor out(in);
  out.Transpose();
  return out;
}
Tensor Transpose(const Tensor &in, const std::vector<size_t> &axes) {
  Tensor out(in);
  out.Transpose(axes);
  return out;
}
Tensor Broadcast(const Tensor &in, const Shape &shape, int ignore_last_dim) {
  Tensor out(in);
  out.Broadcast(shape, ignore_last_dim);
  return out;
}

// dst <- the elements (Noaxis) or the slices along `axis` of src, each
// `repeats[i]` times (`broadcast_flag`: repeats[0] for all).  `num` is the
// number of elements written.  Reference: RepeatDataToFrom,
// src/core/tensor/tensor.cc.
void RepeatDataToFrom(bool broadcast_flag, const std::vector<size_t> &repeats,
                      int axis, Tensor *dst, const Tensor &src, size_t num) {
  CHECK(!repeats.empty());
  CHECK(!src.transpose() && !dst->transpose());
  CHECK_EQ(src.data_type(), dst->data_type());
  CHECK_GE(dst->Size(), num);
  size_t outer = 1, slices = src.Size(), chunk = 1;
  if (axis != Noaxis) {
    if (axis < 0) axis += static_cast<int>(src.nDim());
    CHECK(axis >= 0 && axis < static_cast<int>(src.nDim()));
    for (int i = 0; i < axis; ++i) outer *= src.shape(i);
    slices = src.shape(axis);
    for (size_t i = axis + 1; i < src.nDim(); ++i) chunk *= src.shape(i);
  }
  if (!broadcast_flag) CHECK_EQ(repeats.size(), slices);
  size_t at = 0;
  for (size_t o = 0; o < outer; ++o)
    for (size_t i = 0; i < slices; ++i) {
      const size_t times = broadcast_flag ? repeats[0] : repeats[i];
      for (size_t r = 0; r < times; ++r) {
        CHECK_LE(at + chunk, num) << "more repeats than the destination holds";
        CopyDataToFrom(dst, src, chunk, at, (o * slices + i) * chunk);
        at += chunk;
      }
    }
  CHECK_EQ(at, num);
}

// ---- comparisons, powers ----------------------------------------------------------
#define SINGA_COMPARE_DEF(opname, fn, bcode, ucode)                    \
  void fn(const Tensor &a, const Tensor &b, Tensor *out) {              \
    Binary(bcode, a, b, out, #fn);                                      \
  }                                                                     \
  template <typename SType>                                             \
  void fn(const Tensor &in, SType x, Tensor *out) {                     \
    Unary(ucode, in, out, static_cast<float>(x), 0.f, #fn "Scalar");    \
  }                                                                     \
  template void fn<float>(const Tensor &, float, Tensor *);             \
  template void fn<int>(const Tensor &, int, Tensor *);                 \
  template void fn<double>(const Tensor &, double, Tensor *);
SINGA_COMPARE_DEF(<, LT, sa::B_LT, sa::U_LT_S)
SINGA_COMPARE_DEF(<=, LE, sa::B_LE, sa::U_LE_S)
SINGA_COMPARE_DEF(>, GT, sa::B_GT, sa::U_GT_S)
SINGA_COMPARE_DEF(>=, GE, sa::B_GE, sa::U_GE_S)
SINGA_COMPARE_DEF(==, EQ, sa::B_EQ, sa::U_EQ_S)
#undef SINGA_COMPARE_DEF

Tensor operator<=(const Tensor &a, const Tensor &b) {
  Tensor out(a.shape(), a.device(), a.data_type());
  LE(a, b, &out);
  return out;
}
Tensor operator>=(const Tensor &a, const Tensor &b) {
  Tensor out(a.shape(), a.device(), a.data_type());
  GE(a, b, &out);
  return out;
}
Tensor operator==(const Tensor &a, const Tensor &b) {
  Tensor out(a.shape(), a.device(), a.data_type());
  EQ(a, b, &out);
  return out;
}
template <typename SType>
Tensor operator==(const Tensor &in, SType x) {
  Tensor out(in.shape(), in.device(), in.data_type());
  EQ(in, x, &out);
  return out;
}
template Tensor operator==<float>(const Tensor &, float);
template Tensor operator==<int>(const Tensor &, int);
template Tensor operator==<double>(const Tensor &, double);

void Pow(const Tensor &base, const Tensor &exp, Tensor *out) {
  Binary(sa::B_POW, base, exp, out, "Pow");
}
Tensor Pow(const Tensor &base, const Tensor &exp) {
  Tensor out(base.shape(), base.device(), base.data_type());
  Pow(base, exp, &out);
  return out;
}
template <typename SType>
void Pow(const Tensor &in, SType x, Tensor *out) {
  Unary(sa::U_POW_S, in, out, static_cast<float>(x), 0.f, "PowScalar");
}
template void Pow<float>(const Tensor &, float, Tensor *);
template void Pow<int>(const Tensor &, int, Tensor *);
template void Pow<double>(const Tensor &, double, Tensor *);

void ReLUBackward(const Tensor &dy, const Tensor &x, Tensor *out) {
  Binary(sa::B_RELU_BWD, dy, x, out, "ReLUBackward");
}

template <typename SType>
void Sub(const Tensor &in, SType x, Tensor *out) {
  Unary(sa::U_SUB_S, in, out, static_cast<float>(x), 0.f, "SubScalar");
}
template void Sub<float>(const Tensor &, float, Tensor *);
template <typename SType>
void Div(const Tensor &in, SType x, Tensor *out) {
  Unary(sa::U_DIV_S, in, out, static_cast<float>(x), 0.f, "DivScalar");
}
template void Div<float>(const Tensor &, float, Tensor *);
template <typename SType>
void Div(SType x, const Tensor &in, Tensor *out) {
  Unary(sa::U_RDIV_S, in, out, static_cast<float>(x), 0.f, "RDivScalar");
}
template void Div<float>(float, const Tensor &, Tensor *);

// ---- reductions, rows and columns -----------------------------------------------------
Tensor SumAll(const Tensor &in) {
  Tensor flat = Reshape(in.Contiguous(), Shape{1, in.Size()});
  Tensor out(Shape{1}, in.device(), in.data_type());
  SumColumns(flat, &out);
  return out;
}

// M = alpha * (v along the columns / rows) + beta * M
template <typename SType>
void AddColumn(SType alpha, SType beta, const Tensor &v, Tensor *M) {
  if (static_cast<float>(beta) != 1.f) *M *= static_cast<float>(beta);
  AddColumn(static_cast<float>(alpha) == 1.f ? v : v * static_cast<float>(alpha),
            M);
}
template void AddColumn<float>(float, float, const Tensor &, Tensor *);
template <typename SType>
void AddRow(SType alpha, SType beta, const Tensor &v, Tensor *M) {
  if (static_cast<float>(beta) != 1.f) *M *= static_cast<float>(beta);
  AddRow(static_cast<float>(alpha) == 1.f ? v : v * static_cast<float>(alpha),
         M);
}
template void AddRow<float>(float, float, const Tensor &, Tensor *);

// soft-max over the axes from `axis` on: the tensor is read as a matrix of
// prod(shape[:axis]) rows (reference: SoftMax(in, axis), tensor.cc)
static Shape RowsFrom(const Tensor &in, int axis) {
  if (axis < 0) axis += static_cast<int>(in.nDim());
  CHECK(axis >= 0 && axis <= static_cast<int>(in.nDim()));
  size_t rows = 1;
  for (int i = 0; i < axis; ++i) rows *= in.shape(i);
  return Shape{rows, rows ? in.Size() / rows : 0};
}
void SoftMax(const Tensor &in, Tensor *out, int axis) {
  Tensor x = Reshape(in.Contiguous(), RowsFrom(in, axis));
  Tensor y = Reshape(*out, x.shape());
  SoftMax(x, &y);
}
Tensor SoftMax(const Tensor &in, int axis) {
  Tensor out(in.shape(), in.device(), in.data_type());
  SoftMax(in, &out, axis);
  return out;
}
Tensor SoftMaxBackward(const Tensor &dy, int axis, const Tensor &y) {
  const Shape m = RowsFrom(y, axis);
  Tensor dx = SoftMaxBackward(Reshape(dy.Contiguous(), m),
                              Reshape(y.Contiguous(), m));
  dx.Reshape(y.shape());
  return dx;
}

// out += alpha[0] * in with the factor on the device
void Axpy(const Tensor &alpha, const Tensor &in, Tensor *out) {
  CHECK_EQ(alpha.Size(), 1u);
  CHECK_EQ(in.Size(), out->Size());
  CHECK(!in.transpose() && !out->transpose());
  Tensor scaled(in.shape(), in.device(), in.data_type());
  sa::Desc d;
  d.ndim = 1;
  d.shape[0] = static_cast<int64_t>(in.Size());
  d.s0[0] = 1; d.s1[0] = 0; d.s2[0] = 1;
  BinaryDesc(sa::B_MUL, in, alpha, &scaled, &d, "AxpyScale");
  BinaryDesc(sa::B_ADD, *out, scaled, out, nullptr, "Axpy");
}

Tensor CrossEntropyFwd(const Tensor &p, const Tensor &t) {
  Tensor loss(Shape{p.nDim() == 2 ? p.shape(0) : 1}, p.device(),
              p.data_type());
  ComputeCrossEntropy(p, t, &loss);
  return loss;
}
Tensor SoftmaxCrossEntropyBwd(const Tensor &p, const Tensor &t) {
  Tensor g = p.Clone();
  SoftmaxCrossEntropyBwd(t, &g);
  return g;
}

}  // namespace singa
