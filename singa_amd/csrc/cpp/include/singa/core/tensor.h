// Tensor of the C++ API: shape + strides + data type over a ref-counted Block
// on a Device.  Copies are shallow.  Operations are free functions that submit
// closures to the tensor's device; on HipGPU they launch the hand-written
// gfx950 kernels, on CppCPU plain loops.
//
// API parity: reference include/singa/core/tensor.h:55-664.  float32 and int32
// are supported everywhere, float16 on the GPU elementwise / GEMM / conv paths.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "device.h"

namespace singa_amd { struct SnapshotTensor; }

namespace singa {

class Tensor {
 public:
  ~Tensor();
  Tensor();
  explicit Tensor(const Shape &shape, DataType dtype = kFloat32);
  Tensor(const Shape &shape, std::shared_ptr<Device> dev,
         DataType dtype = kFloat32);
  Tensor(const Tensor &from);
  Tensor(Tensor &&from);
  Tensor &operator=(const Tensor &in);
  Tensor &operator=(Tensor &&in);

  Block *block() const { return block_; }
  std::shared_ptr<Device> device() const { return device_; }
  template <typename SType>
  const SType *data() const {
    return static_cast<const SType *>(block()->data());
  }
  DataType data_type() const { return data_type_; }
  const Shape &shape() const { return shape_; }
  size_t shape(size_t idx) const { return shape_.at(idx); }
  size_t nDim() const { return shape_.size(); }
  bool empty() const { return nDim() == 0; }
  const std::vector<int> &stride() const { return stride_; }
  bool transpose() const;
  bool initialized() const { return block_ && block_->initialized(); }
  size_t Size() const { return block_ ? Product(shape_) : 0; }
  size_t MemSize() const { return block_ ? block_->size() : 0; }

  Tensor &Reshape(const Shape &shape);
  Tensor &Resize(const Shape &shape);
  Tensor &ResetLike(const Tensor &t);
  /// A copy in another element type (the tensor itself when it has it).
  Tensor AsType(DataType type) const;
  /// Converts this tensor.
  Tensor &ToType(DataType type);
  Tensor &ToDevice(std::shared_ptr<Device> dev);
  Tensor &ToHost();
  Tensor Clone(std::shared_ptr<Device> device = nullptr) const;
  /// 2-d transposed view (shares the block).
  Tensor T() const;
  /// Row-major copy of a strided view (the tensor itself when contiguous).
  Tensor Contiguous() const;
  /// Views over the same block: all axes reversed, axes permuted, extents of
  /// 1 (and added leading axes) repeated with stride 0.  Element-wise
  /// functions want Contiguous() operands; Mult reads 2-d views as they are.
  Tensor &Transpose();
  Tensor &Transpose(const std::vector<size_t> &axes);
  Tensor &Broadcast(const Shape &shape, int ignore_last_dim = 0);
  /// Slices along `axis` (elements of the flattened tensor for Noaxis)
  /// repeated `repeats[i]` times, or repeats[0] times each.
  Tensor Repeat(const std::vector<size_t> &repeats, int axis,
                std::shared_ptr<Device> device = nullptr) const;

  template <typename SType>
  void SetValue(SType x);
  template <typename SType>
  void CopyDataFromHostPtr(const SType *src, size_t num, size_t offset = 0);
  void CopyData(const Tensor &other);
  template <typename SType>
  void GetValue(SType *value, size_t num) const;
  template <typename SType>
  std::vector<SType> ToVector() const {
    std::vector<SType> v(Size());
    GetValue(v.data(), v.size());
    return v;
  }

  /// Serialised form used by Snapshot (float32 / int32).
  void ToProto(::singa_amd::SnapshotTensor *proto) const;
  void FromProto(const ::singa_amd::SnapshotTensor &proto);

  Tensor &operator+=(const Tensor &in);
  Tensor &operator-=(const Tensor &in);
  Tensor &operator*=(const Tensor &in);
  Tensor &operator/=(const Tensor &in);
  template <typename SType> Tensor &operator+=(SType x);
  template <typename SType> Tensor &operator-=(SType x);
  template <typename SType> Tensor &operator*=(SType x);
  template <typename SType> Tensor &operator/=(SType x);

  float L1() const;  // mean absolute value
  float L2() const;  // sqrt(sum of squares) / Size()

 private:
  void SetBlock(Block *b);
  void ContiguousStrides();
  DataType data_type_ = kFloat32;
  std::shared_ptr<Device> device_;
  Block *block_ = nullptr;
  Shape shape_;
  std::vector<int> stride_;
};

using TensorVec = std::vector<Tensor>;

/// `axis` value of Repeat / RepeatDataToFrom for "the flattened tensor".
constexpr int Noaxis = 9999;

inline void CheckDataTypeAndLang(const Tensor &in1, const Tensor &in2) {
  CHECK_EQ(in1.data_type(), in2.data_type());
  CHECK_EQ(in1.device()->lang(), in2.device()->lang());
}

template <typename FromType, typename ToType>
ToType TypeCast(const FromType &x) {
  return static_cast<ToType>(x);
}

void CopyDataToFrom(Tensor *dst, const Tensor &src, size_t num,
                    size_t dst_offset = 0, size_t src_offset = 0);
Tensor Reshape(const Tensor &in, const Shape &s);
Tensor Resize(const Tensor &in, const Shape &s);
Tensor Contiguous(const Tensor &in);
/// Views (see the member functions of the same names).
Tensor Transpose(const Tensor &in);
Tensor Transpose(const Tensor &in, const std::vector<size_t> &axes);
Tensor Broadcast(const Tensor &in, const Shape &shape, int ignore_last_dim = 0);
/// Row-major copy of a view.
Tensor Transform(const Tensor &in);
void Transform(const Tensor &in, Tensor *out);
void RepeatDataToFrom(bool broadcast_flag, const std::vector<size_t> &repeats,
                      int axis, Tensor *dst, const Tensor &src, size_t num);

// ---- element-wise ------------------------------------------------------------------
#define SINGA_UNARY_DECL(fn) \
  Tensor fn(const Tensor &in); \
  void fn(const Tensor &in, Tensor *out);
SINGA_UNARY_DECL(Abs)
SINGA_UNARY_DECL(Exp)
SINGA_UNARY_DECL(Log)
SINGA_UNARY_DECL(ReLU)
SINGA_UNARY_DECL(Sigmoid)
SINGA_UNARY_DECL(Sign)
SINGA_UNARY_DECL(Sqrt)
SINGA_UNARY_DECL(Square)
SINGA_UNARY_DECL(Tanh)
SINGA_UNARY_DECL(Erf)
SINGA_UNARY_DECL(Ceil)
SINGA_UNARY_DECL(Floor)
SINGA_UNARY_DECL(Round)    // halves away from zero
SINGA_UNARY_DECL(RoundE)   // halves to even
SINGA_UNARY_DECL(SoftPlus)
SINGA_UNARY_DECL(SoftSign)
SINGA_UNARY_DECL(Cos)
SINGA_UNARY_DECL(Cosh)
SINGA_UNARY_DECL(Acos)
SINGA_UNARY_DECL(Acosh)
SINGA_UNARY_DECL(Sin)
SINGA_UNARY_DECL(Sinh)
SINGA_UNARY_DECL(Asin)
SINGA_UNARY_DECL(Asinh)
SINGA_UNARY_DECL(Tan)
SINGA_UNARY_DECL(Atan)
SINGA_UNARY_DECL(Atanh)
#undef SINGA_UNARY_DECL

Tensor operator+(const Tensor &a, const Tensor &b);
Tensor operator-(const Tensor &a, const Tensor &b);
Tensor operator*(const Tensor &a, const Tensor &b);
Tensor operator/(const Tensor &a, const Tensor &b);
void Add(const Tensor &a, const Tensor &b, Tensor *out);
void Sub(const Tensor &a, const Tensor &b, Tensor *out);
void EltwiseMult(const Tensor &a, const Tensor &b, Tensor *out);
void Div(const Tensor &a, const Tensor &b, Tensor *out);
Tensor operator<(const Tensor &a, const Tensor &b);
Tensor operator>(const Tensor &a, const Tensor &b);
Tensor operator<=(const Tensor &a, const Tensor &b);
Tensor operator>=(const Tensor &a, const Tensor &b);
Tensor operator==(const Tensor &a, const Tensor &b);
/// 1 / 0 results in the operands' type, named forms with an output argument
void LT(const Tensor &a, const Tensor &b, Tensor *out);
void LE(const Tensor &a, const Tensor &b, Tensor *out);
void GT(const Tensor &a, const Tensor &b, Tensor *out);
void GE(const Tensor &a, const Tensor &b, Tensor *out);
void EQ(const Tensor &a, const Tensor &b, Tensor *out);
template <typename SType> void LT(const Tensor &in, SType x, Tensor *out);
template <typename SType> void LE(const Tensor &in, SType x, Tensor *out);
template <typename SType> void GT(const Tensor &in, SType x, Tensor *out);
template <typename SType> void GE(const Tensor &in, SType x, Tensor *out);
template <typename SType> void EQ(const Tensor &in, SType x, Tensor *out);
template <typename SType> Tensor operator==(const Tensor &in, SType x);
Tensor Pow(const Tensor &base, const Tensor &exp);
void Pow(const Tensor &base, const Tensor &exp, Tensor *out);
template <typename SType> void Pow(const Tensor &in, SType x, Tensor *out);
template <typename SType> void Sub(const Tensor &in, SType x, Tensor *out);
template <typename SType> void Div(const Tensor &in, SType x, Tensor *out);
template <typename SType> void Div(SType x, const Tensor &in, Tensor *out);

template <typename SType> Tensor operator+(const Tensor &in, SType x);
template <typename SType> Tensor operator-(const Tensor &in, SType x);
template <typename SType> Tensor operator*(const Tensor &in, SType x);
template <typename SType> Tensor operator/(const Tensor &in, SType x);
template <typename SType> Tensor operator<(const Tensor &in, SType x);
template <typename SType> Tensor operator<=(const Tensor &in, SType x);
template <typename SType> Tensor operator>(const Tensor &in, SType x);
template <typename SType> Tensor operator>=(const Tensor &in, SType x);
template <typename SType> Tensor Pow(const Tensor &in, SType x);
template <typename SType> void Add(const Tensor &in, SType x, Tensor *out);
template <typename SType> void EltwiseMult(const Tensor &in, SType x, Tensor *out);
/// out = x / in
template <typename SType> Tensor Div(SType x, const Tensor &in);
/// out = in2 > 0 ? in1 : 0
Tensor ReLUBackward(const Tensor &dy, const Tensor &x);
void ReLUBackward(const Tensor &dy, const Tensor &x, Tensor *out);

// ---- reductions --------------------------------------------------------------------
template <typename SType> SType Sum(const Tensor &in);
/// axis 0 sums the rows of a matrix (result: one row), axis 1 the columns.
Tensor Sum(const Tensor &in, int axis);
Tensor Average(const Tensor &in, int axis);
/// One-element tensor holding the sum of all elements (stays on the device).
Tensor SumAll(const Tensor &in);
Tensor RowMax(const Tensor &in);
Tensor SoftMax(const Tensor &in);
void SoftMax(const Tensor &in, Tensor *out);
/// dx from dy and y = softmax(x), row-wise.
Tensor SoftMaxBackward(const Tensor &dy, const Tensor &y);
/// Soft-max over the axes from `axis` on (rows = product of the axes before).
Tensor SoftMax(const Tensor &in, int axis);
void SoftMax(const Tensor &in, Tensor *out, int axis);
Tensor SoftMaxBackward(const Tensor &dy, int axis, const Tensor &y);

// ---- matrix / vector -----------------------------------------------------------------
void AddColumn(const Tensor &v, Tensor *M);
void AddRow(const Tensor &v, Tensor *M);
/// M = alpha * v (along the columns / rows) + beta * M
template <typename SType>
void AddColumn(SType alpha, SType beta, const Tensor &v, Tensor *M);
template <typename SType>
void AddRow(SType alpha, SType beta, const Tensor &v, Tensor *M);
void SubColumn(const Tensor &v, Tensor *M);
void SubRow(const Tensor &v, Tensor *M);
void MultColumn(const Tensor &v, Tensor *M);
void MultRow(const Tensor &v, Tensor *M);
void DivColumn(const Tensor &v, Tensor *M);
void DivRow(const Tensor &v, Tensor *M);
/// v[c] = sum_r M[r, c]
void SumRows(const Tensor &M, Tensor *v);
/// v[r] = sum_c M[r, c]
void SumColumns(const Tensor &M, Tensor *v);
Tensor Mult(const Tensor &A, const Tensor &B);
void Mult(const Tensor &A, const Tensor &B, Tensor *C);
/// C = alpha * A * B + beta * C  (A, B may be T() views)
template <typename SType>
void Mult(SType alpha, const Tensor &A, const Tensor &B, SType beta, Tensor *C);
/// out += alpha * in
template <typename SType>
void Axpy(SType alpha, const Tensor &in, Tensor *out);
/// The same with the factor in a one-element device tensor.
void Axpy(const Tensor &alpha, const Tensor &in, Tensor *out);

// ---- random ------------------------------------------------------------------------
template <typename SType> void Bernoulli(SType p, Tensor *out);
template <typename SType> void Gaussian(SType mean, SType std, Tensor *out);
template <typename SType> void Uniform(SType low, SType high, Tensor *out);

// ---- loss helpers --------------------------------------------------------------------
/// loss[i] = -log p[i, t[i]] for (batch, dim) probabilities and int labels.
void ComputeCrossEntropy(const Tensor &p, const Tensor &t, Tensor *loss);
/// p <- p - onehot(t), in place.
void SoftmaxCrossEntropyBwd(const Tensor &t, Tensor *p);
/// The two above returning their result.
Tensor CrossEntropyFwd(const Tensor &p, const Tensor &t);
Tensor SoftmaxCrossEntropyBwd(const Tensor &p, const Tensor &t);

// ---- layout --------------------------------------------------------------------------
Tensor ConcatenateRows(const std::vector<Tensor> &in);
Tensor ConcatenateColumns(const std::vector<Tensor> &in);
Tensor ConcatOn(const std::vector<Tensor> &in, int axis);
Tensor CopyRows(const Tensor &in, size_t start, size_t end);
Tensor CopyColumns(const Tensor &in, size_t start, size_t end);
Tensor SliceRows(const Tensor &in, size_t start, size_t end);
Tensor SliceOn(const Tensor &in, size_t start, size_t end, int axis);
Tensor SliceColumns(const Tensor &in, size_t start, size_t end);
Tensor ConcatRows(const std::vector<Tensor> &in);
Tensor ConcatColumns(const std::vector<Tensor> &in);

}  // namespace singa
