// Implementation of the flat C interface (include/singa/api/c_api.h) over the
// C++ tensor library.  Exceptions never cross the boundary: they are caught,
// stored per thread and reported through the return value.
#include "singa/api/c_api.h"

#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "singa/singa.h"

using singa::Shape;
using singa::Tensor;

struct singa_tensor_s {
  Tensor t;
};
struct singa_device_s {
  std::shared_ptr<singa::Device> d;
};

namespace {
thread_local std::string g_error;

template <typename F>
auto guarded(F &&f, decltype(f()) fail) -> decltype(f()) {
  try {
    g_error.clear();
    return f();
  } catch (const std::exception &e) {
    g_error = e.what();
  } catch (...) {
    g_error = "unknown error";
  }
  return fail;
}

Shape to_shape(const int64_t *shape, int ndim) {
  Shape s;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] < 0) throw std::invalid_argument("negative extent");
    s.push_back(static_cast<size_t>(shape[i]));
  }
  return s;
}

singa_tensor wrap(Tensor &&t) { return new singa_tensor_s{std::move(t)}; }

void need(const void *p, const char *what) {
  if (p == nullptr) throw std::invalid_argument(std::string("null ") + what);
}

template <typename Msg>
std::string transcode(const std::string &in, int in_format, int out_format) {
  Msg conf;
  if (in_format == 0) {
    if (!singa::ParseFromString(in, &conf))
      throw std::invalid_argument("malformed wire-format message");
  } else {
    std::string err;
    if (!singa::ParseFromText(in, &conf, &err))
      throw std::invalid_argument(err);
  }
  return out_format == 0 ? singa::SerializeAsString(conf)
                         : singa::DebugString(conf);
}
}  // namespace

extern "C" {

const char *singa_last_error(void) { return g_error.c_str(); }
const char *singa_version(void) { return "4.2.0"; }

int singa_num_gpus(void) {
  return guarded([] { return singa::Platform::GetNumGPUs(); }, -1);
}

singa_device singa_device_host(void) {
  return guarded([] { return new singa_device_s{singa::defaultDevice}; },
                 (singa_device) nullptr);
}

singa_device singa_device_gpu(int id) {
  return guarded(
      [&] {
        auto devs = singa::Platform::CreateCudaGPUsOn({id});
        if (devs.empty()) throw std::runtime_error("no such GPU");
        return new singa_device_s{devs[0]};
      },
      (singa_device) nullptr);
}

int singa_num_opencl_devices(void) {
  return guarded([&] { return singa::OpenclDevice::GetNumDevices(); }, 0);
}

singa_device singa_device_opencl(int id) {
  return guarded(
      [&] {
        if (id < 0 || id >= singa::OpenclDevice::GetNumDevices())
          throw std::runtime_error("no such OpenCL device");
        return new singa_device_s{std::make_shared<singa::OpenclDevice>(id)};
      },
      (singa_device) nullptr);
}

int singa_device_lang(singa_device d) {
  return guarded([&] { return static_cast<int>(d->d->lang()); }, -1);
}

int singa_device_description(singa_device d, char *buf, int n) {
  return guarded(
      [&] {
        std::string text;
        if (d->d->lang() == singa::kOpencl)
          text = static_cast<singa::OpenclDevice *>(d->d.get())->Description();
        else if (d->d->lang() == singa::kCpp)
          text = "host";
        else
          text = singa::Platform::DeviceQuery(d->d->id());
        if (buf && n > 0) {
          const int len = std::min<int>(n - 1, static_cast<int>(text.size()));
          std::memcpy(buf, text.data(), len);
          buf[len] = 0;
        }
        return static_cast<int>(text.size());
      },
      -1);
}

void singa_device_sync(singa_device d) {
  guarded([&] { if (d) d->d->Sync(); return 0; }, 0);
}

void singa_device_free(singa_device d) { delete d; }

singa_tensor singa_tensor_new(const int64_t *shape, int ndim, int dtype,
                              singa_device d) {
  return guarded(
      [&] {
        need(shape, "shape");
        auto dev = d ? d->d : singa::defaultDevice;
        auto dt = dtype == SINGA_INT32 ? singa::kInt : singa::kFloat32;
        return wrap(Tensor(to_shape(shape, ndim), dev, dt));
      },
      (singa_tensor) nullptr);
}

void singa_tensor_free(singa_tensor t) { delete t; }

singa_tensor singa_tensor_clone(singa_tensor t) {
  return guarded([&] { need(t, "tensor"); return wrap(t->t.Clone()); },
                 (singa_tensor) nullptr);
}

int singa_tensor_ndim(singa_tensor t) {
  return t ? static_cast<int>(t->t.shape().size()) : -1;
}

int64_t singa_tensor_dim(singa_tensor t, int axis) {
  if (!t || axis < 0 || axis >= static_cast<int>(t->t.shape().size()))
    return -1;
  return static_cast<int64_t>(t->t.shape()[axis]);
}

int64_t singa_tensor_size(singa_tensor t) {
  return t ? static_cast<int64_t>(t->t.Size()) : -1;
}

int singa_tensor_dtype(singa_tensor t) {
  if (!t) return -1;
  return t->t.data_type() == singa::kInt ? SINGA_INT32 : SINGA_FLOAT32;
}

int singa_tensor_set_value(singa_tensor t, float v) {
  return guarded([&] { need(t, "tensor"); t->t.SetValue(v); return 0; }, -1);
}

int singa_tensor_copy_from_float(singa_tensor t, const float *src, int64_t n) {
  return guarded(
      [&] {
        need(t, "tensor");
        need(src, "source");
        if (n < 0 || static_cast<size_t>(n) > t->t.Size())
          throw std::out_of_range("more values than the tensor holds");
        t->t.CopyDataFromHostPtr(src, static_cast<size_t>(n));
        return 0;
      },
      -1);
}

int singa_tensor_copy_from_int(singa_tensor t, const int32_t *src, int64_t n) {
  return guarded(
      [&] {
        need(t, "tensor");
        need(src, "source");
        if (n < 0 || static_cast<size_t>(n) > t->t.Size())
          throw std::out_of_range("more values than the tensor holds");
        t->t.CopyDataFromHostPtr(reinterpret_cast<const int *>(src),
                                 static_cast<size_t>(n));
        return 0;
      },
      -1);
}

int singa_tensor_get_float(singa_tensor t, float *dst, int64_t n) {
  return guarded(
      [&] {
        need(t, "tensor");
        need(dst, "destination");
        if (n < 0 || static_cast<size_t>(n) > t->t.Size())
          throw std::out_of_range("more values than the tensor holds");
        t->t.GetValue(dst, static_cast<size_t>(n));
        return 0;
      },
      -1);
}

int singa_tensor_get_int(singa_tensor t, int32_t *dst, int64_t n) {
  return guarded(
      [&] {
        need(t, "tensor");
        need(dst, "destination");
        if (n < 0 || static_cast<size_t>(n) > t->t.Size())
          throw std::out_of_range("more values than the tensor holds");
        t->t.GetValue(reinterpret_cast<int *>(dst), static_cast<size_t>(n));
        return 0;
      },
      -1);
}

int singa_tensor_to_device(singa_tensor t, singa_device d) {
  return guarded(
      [&] {
        need(t, "tensor");
        need(d, "device");
        t->t.ToDevice(d->d);
        return 0;
      },
      -1);
}

singa_tensor singa_tensor_reshape(singa_tensor t, const int64_t *shape,
                                  int ndim) {
  return guarded(
      [&] {
        need(t, "tensor");
        need(shape, "shape");
        return wrap(singa::Reshape(t->t, to_shape(shape, ndim)));
      },
      (singa_tensor) nullptr);
}

singa_tensor singa_tensor_transpose(singa_tensor t) {
  return guarded([&] { need(t, "tensor"); return wrap(t->t.T()); },
                 (singa_tensor) nullptr);
}

singa_tensor singa_unary(const char *op, singa_tensor x) {
  return guarded(
      [&] {
        need(op, "operator name");
        need(x, "tensor");
        const std::string o(op);
        if (o == "abs") return wrap(singa::Abs(x->t));
        if (o == "exp") return wrap(singa::Exp(x->t));
        if (o == "log") return wrap(singa::Log(x->t));
        if (o == "relu") return wrap(singa::ReLU(x->t));
        if (o == "sigmoid") return wrap(singa::Sigmoid(x->t));
        if (o == "sign") return wrap(singa::Sign(x->t));
        if (o == "sqrt") return wrap(singa::Sqrt(x->t));
        if (o == "square") return wrap(singa::Square(x->t));
        if (o == "tanh") return wrap(singa::Tanh(x->t));
        throw std::invalid_argument("unknown unary operator " + o);
      },
      (singa_tensor) nullptr);
}

singa_tensor singa_binary(const char *op, singa_tensor a, singa_tensor b) {
  return guarded(
      [&] {
        need(op, "operator name");
        need(a, "tensor");
        need(b, "tensor");
        const std::string o(op);
        if (o == "add") return wrap(a->t + b->t);
        if (o == "sub") return wrap(a->t - b->t);
        if (o == "mul") return wrap(a->t * b->t);
        if (o == "div") return wrap(a->t / b->t);
        throw std::invalid_argument("unknown binary operator " + o);
      },
      (singa_tensor) nullptr);
}

singa_tensor singa_matmul(singa_tensor a, singa_tensor b) {
  return guarded(
      [&] {
        need(a, "tensor");
        need(b, "tensor");
        return wrap(singa::Mult(a->t, b->t));
      },
      (singa_tensor) nullptr);
}

singa_tensor singa_sum(singa_tensor x, int axis) {
  return guarded([&] { need(x, "tensor"); return wrap(singa::Sum(x->t, axis)); },
                 (singa_tensor) nullptr);
}

singa_tensor singa_softmax(singa_tensor x) {
  return guarded([&] { need(x, "tensor"); return wrap(singa::SoftMax(x->t)); },
                 (singa_tensor) nullptr);
}

int singa_axpy(float alpha, singa_tensor x, singa_tensor y) {
  return guarded(
      [&] {
        need(x, "tensor");
        need(y, "tensor");
        singa::Axpy(alpha, x->t, &y->t);
        return 0;
      },
      -1);
}

int64_t singa_conf_transcode(const char *message, int in_format,
                             const char *in, int64_t in_len, int out_format,
                             char *out, int64_t cap) {
  return guarded(
      [&]() -> int64_t {
        need(message, "message name");
        if (in_len < 0 || (in_len > 0 && in == nullptr) ||
            (cap > 0 && out == nullptr) || (in_format | out_format) & ~1)
          throw std::invalid_argument("bad argument");
        const std::string name = message, src(in ? in : "", (size_t)in_len);
        std::string res;
        if (name == "LayerConf")
          res = transcode<singa::LayerConf>(src, in_format, out_format);
        else if (name == "ParamSpec")
          res = transcode<singa::ParamSpec>(src, in_format, out_format);
        else if (name == "FillerConf")
          res = transcode<singa::FillerConf>(src, in_format, out_format);
        else if (name == "OptimizerConf")
          res = transcode<singa::OptimizerConf>(src, in_format, out_format);
        else if (name == "RegularizerConf")
          res = transcode<singa::RegularizerConf>(src, in_format, out_format);
        else if (name == "ConstraintConf")
          res = transcode<singa::ConstraintConf>(src, in_format, out_format);
        else if (name == "MetricConf")
          res = transcode<singa::MetricConf>(src, in_format, out_format);
        else if (name == "EncoderConf")
          res = transcode<singa::EncoderConf>(src, in_format, out_format);
        else if (name == "DecoderConf")
          res = transcode<singa::DecoderConf>(src, in_format, out_format);
        else if (name == "TransformerConf")
          res = transcode<singa::TransformerConf>(src, in_format, out_format);
        else if (name == "ImageRecord")
          res = transcode<singa::ImageRecord>(src, in_format, out_format);
        else if (name == "MemPoolConf")
          res = transcode<singa::MemPoolConf>(src, in_format, out_format);
        else
          throw std::invalid_argument("unknown message " + name);
        const int64_t n = (int64_t)res.size();
        if (cap > 0) memcpy(out, res.data(), (size_t)(n < cap ? n : cap));
        return n;
      },
      (int64_t)-1);
}

}  // extern "C"
