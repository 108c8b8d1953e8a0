/* Flat C interface of the native tensor library, the base of bindings for
 * languages without a C++ interop (Java through java.lang.foreign, see
 * java/; anything with a C FFI).  Handles are opaque pointers; every call
 * returns 0 or a handle on success and a negative code / NULL on failure,
 * with the message available from singa_last_error().  The reference exposes
 * the C++ classes to Java through SWIG-generated JNI (java/CMakeLists.txt,
 * src/api/ *.i); the same calls are offered here as plain C symbols of
 * libsinga_amd_cpp.so. */
#ifndef SINGA_API_C_API_H_
#define SINGA_API_C_API_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct singa_tensor_s *singa_tensor;
typedef struct singa_device_s *singa_device;

enum { SINGA_FLOAT32 = 0, SINGA_INT32 = 2 };

const char *singa_last_error(void);
const char *singa_version(void);

/* devices: the host device always exists; GPUs are numbered from 0 */
int singa_num_gpus(void);
singa_device singa_device_host(void);
singa_device singa_device_gpu(int id);
/* OpenCL devices (the GPUs as seen by the OpenCL runtime; 0 without one) */
int singa_num_opencl_devices(void);
singa_device singa_device_opencl(int id);
/* 0 = host, 1 = HIP GPU, 2 = OpenCL */
int singa_device_lang(singa_device d);
/* writes a NUL-terminated description (at most n bytes), returns its length */
int singa_device_description(singa_device d, char *buf, int n);
void singa_device_sync(singa_device d);
void singa_device_free(singa_device d);

/* tensors */
singa_tensor singa_tensor_new(const int64_t *shape, int ndim, int dtype,
                              singa_device d);
void singa_tensor_free(singa_tensor t);
singa_tensor singa_tensor_clone(singa_tensor t);
int singa_tensor_ndim(singa_tensor t);
int64_t singa_tensor_dim(singa_tensor t, int axis);
int64_t singa_tensor_size(singa_tensor t);
int singa_tensor_dtype(singa_tensor t);
int singa_tensor_set_value(singa_tensor t, float v);
int singa_tensor_copy_from_float(singa_tensor t, const float *src, int64_t n);
int singa_tensor_copy_from_int(singa_tensor t, const int32_t *src, int64_t n);
int singa_tensor_get_float(singa_tensor t, float *dst, int64_t n);
int singa_tensor_get_int(singa_tensor t, int32_t *dst, int64_t n);
int singa_tensor_to_device(singa_tensor t, singa_device d);
singa_tensor singa_tensor_reshape(singa_tensor t, const int64_t *shape,
                                  int ndim);
singa_tensor singa_tensor_transpose(singa_tensor t);

/* element-wise: op is one of abs exp log relu sigmoid sign sqrt square tanh */
singa_tensor singa_unary(const char *op, singa_tensor x);
/* op is one of add sub mul div */
singa_tensor singa_binary(const char *op, singa_tensor a, singa_tensor b);
singa_tensor singa_matmul(singa_tensor a, singa_tensor b);
singa_tensor singa_sum(singa_tensor x, int axis);
singa_tensor singa_softmax(singa_tensor x);
int singa_axpy(float alpha, singa_tensor x, singa_tensor y);

/* Configuration messages of the model API (the messages of the reference's
   src/proto/model.proto): read `in` as protobuf wire format (format 0) or
   text format (format 1) and write it back in the other or the same format.
   message is one of LayerConf, ParamSpec, FillerConf, OptimizerConf,
   RegularizerConf, ConstraintConf, MetricConf.  Returns the length of the
   result, of which at most cap bytes are copied to out, or -1 (see
   singa_last_error). */
int64_t singa_conf_transcode(const char *message, int in_format,
                             const char *in, int64_t in_len, int out_format,
                             char *out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif  /* SINGA_API_C_API_H_ */
