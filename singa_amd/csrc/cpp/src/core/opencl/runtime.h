// OpenCL plumbing of OpenclDevice: the run-time loaded API table, one
// context / in-order queue / program per device, and the launchers of the
// tensor-math kernels, whose signatures mirror the HIP launchers
// (csrc/hip/kernels.h) so that tensor.cc dispatches to either with the same
// arguments.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>

#define CL_TARGET_OPENCL_VERSION 200
#include <CL/cl.h>

#include "../../../../hip/kernels.h"

namespace singa {
namespace ocl {

/// Entry points of libOpenCL resolved with dlsym; null table = no runtime.
struct Api {
  decltype(&clGetPlatformIDs) GetPlatformIDs;
  decltype(&clGetPlatformInfo) GetPlatformInfo;
  decltype(&clGetDeviceIDs) GetDeviceIDs;
  decltype(&clGetDeviceInfo) GetDeviceInfo;
  decltype(&clCreateContext) CreateContext;
  decltype(&clCreateCommandQueueWithProperties) CreateCommandQueueWithProperties;
  decltype(&clCreateProgramWithSource) CreateProgramWithSource;
  decltype(&clBuildProgram) BuildProgram;
  decltype(&clGetProgramBuildInfo) GetProgramBuildInfo;
  decltype(&clCreateKernel) CreateKernel;
  decltype(&clSetKernelArg) SetKernelArg;
  decltype(&clSetKernelArgSVMPointer) SetKernelArgSVMPointer;
  decltype(&clEnqueueNDRangeKernel) EnqueueNDRangeKernel;
  decltype(&clFinish) Finish;
  decltype(&clSVMAlloc) SVMAlloc;
  decltype(&clSVMFree) SVMFree;
  decltype(&clEnqueueSVMMemcpy) EnqueueSVMMemcpy;
  decltype(&clReleaseKernel) ReleaseKernel;
  decltype(&clReleaseProgram) ReleaseProgram;
  decltype(&clReleaseCommandQueue) ReleaseCommandQueue;
  decltype(&clReleaseContext) ReleaseContext;
};
/// The table, or nullptr (with the reason in *why) when libOpenCL or one of
/// its entry points is missing.
const Api *LoadApi(std::string *why = nullptr);
/// GPU devices of the first OpenCL platform that has any (0 without a runtime).
int CountDevices();

class Runtime {
 public:
  /// Opens GPU device `index` of the first platform that has one.
  explicit Runtime(int index);
  ~Runtime();
  Runtime(const Runtime &) = delete;
  Runtime &operator=(const Runtime &) = delete;

  void *Alloc(size_t bytes);
  void Free(void *p);
  /// Blocking copy between any two of host memory and SVM blocks.
  void Copy(void *dst, const void *src, size_t bytes);
  void Finish();
  const std::string &description() const { return description_; }

  /// Kernel `name` of the embedded program (compiled on first request).
  cl_kernel Kernel(const char *name);
  /// Enqueue with a 1-d or 2-d range (gy == 0: 1-d); sizes in work-items.
  void Launch(cl_kernel k, size_t gx, size_t lx, size_t gy = 0, size_t ly = 0);

  // argument setters: plain value or SVM pointer
  template <typename T>
  void Arg(cl_kernel k, int i, const T &v) { ArgBytes(k, i, sizeof(T), &v); }
  void ArgBytes(cl_kernel k, int i, size_t n, const void *p);
  void ArgPtr(cl_kernel k, int i, const void *p);

  // state of the counter-based generator
  uint64_t seed = 0x5eed;
  uint64_t offset = 0;
  int compute_units = 0;

 private:
  void BuildProgram();
  const Api *api_;
  cl_device_id device_ = nullptr;
  cl_context context_ = nullptr;
  cl_command_queue queue_ = nullptr;
  cl_program program_ = nullptr;
  std::map<std::string, cl_kernel> kernels_;
  std::string description_;
};

static_assert(sizeof(sa::Desc) == 8 + 4 * 8 * 8,
              "the kernels' Desc mirrors sa::Desc field by field");

// ---- launchers (0 = ok, negative = rejected arguments) ------------------------------
int unary(Runtime *rt, int op, int dt_in, int dt_out, const void *x, void *y,
          const sa::Desc *d, int64_t n, float a, float b);
int binary(Runtime *rt, int op, int dtype, const void *x, const void *y,
           void *z, const sa::Desc *d, int64_t n);
int fill(Runtime *rt, int dtype, void *y, int64_t n, float v);
int axpy(Runtime *rt, int dtype, float alpha, const void *x, void *y,
         int64_t n);
int reduce_rows(Runtime *rt, int op, int dtype, const void *x, void *out,
                int64_t rows, int64_t cols);
int reduce_cols(Runtime *rt, int dtype, const void *x, void *out, int64_t rows,
                int64_t cols);
int reduce_all(Runtime *rt, int op, int dtype, const void *x, void *out,
               int64_t n);
int softmax_fwd(Runtime *rt, int dtype, const void *x, void *y, int64_t rows,
                int64_t cols);
int softmax_bwd(Runtime *rt, int dtype, const void *dy, const void *y,
                void *dx, int64_t rows, int64_t cols);
int cross_entropy_fwd(Runtime *rt, const void *p, const void *t, void *loss,
                      int64_t rows, int64_t cols);
int softmax_ce_bwd(Runtime *rt, void *p, const void *t, int64_t rows,
                   int64_t cols);
int gemm(Runtime *rt, int dtype, int64_t M, int64_t N, int64_t K, float alpha,
         const void *A, int64_t a_rs, int64_t a_cs, const void *B,
         int64_t b_rs, int64_t b_cs, float beta, void *C, int64_t ldc);
int rng_uniform(Runtime *rt, int dtype, void *y, int64_t n, float lo,
                float hi);
int rng_normal(Runtime *rt, int dtype, void *y, int64_t n, float mean,
               float std);
int rng_bernoulli(Runtime *rt, int dtype, void *y, int64_t n, float p);


// convolution / pooling (nn.cl): float32, NCHW
int conv_fwd(Runtime *rt, const ConvGeom *g, const void *x, const void *w,
             const void *bias, void *y);
int conv_bwd_data(Runtime *rt, const ConvGeom *g, const void *dy,
                  const void *w, void *dx);
int conv_bwd_weight(Runtime *rt, const ConvGeom *g, const void *dy,
                    const void *x, void *dw);
int conv_bwd_bias(Runtime *rt, const void *dy, void *db, int N, int K, int PQ);
int pool_fwd(Runtime *rt, const PoolGeom *g, const void *x, void *y);
int pool_bwd(Runtime *rt, const PoolGeom *g, const void *dy, const void *x,
             const void *y, void *dx);
// batch normalisation over (N, C, HW), float32; statistics as the HIP side
int bn_fwd_train(Runtime *rt, const void *x, const float *scale,
                 const float *bias, float *rmean, float *rvar, float factor,
                 float eps, void *y, float *mean, float *invstd, int N, int C,
                 int HW);
int bn_fwd_infer(Runtime *rt, const void *x, const float *scale,
                 const float *bias, const float *rmean, const float *rvar,
                 float eps, void *y, int N, int C, int HW);
int bn_bwd(Runtime *rt, const void *dy, const void *x, const float *scale,
           const float *mean, const float *invstd, void *dx, float *dscale,
           float *dbias, int N, int C, int HW);

}  // namespace ocl
}  // namespace singa
