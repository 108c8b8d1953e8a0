#include "runtime.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <sstream>
#include <vector>

#include "singa/utils/logging.h"

namespace singa {
namespace ocl {

namespace {

// the kernel sources, embedded by the build (singa_amd/build.py)
const char kProgramSource[] =
#include "cl_sources.inc"
    ;

#define CL_OK(expr)                                                        \
  do {                                                                     \
    const cl_int cl_rc_ = (expr);                                          \
    CHECK_EQ(cl_rc_, CL_SUCCESS) << "OpenCL call failed: " #expr;          \
  } while (0)

template <typename F>
bool Resolve(void *lib, const char *name, F *out, std::string *why) {
  *out = reinterpret_cast<F>(dlsym(lib, name));
  if (*out == nullptr && why) *why = std::string("libOpenCL lacks ") + name;
  return *out != nullptr;
}

std::string DefineList() {
  // operator codes of the C++ enums, as compiler options of the .cl program
  std::ostringstream o;
  o << "-cl-std=CL2.0";
#define D(name) o << " -D" #name "=" << static_cast<int>(sa::name)
  D(U_COPY); D(U_ABS); D(U_EXP); D(U_LOG); D(U_SQRT); D(U_SQUARE); D(U_SIGN);
  D(U_CEIL); D(U_FLOOR); D(U_ROUND); D(U_ROUNDE); D(U_SIGMOID); D(U_TANH);
  D(U_RELU); D(U_ERF); D(U_COS); D(U_COSH); D(U_ACOS); D(U_ACOSH); D(U_SIN);
  D(U_SINH); D(U_ASIN); D(U_ASINH); D(U_TAN); D(U_ATAN); D(U_ATANH);
  D(U_SOFTPLUS); D(U_SOFTSIGN); D(U_CLAMP); D(U_ADD_S); D(U_SUB_S); D(U_MUL_S);
  D(U_DIV_S); D(U_POW_S); D(U_RDIV_S); D(U_LT_S); D(U_LE_S); D(U_GT_S);
  D(U_GE_S); D(U_EQ_S); D(U_MAX_S); D(U_MIN_S);
  D(B_ADD); D(B_SUB); D(B_MUL); D(B_DIV); D(B_POW); D(B_LT); D(B_LE); D(B_GT);
  D(B_GE); D(B_EQ); D(B_MAX); D(B_MIN); D(B_RELU_BWD);
  D(R_SUM); D(R_MEAN); D(R_MAX); D(R_MIN); D(R_ASUM); D(R_SUMSQ);
#undef D
  return o.str();
}

}  // namespace

const Api *LoadApi(std::string *why) {
  static std::once_flag once;
  static Api api;
  static bool ok = false;
  static std::string reason;
  std::call_once(once, [] {
    void *lib = dlopen("libOpenCL.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("libOpenCL.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
      reason = "libOpenCL.so.1 not found";
      return;
    }
#define R(field, sym) Resolve(lib, #sym, &api.field, &reason)
    ok = R(GetPlatformIDs, clGetPlatformIDs) &&
         R(GetPlatformInfo, clGetPlatformInfo) &&
         R(GetDeviceIDs, clGetDeviceIDs) &&
         R(GetDeviceInfo, clGetDeviceInfo) &&
         R(CreateContext, clCreateContext) &&
         R(CreateCommandQueueWithProperties,
           clCreateCommandQueueWithProperties) &&
         R(CreateProgramWithSource, clCreateProgramWithSource) &&
         R(BuildProgram, clBuildProgram) &&
         R(GetProgramBuildInfo, clGetProgramBuildInfo) &&
         R(CreateKernel, clCreateKernel) && R(SetKernelArg, clSetKernelArg) &&
         R(SetKernelArgSVMPointer, clSetKernelArgSVMPointer) &&
         R(EnqueueNDRangeKernel, clEnqueueNDRangeKernel) &&
         R(Finish, clFinish) && R(SVMAlloc, clSVMAlloc) &&
         R(SVMFree, clSVMFree) && R(EnqueueSVMMemcpy, clEnqueueSVMMemcpy) &&
         R(ReleaseKernel, clReleaseKernel) &&
         R(ReleaseProgram, clReleaseProgram) &&
         R(ReleaseCommandQueue, clReleaseCommandQueue) &&
         R(ReleaseContext, clReleaseContext);
#undef R
  });
  if (!ok && why) *why = reason;
  return ok ? &api : nullptr;
}

static std::vector<cl_device_id> GpuDevices(const Api *api,
                                            std::string *platform_name) {
  std::vector<cl_device_id> found;
  cl_uint np = 0;
  if (api->GetPlatformIDs(0, nullptr, &np) != CL_SUCCESS || np == 0)
    return found;
  std::vector<cl_platform_id> platforms(np);
  api->GetPlatformIDs(np, platforms.data(), nullptr);
  for (cl_platform_id p : platforms) {
    cl_uint nd = 0;
    if (api->GetDeviceIDs(p, CL_DEVICE_TYPE_GPU, 0, nullptr, &nd) !=
            CL_SUCCESS || nd == 0)
      continue;
    found.resize(nd);
    api->GetDeviceIDs(p, CL_DEVICE_TYPE_GPU, nd, found.data(), nullptr);
    if (platform_name) {
      char name[256] = {0};
      api->GetPlatformInfo(p, CL_PLATFORM_NAME, sizeof(name) - 1, name,
                           nullptr);
      *platform_name = name;
    }
    break;
  }
  return found;
}

int CountDevices() {
  const Api *api = LoadApi();
  return api ? static_cast<int>(GpuDevices(api, nullptr).size()) : 0;
}

Runtime::Runtime(int index) {
  std::string why;
  api_ = LoadApi(&why);
  CHECK(api_ != nullptr) << "no OpenCL runtime: " << why;
  std::string platform;
  std::vector<cl_device_id> devices = GpuDevices(api_, &platform);
  CHECK_LT(static_cast<size_t>(index), devices.size())
      << "OpenCL GPU " << index << " requested, " << devices.size()
      << " present";
  device_ = devices[index];
  char name[256] = {0}, version[256] = {0};
  cl_uint cus = 0;
  api_->GetDeviceInfo(device_, CL_DEVICE_NAME, sizeof(name) - 1, name, nullptr);
  api_->GetDeviceInfo(device_, CL_DEVICE_OPENCL_C_VERSION, sizeof(version) - 1,
                      version, nullptr);
  api_->GetDeviceInfo(device_, CL_DEVICE_MAX_COMPUTE_UNITS, sizeof(cus), &cus,
                      nullptr);
  cl_device_svm_capabilities svm = 0;
  api_->GetDeviceInfo(device_, CL_DEVICE_SVM_CAPABILITIES, sizeof(svm), &svm,
                      nullptr);
  CHECK(svm & CL_DEVICE_SVM_COARSE_GRAIN_BUFFER)
      << name << " has no coarse-grained shared virtual memory";
  compute_units = static_cast<int>(cus);
  std::ostringstream d;
  d << platform << " / " << name << ", " << cus << " compute units, "
    << version;
  description_ = d.str();
  cl_int rc = CL_SUCCESS;
  context_ = api_->CreateContext(nullptr, 1, &device_, nullptr, nullptr, &rc);
  CHECK_EQ(rc, CL_SUCCESS) << "clCreateContext";
  queue_ = api_->CreateCommandQueueWithProperties(context_, device_, nullptr,
                                                  &rc);
  CHECK_EQ(rc, CL_SUCCESS) << "clCreateCommandQueueWithProperties";
}

Runtime::~Runtime() {
  if (queue_) api_->Finish(queue_);
  for (auto &kv : kernels_) api_->ReleaseKernel(kv.second);
  if (program_) api_->ReleaseProgram(program_);
  if (queue_) api_->ReleaseCommandQueue(queue_);
  if (context_) api_->ReleaseContext(context_);
}

void *Runtime::Alloc(size_t bytes) {
  void *p = api_->SVMAlloc(context_, CL_MEM_READ_WRITE, bytes, 0);
  CHECK(p != nullptr) << "clSVMAlloc of " << bytes << " bytes failed";
  return p;
}

void Runtime::Free(void *p) {
  if (!p) return;
  api_->Finish(queue_);   // kernels in flight may still use the block
  api_->SVMFree(context_, p);
}

void Runtime::Copy(void *dst, const void *src, size_t bytes) {
  if (bytes == 0) return;
  CL_OK(api_->EnqueueSVMMemcpy(queue_, CL_TRUE, dst, src, bytes, 0, nullptr,
                               nullptr));
}

void Runtime::Finish() { CL_OK(api_->Finish(queue_)); }

void Runtime::BuildProgram() {
  const char *src = kProgramSource;
  const size_t len = sizeof(kProgramSource) - 1;
  cl_int rc = CL_SUCCESS;
  program_ = api_->CreateProgramWithSource(context_, 1, &src, &len, &rc);
  CHECK_EQ(rc, CL_SUCCESS) << "clCreateProgramWithSource";
  const std::string options = DefineList();
  rc = api_->BuildProgram(program_, 1, &device_, options.c_str(), nullptr,
                          nullptr);
  if (rc != CL_SUCCESS) {
    size_t n = 0;
    api_->GetProgramBuildInfo(program_, device_, CL_PROGRAM_BUILD_LOG, 0,
                              nullptr, &n);
    std::string log(n, '\0');
    api_->GetProgramBuildInfo(program_, device_, CL_PROGRAM_BUILD_LOG, n,
                              &log[0], nullptr);
    LOG(FATAL) << "OpenCL program build failed (" << rc << "):\n"
               << log.c_str();   // up to the terminator the runtime wrote
  }
}

cl_kernel Runtime::Kernel(const char *name) {
  auto it = kernels_.find(name);
  if (it != kernels_.end()) return it->second;
  if (!program_) BuildProgram();
  cl_int rc = CL_SUCCESS;
  cl_kernel k = api_->CreateKernel(program_, name, &rc);
  CHECK_EQ(rc, CL_SUCCESS) << "no kernel " << name << " in the program";
  kernels_[name] = k;
  return k;
}

void Runtime::ArgBytes(cl_kernel k, int i, size_t n, const void *p) {
  CL_OK(api_->SetKernelArg(k, static_cast<cl_uint>(i), n, p));
}

void Runtime::ArgPtr(cl_kernel k, int i, const void *p) {
  CL_OK(api_->SetKernelArgSVMPointer(k, static_cast<cl_uint>(i), p));
}

void Runtime::Launch(cl_kernel k, size_t gx, size_t lx, size_t gy, size_t ly) {
  const size_t global[2] = {gx, gy}, local[2] = {lx, ly};
  CL_OK(api_->EnqueueNDRangeKernel(queue_, k, gy ? 2 : 1, nullptr, global,
                                   local, 0, nullptr, nullptr));
}

// ---- launchers ----------------------------------------------------------------------
namespace {

constexpr int kF32 = 0, kI32 = 2;   // DataType codes shared with the HIP side
constexpr size_t kGroup = 256;      // four 64-wide wavefronts

bool known(int dt) { return dt == kF32 || dt == kI32; }

// work-items for an n-element grid-stride kernel: whole groups, at most
// eight groups per compute unit
size_t Items(Runtime *rt, int64_t n) {
  size_t groups = static_cast<size_t>((n + kGroup - 1) / kGroup);
  const size_t cap = static_cast<size_t>(std::max(1, rt->compute_units)) * 8;
  groups = std::max<size_t>(1, std::min(groups, cap));
  return groups * kGroup;
}

size_t RowGroups(Runtime *rt, int64_t rows) {
  const size_t cap = static_cast<size_t>(std::max(1, rt->compute_units)) * 8;
  return std::max<size_t>(1, std::min<size_t>(rows, cap));
}

}  // namespace

int unary(Runtime *rt, int op, int dt_in, int dt_out, const void *x, void *y,
          const sa::Desc *d, int64_t n, float a, float b) {
  if (!known(dt_in) || !known(dt_out)) return -3;
  if (n <= 0) return 0;
  const int in_int = dt_in == kI32, out_int = dt_out == kI32;
  cl_kernel k = rt->Kernel(d ? "k_unary_desc" : "k_unary");
  int i = 0;
  rt->Arg(k, i++, op);
  rt->ArgPtr(k, i++, x);
  rt->Arg(k, i++, in_int);
  rt->ArgPtr(k, i++, y);
  rt->Arg(k, i++, out_int);
  if (d) rt->Arg(k, i++, *d);
  rt->Arg(k, i++, static_cast<cl_long>(n));
  rt->Arg(k, i++, a);
  rt->Arg(k, i++, b);
  rt->Launch(k, Items(rt, n), kGroup);
  return 0;
}

int binary(Runtime *rt, int op, int dtype, const void *x, const void *y,
           void *z, const sa::Desc *d, int64_t n) {
  if (!known(dtype)) return -3;
  if (n <= 0) return 0;
  const int is_int = dtype == kI32;
  cl_kernel k = rt->Kernel(d ? "k_binary_desc" : "k_binary");
  int i = 0;
  rt->Arg(k, i++, op);
  rt->ArgPtr(k, i++, x);
  rt->ArgPtr(k, i++, y);
  rt->ArgPtr(k, i++, z);
  rt->Arg(k, i++, is_int);
  if (d) rt->Arg(k, i++, *d);
  rt->Arg(k, i++, static_cast<cl_long>(n));
  rt->Launch(k, Items(rt, n), kGroup);
  return 0;
}

int fill(Runtime *rt, int dtype, void *y, int64_t n, float v) {
  if (!known(dtype)) return -3;
  if (n <= 0) return 0;
  cl_kernel k = rt->Kernel("k_fill");
  rt->ArgPtr(k, 0, y);
  rt->Arg(k, 1, static_cast<int>(dtype == kI32));
  rt->Arg(k, 2, static_cast<cl_long>(n));
  rt->Arg(k, 3, v);
  rt->Launch(k, Items(rt, n), kGroup);
  return 0;
}

int axpy(Runtime *rt, int dtype, float alpha, const void *x, void *y,
         int64_t n) {
  if (dtype != kF32) return -3;
  if (n <= 0) return 0;
  cl_kernel k = rt->Kernel("k_axpy");
  rt->Arg(k, 0, alpha);
  rt->ArgPtr(k, 1, x);
  rt->ArgPtr(k, 2, y);
  rt->Arg(k, 3, static_cast<cl_long>(n));
  rt->Launch(k, Items(rt, n), kGroup);
  return 0;
}

int reduce_rows(Runtime *rt, int op, int dtype, const void *x, void *out,
                int64_t rows, int64_t cols) {
  if (dtype != kF32) return -3;
  if (rows <= 0) return 0;
  cl_kernel k = rt->Kernel("k_reduce_rows");
  rt->Arg(k, 0, op);
  rt->ArgPtr(k, 1, x);
  rt->ArgPtr(k, 2, out);
  rt->Arg(k, 3, static_cast<cl_long>(rows));
  rt->Arg(k, 4, static_cast<cl_long>(cols));
  rt->Launch(k, RowGroups(rt, rows) * kGroup, kGroup);
  return 0;
}

int reduce_cols(Runtime *rt, int dtype, const void *x, void *out, int64_t rows,
                int64_t cols) {
  if (dtype != kF32) return -3;
  if (cols <= 0) return 0;
  cl_kernel k = rt->Kernel("k_reduce_cols");
  rt->ArgPtr(k, 0, x);
  rt->ArgPtr(k, 1, out);
  rt->Arg(k, 2, static_cast<cl_long>(rows));
  rt->Arg(k, 3, static_cast<cl_long>(cols));
  rt->Launch(k, Items(rt, cols), kGroup);
  return 0;
}

int reduce_all(Runtime *rt, int op, int dtype, const void *x, void *out,
               int64_t n) {
  if (dtype != kF32) return -3;
  const size_t items = Items(rt, std::max<int64_t>(n, 1));
  const int groups = static_cast<int>(items / kGroup);
  float *partial = static_cast<float *>(rt->Alloc(groups * sizeof(float)));
  cl_kernel k = rt->Kernel("k_reduce_all");
  rt->Arg(k, 0, op);
  rt->ArgPtr(k, 1, x);
  rt->ArgPtr(k, 2, partial);
  rt->Arg(k, 3, static_cast<cl_long>(n));
  rt->Launch(k, items, kGroup);
  cl_kernel f = rt->Kernel("k_reduce_all_final");
  rt->Arg(f, 0, op);
  rt->ArgPtr(f, 1, partial);
  rt->Arg(f, 2, groups);
  rt->ArgPtr(f, 3, out);
  rt->Launch(f, kGroup, kGroup);
  rt->Free(partial);   // waits for the queue
  return 0;
}

static int rows_kernel(Runtime *rt, const char *name, const void *a,
                       const void *b, void *c, int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  cl_kernel k = rt->Kernel(name);
  int i = 0;
  rt->ArgPtr(k, i++, a);
  if (b) rt->ArgPtr(k, i++, b);
  rt->ArgPtr(k, i++, c);
  rt->Arg(k, i++, static_cast<cl_long>(rows));
  rt->Arg(k, i++, static_cast<cl_long>(cols));
  rt->Launch(k, RowGroups(rt, rows) * kGroup, kGroup);
  return 0;
}

int softmax_fwd(Runtime *rt, int dtype, const void *x, void *y, int64_t rows,
                int64_t cols) {
  if (dtype != kF32) return -3;
  return rows_kernel(rt, "k_softmax_fwd", x, nullptr, y, rows, cols);
}

int softmax_bwd(Runtime *rt, int dtype, const void *dy, const void *y,
                void *dx, int64_t rows, int64_t cols) {
  if (dtype != kF32) return -3;
  return rows_kernel(rt, "k_softmax_bwd", dy, y, dx, rows, cols);
}

int cross_entropy_fwd(Runtime *rt, const void *p, const void *t, void *loss,
                      int64_t rows, int64_t cols) {
  if (rows <= 0) return 0;
  cl_kernel k = rt->Kernel("k_cross_entropy_fwd");
  rt->ArgPtr(k, 0, p);
  rt->ArgPtr(k, 1, t);
  rt->ArgPtr(k, 2, loss);
  rt->Arg(k, 3, static_cast<cl_long>(rows));
  rt->Arg(k, 4, static_cast<cl_long>(cols));
  rt->Launch(k, Items(rt, rows), kGroup);
  return 0;
}

int softmax_ce_bwd(Runtime *rt, void *p, const void *t, int64_t rows,
                   int64_t cols) {
  if (rows <= 0) return 0;
  cl_kernel k = rt->Kernel("k_softmax_ce_bwd");
  rt->ArgPtr(k, 0, p);
  rt->ArgPtr(k, 1, t);
  rt->Arg(k, 2, static_cast<cl_long>(rows));
  rt->Arg(k, 3, static_cast<cl_long>(cols));
  rt->Launch(k, Items(rt, rows), kGroup);
  return 0;
}

int gemm(Runtime *rt, int dtype, int64_t M, int64_t N, int64_t K, float alpha,
         const void *A, int64_t a_rs, int64_t a_cs, const void *B,
         int64_t b_rs, int64_t b_cs, float beta, void *C, int64_t ldc) {
  if (dtype != kF32) return -3;
  if (M <= 0 || N <= 0) return 0;
  cl_kernel k = rt->Kernel("k_gemm");
  int i = 0;
  rt->Arg(k, i++, static_cast<cl_long>(M));
  rt->Arg(k, i++, static_cast<cl_long>(N));
  rt->Arg(k, i++, static_cast<cl_long>(K));
  rt->Arg(k, i++, alpha);
  rt->ArgPtr(k, i++, A);
  rt->Arg(k, i++, static_cast<cl_long>(a_rs));
  rt->Arg(k, i++, static_cast<cl_long>(a_cs));
  rt->ArgPtr(k, i++, B);
  rt->Arg(k, i++, static_cast<cl_long>(b_rs));
  rt->Arg(k, i++, static_cast<cl_long>(b_cs));
  rt->Arg(k, i++, beta);
  rt->ArgPtr(k, i++, C);
  rt->Arg(k, i++, static_cast<cl_long>(ldc));
  const size_t gx = static_cast<size_t>((N + 63) / 64) * 16;
  const size_t gy = static_cast<size_t>((M + 63) / 64) * 16;
  rt->Launch(k, gx, 16, gy, 16);
  return 0;
}

static int rng(Runtime *rt, const char *name, int dtype, void *y, int64_t n,
               float a, float b, bool two_params, uint64_t draws_per_item) {
  if (dtype != kF32) return -3;
  if (n <= 0) return 0;
  cl_kernel k = rt->Kernel(name);
  int i = 0;
  rt->ArgPtr(k, i++, y);
  rt->Arg(k, i++, static_cast<cl_long>(n));
  rt->Arg(k, i++, a);
  if (two_params) rt->Arg(k, i++, b);
  rt->Arg(k, i++, static_cast<cl_ulong>(rt->seed));
  rt->Arg(k, i++, static_cast<cl_ulong>(rt->offset));
  rt->Launch(k, Items(rt, n), kGroup);
  rt->offset += static_cast<uint64_t>(n) * draws_per_item;
  return 0;
}

int rng_uniform(Runtime *rt, int dtype, void *y, int64_t n, float lo,
                float hi) {
  return rng(rt, "k_rng_uniform", dtype, y, n, lo, hi, true, 1);
}

int rng_normal(Runtime *rt, int dtype, void *y, int64_t n, float mean,
               float std) {
  return rng(rt, "k_rng_normal", dtype, y, n, mean, std, true, 2);
}

int rng_bernoulli(Runtime *rt, int dtype, void *y, int64_t n, float p) {
  return rng(rt, "k_rng_bernoulli", dtype, y, n, p, 0.f, false, 1);
}

}  // namespace ocl
}  // namespace singa
