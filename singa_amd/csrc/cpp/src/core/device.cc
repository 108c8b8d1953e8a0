#include "singa/core/device.h"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstring>
#include <iostream>
#include <sstream>

#define HIP_CHECK(expr)                                                  \
  do {                                                                   \
    hipError_t e_ = (expr);                                              \
    CHECK(e_ == hipSuccess) << "HIP error: " << hipGetErrorString(e_);   \
  } while (0)

namespace singa {

// ---- Block -------------------------------------------------------------------
Block::~Block() { free_data(); }

void *Block::mutable_data() {
  if (data_ == nullptr && size_ > 0) data_ = dev_->Malloc(size_);
  return data_;
}

const void *Block::data() const {
  CHECK(data_ != nullptr) << "read of a block that was never written";
  return data_;
}

void Block::free_data() {
  if (data_) {
    dev_->Free(data_);
    data_ = nullptr;
  }
}

// ---- Device ------------------------------------------------------------------
Device::Device(int id, int num_executors)
    : id_(id), num_executors_(num_executors),
      graph_(new ::singa_amd::Graph()) {
  ctx_.device_id = id;
  ::singa_amd::ExecHooks hooks;
  hooks.sync = [this]() { this->Sync(); };
  hooks.timed_exec = [this](::singa_amd::Node &n) {
    this->Sync();
    auto t0 = std::chrono::high_resolution_clock::now();
    n.Run();
    this->Sync();
    std::chrono::duration<float> dt =
        std::chrono::high_resolution_clock::now() - t0;
    n.time_elapsed_inc(dt.count());
  };
  hooks.collect = [](::singa_amd::Node &) { return 0.f; };
  graph_->set_hooks(std::move(hooks));
}

Device::~Device() {
  graph_->Reset();
  for (Block *b : held_blocks_)
    if (b->DecRefCount() == 0) delete b;
}

Block *Device::NewBlock(size_t size) {
  if (size == 0) return nullptr;
  return new Block(this, size, next_block_id_++);
}

void Device::FreeBlock(Block *block) { delete block; }

void Device::Exec(std::function<void(Context *)> &&fn,
                  const std::vector<Block *> &read_blocks,
                  const std::vector<Block *> &write_blocks,
                  const std::string &op_name) {
  if (graph_enabled_) {
    ::singa_amd::BlockVec r(read_blocks.begin(), read_blocks.end());
    ::singa_amd::BlockVec w(write_blocks.begin(), write_blocks.end());
    Context *ctx = &ctx_;
    for (Block *b : read_blocks) { b->IncRefCount(); held_blocks_.push_back(b); }
    for (Block *b : write_blocks) { b->IncRefCount(); held_blocks_.push_back(b); }
    graph_->AddOperation([fn, ctx]() { fn(ctx); }, r, w, op_name);
  } else {
    fn(&ctx_);
  }
}

void Device::RunGraph(bool serial) {
  bool prev = graph_enabled_;
  graph_enabled_ = false;  // operations issued while replaying run directly
  if (serial)
    graph_->RunInSerial();
  else
    graph_->RunGraph();
  graph_enabled_ = prev;
}

void Device::ResetGraph() {
  Sync();
  graph_->Reset();
  for (Block *b : held_blocks_)
    if (b->DecRefCount() == 0) delete b;
  held_blocks_.clear();
}

void Device::Reset() {
  graph_enabled_ = false;
  ResetGraph();
  graph_->set_verbosity(0);
}

void Device::PrintTimeProfiling() {
  std::cout << graph_->TimeProfiling() << std::endl;
}

void Device::CopyDataToFrom(Block *dst, Block *src, size_t nBytes,
                            size_t dst_offset, size_t src_offset) {
  Device *sdev = src->device(), *ddev = dst->device();
  // the device that owns GPU memory performs the copy
  Device *runner = ddev->lang() != kCpp ? ddev : sdev;
  bool d_on = ddev->lang() != kCpp, s_on = sdev->lang() != kCpp;
  runner->Exec(
      [=](Context *) {
        runner->CopyRaw(static_cast<char *>(dst->mutable_data()) + dst_offset,
                        static_cast<const char *>(src->data()) + src_offset,
                        nBytes, d_on, s_on);
      },
      {src}, {dst}, "CopyDataToFrom");
}

void Device::CopyDataFromHostPtr(Block *dst, const void *src, size_t nBytes,
                                 size_t dst_offset) {
  // the host buffer may be gone by the time a recorded graph runs: stage it
  std::shared_ptr<std::vector<char>> stage(new std::vector<char>(
      static_cast<const char *>(src), static_cast<const char *>(src) + nBytes));
  bool on_dev = lang_ != kCpp;
  Exec(
      [=](Context *) {
        this->CopyRaw(static_cast<char *>(dst->mutable_data()) + dst_offset,
                      stage->data(), nBytes, on_dev, false);
        if (on_dev) this->Sync();  // staging memory is pageable
      },
      {}, {dst}, "CopyDataFromHostPtr");
}

void Device::CopyDataToHostPtr(void *dst, Block *src, size_t nBytes,
                               size_t src_offset) {
  Sync();
  CopyRaw(dst, static_cast<const char *>(src->data()) + src_offset, nBytes,
          false, lang_ != kCpp);
  Sync();
}

// ---- CppCPU --------------------------------------------------------------------
std::shared_ptr<Device> defaultDevice = std::make_shared<CppCPU>();
std::shared_ptr<Device> CppCPU::defaultDevice = singa::defaultDevice;

CppCPU::CppCPU() : Device(-1, 1) {
  lang_ = kCpp;
  rng_ = new std::mt19937_64(
      std::chrono::system_clock::now().time_since_epoch().count());
}
CppCPU::~CppCPU() { delete static_cast<std::mt19937_64 *>(rng_); }
std::mt19937_64 *CppCPU::rng() { return static_cast<std::mt19937_64 *>(rng_); }
void CppCPU::SetRandSeed(unsigned seed) {
  ctx_.seed = seed;
  rng()->seed(seed);
}
void *CppCPU::Malloc(size_t size) {
  void *p = nullptr;
  pool_.Malloc(&p, size);
  std::memset(p, 0, size);
  return p;
}
void CppCPU::Free(void *ptr) { pool_.Free(ptr); }
void CppCPU::CopyRaw(void *dst, const void *src, size_t n, bool, bool) {
  std::memcpy(dst, src, n);
}

// ---- HipGPU --------------------------------------------------------------------
HipGPU::HipGPU(int id, std::shared_ptr<DeviceMemPool> pool)
    : Device(id, 1), pool_(pool) {
  lang_ = kHip;
  host_ = singa::defaultDevice;
  HIP_CHECK(hipSetDevice(id));
  if (!pool_) pool_ = std::make_shared<HipMemPool>();
  hipStream_t s;
  HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  ctx_.stream = s;
  void *st = nullptr;
  pool_->Malloc(&st, 2 * sizeof(int64_t));
  ctx_.rng_state = static_cast<int64_t *>(st);
  SetRandSeed(static_cast<unsigned>(
      std::chrono::system_clock::now().time_since_epoch().count()));
}

HipGPU::~HipGPU() {
  (void)hipStreamSynchronize(static_cast<hipStream_t>(ctx_.stream));
  pool_->Free(ctx_.rng_state);
  (void)hipStreamDestroy(static_cast<hipStream_t>(ctx_.stream));
}

void HipGPU::SetRandSeed(unsigned seed) {
  ctx_.seed = seed;
  int64_t st[2] = {static_cast<int64_t>(seed), 0};
  HIP_CHECK(hipSetDevice(id_));
  HIP_CHECK(hipMemcpy(ctx_.rng_state, st, sizeof(st), hipMemcpyHostToDevice));
}

void HipGPU::Sync() {
  HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(ctx_.stream)));
}

size_t HipGPU::GetAllocatedMem() {
  auto *p = dynamic_cast<HipMemPool *>(pool_.get());
  return p ? p->bytes_in_use() : 0;
}

void *HipGPU::Malloc(size_t size) {
  HIP_CHECK(hipSetDevice(id_));
  void *p = nullptr;
  pool_->Malloc(&p, size);
  return p;
}

void HipGPU::Free(void *ptr) { pool_->Free(ptr); }

void HipGPU::CopyRaw(void *dst, const void *src, size_t n, bool dst_on_dev,
                     bool src_on_dev) {
  hipMemcpyKind kind = dst_on_dev
                           ? (src_on_dev ? hipMemcpyDeviceToDevice
                                         : hipMemcpyHostToDevice)
                           : (src_on_dev ? hipMemcpyDeviceToHost
                                         : hipMemcpyHostToHost);
  HIP_CHECK(hipMemcpyAsync(dst, src, n, kind,
                           static_cast<hipStream_t>(ctx_.stream)));
}

// ---- Platform ------------------------------------------------------------------
int Platform::GetNumGPUs() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

std::vector<int> Platform::GetGPUIDs() {
  std::vector<int> ids;
  for (int i = 0; i < GetNumGPUs(); ++i) ids.push_back(i);
  return ids;
}

std::pair<size_t, size_t> Platform::GetGPUMemSize(int device) {
  int cur = 0;
  HIP_CHECK(hipGetDevice(&cur));
  HIP_CHECK(hipSetDevice(device));
  size_t f = 0, t = 0;
  HIP_CHECK(hipMemGetInfo(&f, &t));
  HIP_CHECK(hipSetDevice(cur));
  return {f, t};
}

std::vector<std::pair<size_t, size_t>> Platform::GetGPUMemSize() {
  std::vector<std::pair<size_t, size_t>> out;
  for (int i : GetGPUIDs()) out.push_back(GetGPUMemSize(i));
  return out;
}

std::string Platform::DeviceQuery(int id, bool verbose) {
  hipDeviceProp_t p;
  HIP_CHECK(hipGetDeviceProperties(&p, id));
  std::ostringstream os;
  os << "Device id: " << id << "\nName: " << p.name << " (" << p.gcnArchName
     << ")\nTotal global memory: " << p.totalGlobalMem
     << "\nCompute units: " << p.multiProcessorCount
     << "\nWavefront size: " << p.warpSize
     << "\nLDS per workgroup: " << p.sharedMemPerBlock
     << "\nClock rate (kHz): " << p.clockRate << "\n";
  if (verbose)
    os << "Max threads per block: " << p.maxThreadsPerBlock
       << "\nMemory bus width: " << p.memoryBusWidth
       << "\nL2 cache size: " << p.l2CacheSize << "\n";
  return os.str();
}

std::vector<std::shared_ptr<Device>> Platform::CreateCudaGPUs(
    size_t num_devices, size_t init_size) {
  std::vector<int> ids;
  for (size_t i = 0; i < num_devices; ++i) ids.push_back(static_cast<int>(i));
  return CreateCudaGPUsOn(ids, init_size);
}

std::vector<std::shared_ptr<Device>> Platform::CreateCudaGPUsOn(
    const std::vector<int> &devices, size_t init_size) {
  MemPoolConf conf;
  if (init_size) conf.init_size_mb = init_size;
  // one pool shared by the devices of this process
  auto pool = std::make_shared<HipMemPool>(conf);
  std::vector<std::shared_ptr<Device>> out;
  for (int id : devices) {
    CHECK_LT(id, GetNumGPUs()) << "no such GPU";
    out.push_back(std::make_shared<HipGPU>(id, pool));
  }
  return out;
}

}  // namespace singa
