// Devices of the C++ API: the host (CppCPU) and one MI355X (HipGPU).
//
// Every tensor operation is submitted through Device::Exec as a closure plus
// the blocks it reads and writes.  With graph buffering off the closure runs
// at once; with it on the closure is recorded in the device's Graph and
// executed by RunGraph(), which also releases intermediate blocks as soon as
// their last reader has run.  HipGPU launches on one HIP stream per device and
// never synchronises inside an operation.
//
// API parity: reference include/singa/core/device.h:57-390 (Device, CppCPU,
// CudaGPU, Platform).
#pragma once
#include <functional>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../../../../runtime/graph.h"
#include "common.h"
#include "memory.h"

namespace singa {

class Device {
 public:
  virtual ~Device();
  Device(int id, int num_executors);

  /// New lazily allocated block of `size` bytes (nullptr for size 0).
  Block *NewBlock(size_t size);
  void FreeBlock(Block *block);

  virtual void SetRandSeed(unsigned seed) = 0;
  virtual std::shared_ptr<Device> host() const { return host_; }
  LangType lang() const { return lang_; }
  int id() const { return id_; }
  Context *context(int /*k*/ = 0) { return &ctx_; }

  /// Submit an operation (see file comment).
  void Exec(std::function<void(Context *)> &&fn,
            const std::vector<Block *> &read_blocks,
            const std::vector<Block *> &write_blocks,
            const std::string &op_name = "no_name");

  /// Copy `nBytes` between blocks of any two devices.
  void CopyDataToFrom(Block *dst, Block *src, size_t nBytes,
                      size_t dst_offset = 0, size_t src_offset = 0);
  void CopyDataFromHostPtr(Block *dst, const void *src, size_t nBytes,
                           size_t dst_offset = 0);
  /// Blocking copy from a block to host memory.
  void CopyDataToHostPtr(void *dst, Block *src, size_t nBytes,
                         size_t src_offset = 0);

  // ---- graph buffering -------------------------------------------------------
  void EnableGraph(bool enable) { graph_enabled_ = enable; }
  bool graph_enabled() const { return graph_enabled_; }
  void RunGraph(bool serial = false);
  void ResetGraph();
  void Reset();
  ::singa_amd::Graph *graph() const { return graph_.get(); }
  void PrintTimeProfiling();
  void SetVerbosity(int v) { graph_->set_verbosity(v); }
  void SetSkipIteration(int n) { graph_->set_skip_iteration(n); }
  int verbosity() const { return graph_->verbosity(); }

  /// Block the caller until all submitted work has finished.
  virtual void Sync() {}
  virtual size_t GetAllocatedMem() { return 0; }

  // used by Block
  virtual void *Malloc(size_t size) = 0;
  virtual void Free(void *ptr) = 0;
  /// dst/src are raw pointers on (dst_dev, src_dev); nullptr device = host.
  virtual void CopyRaw(void *dst, const void *src, size_t n, bool dst_on_dev,
                       bool src_on_dev) = 0;

 protected:
  int id_ = 0;
  int num_executors_ = 0;
  LangType lang_ = kCpp;
  bool graph_enabled_ = false;
  std::shared_ptr<Device> host_;
  Context ctx_;
  std::unique_ptr<::singa_amd::Graph> graph_;
  int64_t next_block_id_ = 0;
  // one reference per (operation, block) pair recorded in the graph, so the
  // blocks outlive the tensors that were alive while recording
  std::vector<Block *> held_blocks_;
};

class CppCPU : public Device {
 public:
  CppCPU();
  ~CppCPU() override;
  std::shared_ptr<Device> host() const override { return defaultDevice; }
  void SetRandSeed(unsigned seed) override;
  void *Malloc(size_t size) override;
  void Free(void *ptr) override;
  void CopyRaw(void *dst, const void *src, size_t n, bool, bool) override;
  std::mt19937_64 *rng();
  static std::shared_ptr<Device> defaultDevice;

 private:
  HostMemPool pool_;
  void *rng_ = nullptr;
};

extern std::shared_ptr<Device> defaultDevice;

class HipGPU : public Device {
 public:
  explicit HipGPU(int id = 0, std::shared_ptr<DeviceMemPool> pool = nullptr);
  ~HipGPU() override;
  void SetRandSeed(unsigned seed) override;
  void Sync() override;
  size_t GetAllocatedMem() override;
  void *Malloc(size_t size) override;
  void Free(void *ptr) override;
  void CopyRaw(void *dst, const void *src, size_t n, bool dst_on_dev,
               bool src_on_dev) override;

 private:
  std::shared_ptr<DeviceMemPool> pool_;
};
using CudaGPU = HipGPU;  // name used by programs written for the reference

class Platform {
 public:
  static int GetNumGPUs();
  static std::vector<int> GetGPUIDs();
  static std::pair<size_t, size_t> GetGPUMemSize(int device);
  static std::vector<std::pair<size_t, size_t>> GetGPUMemSize();
  static std::string DeviceQuery(int id, bool verbose = false);
  static std::vector<std::shared_ptr<Device>> CreateCudaGPUs(
      size_t num_devices, size_t init_size = 0);
  static std::vector<std::shared_ptr<Device>> CreateCudaGPUsOn(
      const std::vector<int> &devices, size_t init_size = 0);
  static std::shared_ptr<Device> GetDefaultDevice() { return defaultDevice; }
};

}  // namespace singa
