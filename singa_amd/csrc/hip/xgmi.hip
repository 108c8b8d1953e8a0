// Collectives that move data directly between the GPUs of one node.
//
// The eight MI355X of a node are connected pairwise by xGMI links: a GPU has
// seven links, one to each peer, and no switch in between.  A ring collective
// keeps one link per GPU busy at a time; a kernel that addresses every peer's
// memory at once keeps all seven busy.  Every rank therefore owns a *window*
// (device memory allocated without caching and exported to the other ranks'
// processes through IPC handles) and a *signal area* of flag words.  The
// all-reduce of a bucket that already lies in the window is one kernel per
// rank:
//
//   two-shot (large buckets)
//     barrier  -- every peer's bucket is complete in its window
//     rank r reads slice r from all W windows (W-1 remote reads in flight per
//       thread, one per link), adds them in rank order in float32 and writes
//       the sum back to slice r of its own window
//     barrier  -- every slice is reduced
//     rank r copies slice p from peer p's window into its own, for all p != r
//     barrier  -- nobody still reads this rank's window
//   one-shot (small buckets: latency matters more than bytes)
//     barrier; every rank adds the whole bucket of all ranks, in rank order,
//     one 16-byte vector per thread, kept in registers; barrier (no peer reads
//     this rank's input any more); the sums replace the input
//
// Every element is summed by one rank (two-shot) or by all ranks in the same
// order (one-shot), so the replicas of a data-parallel job stay bit-identical.
// The reference has a single ring all-reduce from the vendor library
// (src/io/communicator.cc:255-301).
//
// A barrier is flag traffic between workgroups with the same index: workgroup
// b of rank r stores its epoch into word [b][r] of every peer's signal area
// and waits until the W words [b][*] of its own area have reached that epoch.
// Epochs only grow and live in device memory, so the kernel can be replayed
// from a hipGraph without host involvement.  A wait is bounded by a wall-clock
// limit: when it expires the workgroup sets a sticky error word in host memory
// and runs to its end without waiting again, so the grid always drains.
#include <string.h>

#include "common.cuh"
#include "kernels.h"

namespace sa {

__device__ __forceinline__ int64_t i64min_d(int64_t a, int64_t b) {
  return a < b ? a : b;
}

__device__ __forceinline__ unsigned ld_flag(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_flag(unsigned *p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// All threads of the workgroup call this.  `epoch` is the value this barrier
// publishes; `dead` (shared) turns the remaining barriers of the kernel into
// no-ops once a wait has timed out.
__device__ __forceinline__ void peer_barrier(const XgmiPeers &pr, int rank,
                                             int world, unsigned epoch,
                                             long long limit,
                                             volatile int *dead) {
  // this thread's stores (to its own or a peer's memory) are performed
  // before any flag goes out
  __threadfence_system();
  __syncthreads();
  const int t = threadIdx.x;
  if (t < world && !*dead) {
    st_flag(pr.flag[t] + blockIdx.x * kXgmiMaxRanks + rank, epoch);
    const unsigned *mine = pr.flag[rank] + blockIdx.x * kXgmiMaxRanks + t;
    const long long start = wall_clock64();
    // epochs wrap after 2^32 barriers: compare as a signed distance
    while ((int)(ld_flag(mine) - epoch) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - start > limit) {
        *dead = 1;
        __hip_atomic_store(pr.error, 1 + t, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  // loads after the barrier must not be served from lines cached before it
  __threadfence_system();
}

template <typename T> struct Vec16 {
  static constexpr int kN = 16 / sizeof(T);
  uint4 raw;
  __device__ __forceinline__ void add_to(float (&acc)[kN]) const {
    const T *v = reinterpret_cast<const T *>(&raw);
#pragma unroll
    for (int i = 0; i < kN; ++i) acc[i] += to_f<T>(v[i]);
  }
  __device__ __forceinline__ void from(const float (&acc)[kN]) {
    T *v = reinterpret_cast<T *>(&raw);
#pragma unroll
    for (int i = 0; i < kN; ++i) v[i] = from_f<T>(acc[i]);
  }
};

// sum over the windows of vector `i` (16 bytes), in rank order
template <typename T, int W>
__device__ __forceinline__ uint4 sum_peers(const XgmiPeers &pr, int64_t off,
                                           int64_t i, int world) {
  constexpr int kN = Vec16<T>::kN;
  Vec16<T> in[W];
  // all loads are issued before the first is consumed
#pragma unroll
  for (int p = 0; p < W; ++p)
    if (p < world)
      in[p].raw = reinterpret_cast<const uint4 *>(pr.data[p] + off)[i];
  float acc[kN];
#pragma unroll
  for (int k = 0; k < kN; ++k) acc[k] = 0.f;
#pragma unroll
  for (int p = 0; p < W; ++p)
    if (p < world) in[p].add_to(acc);
  Vec16<T> out;
  out.from(acc);
  return out.raw;
}

// Vectors i, i + step, ... (kXgmiUnroll of them) at once: world x kXgmiUnroll
// loads are in flight per thread before the first sum is formed.  Window
// memory is uncached and a peer's sits behind a link, so the latency of one
// access is long; the bytes in flight are what sets the rate.  Two, not more:
// with eight peers that is 64 registers of loads per thread, and the kernels
// keep to 128 registers (four waves per SIMD) so that the ranks of a
// rehearsal that share one GPU are all resident at once -- a barrier between
// workgroups that cannot all be scheduled would only end by its time limit.
constexpr int kXgmiUnroll = 2;

template <typename T, int W>
__device__ __forceinline__ void sum_peers_many(const XgmiPeers &pr, int64_t off,
                                               int64_t i, int64_t step,
                                               int64_t hi, int world,
                                               uint4 *__restrict__ dst) {
  constexpr int kN = Vec16<T>::kN;
  Vec16<T> in[kXgmiUnroll][W];
#pragma unroll
  for (int u = 0; u < kXgmiUnroll; ++u)
#pragma unroll
    for (int p = 0; p < W; ++p)
      if (p < world && i + u * step < hi)
        in[u][p].raw =
            reinterpret_cast<const uint4 *>(pr.data[p] + off)[i + u * step];
#pragma unroll
  for (int u = 0; u < kXgmiUnroll; ++u) {
    if (i + u * step >= hi) break;
    float acc[kN];
#pragma unroll
    for (int k = 0; k < kN; ++k) acc[k] = 0.f;
#pragma unroll
    for (int p = 0; p < W; ++p)
      if (p < world) in[u][p].add_to(acc);
    Vec16<T> out;
    out.from(acc);
    dst[i + u * step] = out.raw;
  }
}

// nvec 16-byte vectors at byte offset `off` of every window
template <typename T, bool kGather>
__global__ void __launch_bounds__(kXgmiThreads, 4)
xgmi_allreduce_two_shot_kernel(XgmiPeers pr, int rank, int world, int64_t off,
                               int64_t nvec, long long limit) {
  __shared__ int dead;
  __shared__ unsigned epoch0;
  if (threadIdx.x == 0) {
    dead = 0;
    epoch0 = pr.epoch[blockIdx.x];
  }
  __syncthreads();
  const unsigned e = epoch0;
  peer_barrier(pr, rank, world, e + 1, limit, &dead);
  const int64_t per = (nvec + world - 1) / world;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  {
    const int64_t lo = rank * per, hi = i64min_d(lo + per, nvec);
    uint4 *mine = reinterpret_cast<uint4 *>(pr.data[rank] + off);
    for (int64_t i = lo + first; i < hi; i += kXgmiUnroll * step)
      sum_peers_many<T, kXgmiMaxRanks>(pr, off, i, step, hi, world, mine);
  }
  // no peer reads this rank's input any more once all have passed here
  peer_barrier(pr, rank, world, e + 2, limit, &dead);
  if (!kGather) {     // reduce-scatter: this rank's slice is the result
    if (threadIdx.x == 0) pr.epoch[blockIdx.x] = e + 2;
    return;
  }
  {
    uint4 *mine = reinterpret_cast<uint4 *>(pr.data[rank] + off);
    // start with the next rank so that at any moment the W ranks read from W
    // different peers (each over its own link)
    for (int d = 1; d < world; ++d) {
      const int p = (rank + d) % world;
      const int64_t lo = p * per, hi = i64min_d(lo + per, nvec);
      const uint4 *src = reinterpret_cast<const uint4 *>(pr.data[p] + off);
      for (int64_t i = lo + first; i < hi; i += kXgmiUnroll * step) {
        uint4 v[kXgmiUnroll];
#pragma unroll
        for (int u = 0; u < kXgmiUnroll; ++u)
          if (i + u * step < hi) v[u] = src[i + u * step];
#pragma unroll
        for (int u = 0; u < kXgmiUnroll; ++u)
          if (i + u * step < hi) mine[i + u * step] = v[u];
      }
    }
  }
  peer_barrier(pr, rank, world, e + 3, limit, &dead);
  if (threadIdx.x == 0) pr.epoch[blockIdx.x] = e + 3;
}

// Buckets of at most one vector per thread of the grid: the sums wait in
// registers until no peer reads this rank's input any more, then replace it.
template <typename T>
__global__ void __launch_bounds__(kXgmiThreads)
xgmi_allreduce_one_shot_kernel(XgmiPeers pr, int rank, int world, int64_t off,
                               int64_t nvec, long long limit) {
  __shared__ int dead;
  __shared__ unsigned epoch0;
  if (threadIdx.x == 0) {
    dead = 0;
    epoch0 = pr.epoch[blockIdx.x];
  }
  __syncthreads();
  const unsigned e = epoch0;
  peer_barrier(pr, rank, world, e + 1, limit, &dead);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 sum = make_uint4(0, 0, 0, 0);
  if (i < nvec) sum = sum_peers<T, kXgmiMaxRanks>(pr, off, i, world);
  peer_barrier(pr, rank, world, e + 2, limit, &dead);
  if (i < nvec) reinterpret_cast<uint4 *>(pr.data[rank] + off)[i] = sum;
  if (threadIdx.x == 0) pr.epoch[blockIdx.x] = e + 2;
}

// out (local, `goff` bytes into this rank's window, W * nvec vectors) = the
// nvec vectors at `off` of every window, in rank order.  V is uint4, or
// uint32_t for payloads that are not a whole number of 16-byte vectors.
template <typename V>
__global__ void __launch_bounds__(kXgmiThreads)
xgmi_allgather_kernel(XgmiPeers pr, int rank, int world, int64_t off,
                      int64_t goff, int64_t nvec, int64_t sstride,
                      long long limit) {
  __shared__ int dead;
  __shared__ unsigned epoch0;
  if (threadIdx.x == 0) {
    dead = 0;
    epoch0 = pr.epoch[blockIdx.x];
  }
  __syncthreads();
  const unsigned e = epoch0;
  peer_barrier(pr, rank, world, e + 1, limit, &dead);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  V *out = reinterpret_cast<V *>(pr.data[rank] + goff);
  for (int d = 0; d < world; ++d) {
    const int p = (rank + d) % world;
    // sstride 0: every reader takes the same nvec vectors of peer p
    // (all-gather); nvec: reader r takes the r-th chunk (all-to-all)
    const V *src = reinterpret_cast<const V *>(pr.data[p] + off) +
                   (int64_t)rank * sstride;
    for (int64_t i = first; i < nvec; i += kXgmiUnroll * step) {
      V v[kXgmiUnroll];
#pragma unroll
      for (int u = 0; u < kXgmiUnroll; ++u)
        if (i + u * step < nvec) v[u] = src[i + u * step];
#pragma unroll
      for (int u = 0; u < kXgmiUnroll; ++u)
        if (i + u * step < nvec) out[p * nvec + i + u * step] = v[u];
    }
  }
  peer_barrier(pr, rank, world, e + 2, limit, &dead);
  if (threadIdx.x == 0) pr.epoch[blockIdx.x] = e + 2;
}

inline int xgmi_blocks(int64_t nvec) {
  int64_t b = (nvec + kXgmiThreads * 4 - 1) / (kXgmiThreads * 4);
  if (b < 1) b = 1;
  if (b > kXgmiMaxBlocks) b = kXgmiMaxBlocks;
  return (int)b;
}

inline bool xgmi_args_ok(const XgmiPeers &pr, int rank, int world,
                         int64_t off, int64_t nvec) {
  if (world < 1 || world > kXgmiMaxRanks || rank < 0 || rank >= world)
    return false;
  if (nvec <= 0 || (off & 15) != 0 || off < 0) return false;
  if (!pr.error || !pr.epoch) return false;
  for (int p = 0; p < world; ++p)
    if (!pr.data[p] || !pr.flag[p] ||
        (reinterpret_cast<uintptr_t>(pr.data[p]) & 15) != 0)
      return false;
  return true;
}

}  // namespace sa

using namespace sa;

int64_t sa_xgmi_signal_bytes() {
  // flags [kXgmiMaxBlocks][kXgmiMaxRanks], then epochs [kXgmiMaxBlocks]
  return (int64_t)sizeof(unsigned) * kXgmiMaxBlocks * (kXgmiMaxRanks + 1);
}

int sa_xgmi_alloc(int64_t bytes, void **out) {
  void *p = nullptr;
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached) !=
      hipSuccess) {
    (void)hipGetLastError();
    return -2;
  }
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    return -2;
  }
  *out = p;
  return 0;
}

int sa_xgmi_free(void *p) { return hipFree(p) == hipSuccess ? 0 : -2; }

int sa_xgmi_export(void *p, char *handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipGetLastError();
    return -2;
  }
  memcpy(handle64, &h, 64);
  return 0;
}

int sa_xgmi_import(const char *handle64, void **out) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void *p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) !=
      hipSuccess) {
    (void)hipGetLastError();
    return -2;
  }
  *out = p;
  return 0;
}

int sa_xgmi_close(void *p) {
  return hipIpcCloseMemHandle(p) == hipSuccess ? 0 : -2;
}

int sa_xgmi_error_word(void **host, void **device) {
  void *h = nullptr, *d = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return -2;
  memset(h, 0, 64);
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    (void)hipHostFree(h);
    return -2;
  }
  *host = h;
  *device = d;
  return 0;
}

int sa_xgmi_error_word_free(void *host) {
  return hipHostFree(host) == hipSuccess ? 0 : -2;
}

static long long xgmi_ticks(double seconds) {
  // wall_clock64 counts at 100 MHz on gfx9
  const double t = seconds * 1e8;
  return t < 1e3 ? 1000 : (long long)t;
}

int64_t sa_xgmi_one_shot_max_bytes() {
  return (int64_t)16 * kXgmiMaxBlocks * kXgmiThreads;
}

int sa_xgmi_allreduce(int dtype, const sa::XgmiPeers *peers, int rank,
                      int world, int64_t off, int64_t bytes, int one_shot,
                      double timeout_s, hipStream_t s) {
  if ((bytes & 15) != 0) return -2;
  const int64_t nvec = bytes >> 4;
  if (!xgmi_args_ok(*peers, rank, world, off, nvec)) return -2;
  if (one_shot && bytes > sa_xgmi_one_shot_max_bytes()) return -2;
  const long long limit = xgmi_ticks(timeout_s);
  SA_DISPATCH_FLOAT(dtype, T, {
    if (one_shot)
      hipLaunchKernelGGL((xgmi_allreduce_one_shot_kernel<T>),
                         dim3((unsigned)((nvec + kXgmiThreads - 1) / kXgmiThreads)),
                         dim3(kXgmiThreads), 0, s, *peers, rank, world, off,
                         nvec, limit);
    else
      hipLaunchKernelGGL((xgmi_allreduce_two_shot_kernel<T, true>),
                         dim3(xgmi_blocks((nvec + world - 1) / world)),
                         dim3(kXgmiThreads), 0, s, *peers, rank, world, off,
                         nvec, limit);
  })
  return 0;
}

static int xgmi_gather_launch(const sa::XgmiPeers *peers, int rank, int world,
                              int64_t off, int64_t gathered_off,
                              int64_t bytes, bool chunked, double timeout_s,
                              hipStream_t s) {
  if ((bytes & 3) != 0 || (gathered_off & 15) != 0 || gathered_off < 0)
    return -2;
  // chunks of an all-to-all start at multiples of the chunk size: 16-byte
  // vectors need 16-byte chunks
  const bool wide = (bytes & 15) == 0;
  const int64_t nvec = wide ? bytes >> 4 : bytes >> 2;
  if (!xgmi_args_ok(*peers, rank, world, off, nvec)) return -2;
  const int64_t sstride = chunked ? nvec : 0;
  if (wide)
    hipLaunchKernelGGL(xgmi_allgather_kernel<uint4>, dim3(xgmi_blocks(nvec)),
                       dim3(kXgmiThreads), 0, s, *peers, rank, world, off,
                       gathered_off, nvec, sstride, xgmi_ticks(timeout_s));
  else
    hipLaunchKernelGGL(xgmi_allgather_kernel<uint32_t>,
                       dim3(xgmi_blocks(nvec)), dim3(kXgmiThreads), 0, s,
                       *peers, rank, world, off, gathered_off, nvec, sstride,
                       xgmi_ticks(timeout_s));
  return 0;
}

int sa_xgmi_allgather(const sa::XgmiPeers *peers, int rank, int world,
                      int64_t off, int64_t gathered_off, int64_t bytes,
                      double timeout_s, hipStream_t s) {
  return xgmi_gather_launch(peers, rank, world, off, gathered_off, bytes,
                            false, timeout_s, s);
}

// Every rank holds world chunks of chunk_bytes at off; afterwards the p-th
// chunk at gathered_off is the chunk that rank p held for this rank.
int sa_xgmi_alltoall(const sa::XgmiPeers *peers, int rank, int world,
                     int64_t off, int64_t gathered_off, int64_t chunk_bytes,
                     double timeout_s, hipStream_t s) {
  return xgmi_gather_launch(peers, rank, world, off, gathered_off,
                            chunk_bytes, true, timeout_s, s);
}

// Sum over the ranks of the bytes at off, cut into world slices of equal
// size: afterwards slice `rank` of this rank's buffer holds the sum (the
// other slices keep this rank's input).
int sa_xgmi_reduce_scatter(int dtype, const sa::XgmiPeers *peers, int rank,
                           int world, int64_t off, int64_t bytes,
                           double timeout_s, hipStream_t s) {
  if ((bytes & 15) != 0 || world < 1 || ((bytes >> 4) % world) != 0)
    return -2;
  const int64_t nvec = bytes >> 4;
  if (!xgmi_args_ok(*peers, rank, world, off, nvec)) return -2;
  const long long limit = xgmi_ticks(timeout_s);
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((xgmi_allreduce_two_shot_kernel<T, false>),
                       dim3(xgmi_blocks(nvec / world)), dim3(kXgmiThreads), 0,
                       s, *peers, rank, world, off, nvec, limit);
  })
  return 0;
}
