// GEMM on the CDNA4 matrix cores.
//
//   C[b] = act(alpha * A[b] * B[b] + beta * C[b] + bias)
//
// float32 uses v_mfma_f32_32x32x2_f32 (exact f32 products and accumulation, the
// only f32 matrix path on gfx950); fp16 / bf16 use v_mfma_f32_32x32x16_{f16,
// bf16} with float accumulation.  Operands are described by element strides,
// so transposed views produced by autograd (x^T, W^T) are consumed directly.
//
// Block = 256 threads = 4 waves in a 2 x 2 arrangement; tile BM x BN x BK
// with BM, BN in {64, 128}.  A and B tiles are staged global -> registers ->
// LDS; the loads of tile t+1 are issued before the MFMAs of tile t so their
// latency hides under matrix work.  For f32 both LDS tiles are k-major
// ([k][m], [k][n]) because the 32x32x2 operand layout wants lane = row index
// for a fixed k, which makes every ds_read_b32 conflict free.  For 16-bit
// types the tiles are [m][k] / [n][k] with k contiguous (8 values per lane).
//
// Replaces cublasSgemm / cublasHgemm / cublasSgemmStridedBatched / cublasSgemv
// of the reference (tensor_math_cuda.h:1129-1272).
#include <algorithm>
#include <type_traits>

#include "common.cuh"
#include "kernels.h"

namespace sa {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

enum { LD_GENERIC = 0, LD_VEC_K = 1, LD_VEC_MN = 2 };

struct GemmArgs {
  int64_t M, N, K;
  float alpha, beta;
  const void *A;
  int64_t a_rs, a_cs, a_bs;
  const void *B;
  int64_t b_rs, b_cs, b_bs;
  void *C;
  int64_t ldc, c_bs;
  const void *bias;
  int act;  // 0 none, 1 relu
  // 16-bit tiled kernel only: C is a float buffer receiving the plain
  // product (the partial products of a split reduction)
  int c_float = 0;
  // 16-bit tiled kernel only: tensor of C's shape and layout added to the
  // product (a residual gradient that reached the same tensor)
  const void *addend = nullptr;
  // float32 kernel only: total reduction length when the batch index walks
  // slices of K reduction indices each (the last slice may be shorter); 0 when
  // every batch entry reduces over K
  int64_t k_total = 0;
};

// ---------------------------------------------------------------------------
// float32
// ---------------------------------------------------------------------------
template <int BM, int BN, int AMODE, int BMODE>
__global__ void __launch_bounds__(256)
gemm_f32_kernel(GemmArgs g) {
  constexpr int BK = 16;  // 32 measured 3-7 % slower (fewer resident workgroups)
  constexpr int WM = BM / 2, WN = BN / 2;  // per-wave tile
  constexpr int TM = WM / 32, TN = WN / 32;  // 32x32 MFMA blocks per wave
  constexpr int LDA = BM + 4, LDB = BN + 4;  // keep 16-byte row alignment
  constexpr int A_PER = BM * BK / 256, B_PER = BN * BK / 256;
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  // 1-d grid (padded to a multiple of 8): XCD x works through the x-th eighth
  // of the tile list in bands of 8 tile rows, so that the workgroups resident
  // on one XCD share A and B panels through its L2
  int64_t m0, n0;
  {
    const int ntm = (int)((g.M + BM - 1) / BM), ntn = (int)((g.N + BN - 1) / BN);
    const int w = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (w >= ntm * ntn) return;
    const int band = 8 * ntn;
    const int first = (w / band) * 8;
    const int rows = min(8, ntm - first);
    const int in_band = w - (w / band) * band;
    m0 = (int64_t)(first + in_band % rows) * BM;
    n0 = (int64_t)(in_band / rows) * BN;
  }
  const float *A = (const float *)g.A + (int64_t)blockIdx.z * g.a_bs;
  const float *B = (const float *)g.B + (int64_t)blockIdx.z * g.b_bs;
  float *C = (float *)g.C + (int64_t)blockIdx.z * g.c_bs;
  // reduction length of this batch entry
  const int64_t Kz = g.k_total
                         ? max((int64_t)0, min(g.K, g.k_total -
                                                        (int64_t)blockIdx.z * g.K))
                         : g.K;

  float ra[A_PER], rb[B_PER];

  auto load_tiles = [&](int64_t k0) {
    // ---- A tile: BM x BK ----
    if (AMODE == LD_VEC_K) {
      // float4 along k; thread -> (m, k quad)
#pragma unroll
      for (int j = 0; j < A_PER / 4; ++j) {
        const int e = tid + 256 * j;
        const int m = e % BM, kq = e / BM;
        const int64_t gm = m0 + m, gk = k0 + 4 * kq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gm < g.M && gk < Kz)
          v = *reinterpret_cast<const float4 *>(A + gm * g.a_rs + gk);
        ra[4 * j + 0] = v.x; ra[4 * j + 1] = v.y;
        ra[4 * j + 2] = v.z; ra[4 * j + 3] = v.w;
      }
    } else if (AMODE == LD_VEC_MN) {
      // float4 along m; thread -> (m quad, k)
#pragma unroll
      for (int j = 0; j < A_PER / 4; ++j) {
        const int e = tid + 256 * j;
        const int mq = e % (BM / 4), k = e / (BM / 4);
        const int64_t gm = m0 + 4 * mq, gk = k0 + k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gm < g.M && gk < Kz)
          v = *reinterpret_cast<const float4 *>(A + gk * g.a_cs + gm);
        ra[4 * j + 0] = v.x; ra[4 * j + 1] = v.y;
        ra[4 * j + 2] = v.z; ra[4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < A_PER; ++j) {
        const int e = tid + 256 * j;
        const int m = e % BM, k = e / BM;
        const int64_t gm = m0 + m, gk = k0 + k;
        ra[j] = (gm < g.M && gk < Kz) ? A[gm * g.a_rs + gk * g.a_cs] : 0.f;
      }
    }
    // ---- B tile: BK x BN ----
    if (BMODE == LD_VEC_MN) {
#pragma unroll
      for (int j = 0; j < B_PER / 4; ++j) {
        const int e = tid + 256 * j;
        const int nq = e % (BN / 4), k = e / (BN / 4);
        const int64_t gn = n0 + 4 * nq, gk = k0 + k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gn < g.N && gk < Kz)
          v = *reinterpret_cast<const float4 *>(B + gk * g.b_rs + gn);
        rb[4 * j + 0] = v.x; rb[4 * j + 1] = v.y;
        rb[4 * j + 2] = v.z; rb[4 * j + 3] = v.w;
      }
    } else if (BMODE == LD_VEC_K) {
#pragma unroll
      for (int j = 0; j < B_PER / 4; ++j) {
        const int e = tid + 256 * j;
        const int n = e % BN, kq = e / BN;
        const int64_t gn = n0 + n, gk = k0 + 4 * kq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gn < g.N && gk < Kz)
          v = *reinterpret_cast<const float4 *>(B + gn * g.b_cs + gk);
        rb[4 * j + 0] = v.x; rb[4 * j + 1] = v.y;
        rb[4 * j + 2] = v.z; rb[4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < B_PER; ++j) {
        const int e = tid + 256 * j;
        const int n = e % BN, k = e / BN;
        const int64_t gn = n0 + n, gk = k0 + k;
        rb[j] = (gn < g.N && gk < Kz) ? B[gk * g.b_rs + gn * g.b_cs] : 0.f;
      }
    }
  };

  auto store_tiles = [&]() {
    if (AMODE == LD_VEC_K) {
#pragma unroll
      for (int j = 0; j < A_PER / 4; ++j) {
        const int e = tid + 256 * j;
        const int m = e % BM, kq = e / BM;
#pragma unroll
        for (int i = 0; i < 4; ++i) As[(4 * kq + i) * LDA + m] = ra[4 * j + i];
      }
    } else if (AMODE == LD_VEC_MN) {
#pragma unroll
      for (int j = 0; j < A_PER / 4; ++j) {
        const int e = tid + 256 * j;
        const int mq = e % (BM / 4), k = e / (BM / 4);
        *reinterpret_cast<float4 *>(&As[k * LDA + 4 * mq]) =
            make_float4(ra[4 * j], ra[4 * j + 1], ra[4 * j + 2], ra[4 * j + 3]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < A_PER; ++j) {
        const int e = tid + 256 * j;
        As[(e / BM) * LDA + (e % BM)] = ra[j];
      }
    }
    if (BMODE == LD_VEC_MN) {
#pragma unroll
      for (int j = 0; j < B_PER / 4; ++j) {
        const int e = tid + 256 * j;
        const int nq = e % (BN / 4), k = e / (BN / 4);
        *reinterpret_cast<float4 *>(&Bs[k * LDB + 4 * nq]) =
            make_float4(rb[4 * j], rb[4 * j + 1], rb[4 * j + 2], rb[4 * j + 3]);
      }
    } else if (BMODE == LD_VEC_K) {
#pragma unroll
      for (int j = 0; j < B_PER / 4; ++j) {
        const int e = tid + 256 * j;
        const int n = e % BN, kq = e / BN;
#pragma unroll
        for (int i = 0; i < 4; ++i) Bs[(4 * kq + i) * LDB + n] = rb[4 * j + i];
      }
    } else {
#pragma unroll
      for (int j = 0; j < B_PER; ++j) {
        const int e = tid + 256 * j;
        Bs[(e / BN) * LDB + (e % BN)] = rb[j];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = (int)((Kz + BK - 1) / BK);
  load_tiles(0);
  store_tiles();
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) load_tiles((int64_t)(t + 1) * BK);
    // operands of reduction step ks + 2 are read from LDS before the MFMAs
    // of step ks issue
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
      a[0][i] = As[lk * LDA + wm * WM + 32 * i + lrow];
#pragma unroll
    for (int j = 0; j < TN; ++j)
      b[0][j] = Bs[lk * LDB + wn * WN + 32 * j + lrow];
#pragma unroll
    for (int ks = 0; ks < BK; ks += 2) {
      const int cur = (ks >> 1) & 1, nxt = cur ^ 1;
      if (ks + 2 < BK) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[nxt][i] = As[(ks + 2 + lk) * LDA + wm * WM + 32 * i + lrow];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[nxt][j] = Bs[(ks + 2 + lk) * LDB + wn * WN + 32 * j + lrow];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
              a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (t + 1 < nt) {
      store_tiles();
      __syncthreads();
    }
  }

  // epilogue: lane holds column (lane & 31); register r holds row
  // (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) of its 32x32 block
  const float *bias = (const float *)g.bias;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + wn * WN + 32 * j + (lane & 31);
      if (col >= g.N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row =
            m0 + wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        float v = g.alpha * acc[i][j][r] + bv;
        float *cp = C + row * g.ldc + col;
        if (g.beta != 0.f) v += g.beta * (*cp);
        if (g.act == 1) v = fmaxf(v, 0.f);
        *cp = v;
      }
    }
}

// ---------------------------------------------------------------------------
// float32, tile family with two LDS stages
// ---------------------------------------------------------------------------
// Four waves arranged WVM x WVN, each owning TM x TN accumulator blocks of
// 32 x 32: workgroup tile (32 TM WVM) x (32 TN WVN) x 16.  A wave that owns
// 2 x 2 or more blocks issues four or more 64-cycle MFMAs per four operand
// reads.  The LDS tiles are double buffered: the global loads of reduction
// step t + 1 are in flight during the MFMAs of step t and land in the other
// stage, so a step costs ONE workgroup barrier.  The tile shape is chosen per
// problem (f32_tile_for) so that the tile count divides well over 256 CUs.
//
// One operand tile of ROWS rows (m or n) x BK reduction indices, staged in
// float4 units.  MODE says which index is contiguous in memory.
template <int ROWS, int BK, int MODE>
struct F32Stage {
  static constexpr int UNITS = ROWS * BK / 4;
  static constexpr int PER = (UNITS + 255) / 256;
  static constexpr int LD = ROWS + 4;  // rows keep 16-byte alignment
  float4 r[PER];

  // rs / ks: element strides of the row index and of the reduction index
  __device__ __forceinline__ void load(const float *__restrict__ P, int tid,
                                       int64_t r0, int64_t k0, int64_t R,
                                       int64_t Kz, int64_t rs, int64_t ks) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int u = tid + 256 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (UNITS % 256 == 0 || u < UNITS) {
        if (MODE == LD_VEC_K) {
          const int row = u % ROWS, kq = u / ROWS;
          const int64_t gr = r0 + row, gk = k0 + 4 * kq;
          if (gr < R && gk < Kz)
            v = *reinterpret_cast<const float4 *>(P + gr * rs + gk);
        } else if (MODE == LD_VEC_MN) {
          const int rq = u % (ROWS / 4), k = u / (ROWS / 4);
          const int64_t gr = r0 + 4 * rq, gk = k0 + k;
          if (gr < R && gk < Kz)
            v = *reinterpret_cast<const float4 *>(P + gk * ks + gr);
        } else {
          float e[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int el = 4 * u + i;
            const int row = el % ROWS, k = el / ROWS;
            const int64_t gr = r0 + row, gk = k0 + k;
            e[i] = (gr < R && gk < Kz) ? P[gr * rs + gk * ks] : 0.f;
          }
          v = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      r[j] = v;
    }
  }

  // S: float[BK][LD], reduction index major
  __device__ __forceinline__ void store(float *__restrict__ S, int tid) const {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int u = tid + 256 * j;
      if (UNITS % 256 != 0 && u >= UNITS) continue;
      if (MODE == LD_VEC_K) {
        const int row = u % ROWS, kq = u / ROWS;
        S[(4 * kq + 0) * LD + row] = r[j].x;
        S[(4 * kq + 1) * LD + row] = r[j].y;
        S[(4 * kq + 2) * LD + row] = r[j].z;
        S[(4 * kq + 3) * LD + row] = r[j].w;
      } else if (MODE == LD_VEC_MN) {
        const int rq = u % (ROWS / 4), k = u / (ROWS / 4);
        *reinterpret_cast<float4 *>(&S[k * LD + 4 * rq]) = r[j];
      } else {
        const float e[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int el = 4 * u + i;
          S[(el / ROWS) * LD + (el % ROWS)] = e[i];
        }
      }
    }
  }
};

template <int WVM, int WVN, int TM, int TN, int AMODE, int BMODE>
__global__ void __launch_bounds__(256)
gemm_f32_tiles_kernel(GemmArgs g) {
  static_assert(WVM * WVN == 4, "four waves");
  constexpr int BK = 16;
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  using StageA = F32Stage<BM, BK, AMODE>;
  using StageB = F32Stage<BN, BK, BMODE>;
  constexpr int LDA = StageA::LD, LDB = StageB::LD;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WVN, wn = wid % WVN;
  // XCD x walks the x-th eighth of the tile list in bands of 8 tile rows (see
  // gemm_f32_kernel)
  int64_t m0, n0;
  {
    const int ntm = (int)((g.M + BM - 1) / BM), ntn = (int)((g.N + BN - 1) / BN);
    const int w = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (w >= ntm * ntn) return;
    const int band = 8 * ntn;
    const int first = (w / band) * 8;
    const int rows = min(8, ntm - first);
    const int in_band = w - (w / band) * band;
    m0 = (int64_t)(first + in_band % rows) * BM;
    n0 = (int64_t)(in_band / rows) * BN;
  }
  const float *A = (const float *)g.A + (int64_t)blockIdx.z * g.a_bs;
  const float *B = (const float *)g.B + (int64_t)blockIdx.z * g.b_bs;
  float *C = (float *)g.C + (int64_t)blockIdx.z * g.c_bs;
  const int64_t Kz = g.k_total
                         ? max((int64_t)0, min(g.K, g.k_total -
                                                        (int64_t)blockIdx.z * g.K))
                         : g.K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  StageA sa;
  StageB sb;
  const int nt = (int)((Kz + BK - 1) / BK);
  sa.load(A, tid, m0, 0, g.M, Kz, g.a_rs, g.a_cs);
  sb.load(B, tid, n0, 0, g.N, Kz, g.b_cs, g.b_rs);
  sa.store(As[0], tid);
  sb.store(Bs[0], tid);
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  const int a_off = lk * LDA + wm * (32 * TM) + lrow;
  const int b_off = lk * LDB + wn * (32 * TN) + lrow;
  for (int t = 0; t < nt; ++t) {
    const float *as = As[t & 1], *bs = Bs[t & 1];
    if (t + 1 < nt) {
      sa.load(A, tid, m0, (int64_t)(t + 1) * BK, g.M, Kz, g.a_rs, g.a_cs);
      sb.load(B, tid, n0, (int64_t)(t + 1) * BK, g.N, Kz, g.b_cs, g.b_rs);
    }
#pragma unroll
    for (int ks = 0; ks < BK; ks += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[ks * LDA + a_off + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[ks * LDB + b_off + 32 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j],
                                                           acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) {
      // the other stage was last read in step t - 1, before the barrier that
      // ended it
      sa.store(As[(t + 1) & 1], tid);
      sb.store(Bs[(t + 1) & 1], tid);
    }
    __syncthreads();
  }

  const float *bias = (const float *)g.bias;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + wn * (32 * TN) + 32 * j + (lane & 31);
      if (col >= g.N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * (32 * TM) + 32 * i + (r & 3) +
                            8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        float v = g.alpha * acc[i][j][r] + bv;
        float *cp = C + row * g.ldc + col;
        if (g.beta != 0.f) v += g.beta * (*cp);
        if (g.act == 1) v = fmaxf(v, 0.f);
        *cp = v;
      }
    }
}

// ---------------------------------------------------------------------------
// fp16 / bf16 (float accumulation)
// ---------------------------------------------------------------------------
template <typename T> struct Half16;
template <> struct Half16<__half> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Half16<__hip_bfloat16> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// A 16-bit tile is kept as [row][k] with BK = 32 values (64 bytes) per row;
// row r's four 16-byte chunks are rotated by (r >> 1) & 3 so that the 16-byte
// fragment reads of a wave (32 rows x 2 k-halves) spread over all LDS banks.
template <typename T, int BM, int BN>
__global__ void __launch_bounds__(256)
gemm_h16_kernel(GemmArgs g) {
  constexpr int BK = 32;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_PER = BM * BK / 256, B_PER = BN * BK / 256;
  using frag = typename Half16<T>::frag;
  __shared__ __attribute__((aligned(16))) uint16_t As[BM * BK];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[BN * BK];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const uint16_t *A = (const uint16_t *)g.A + (int64_t)blockIdx.z * g.a_bs;
  const uint16_t *B = (const uint16_t *)g.B + (int64_t)blockIdx.z * g.b_bs;
  T *C = (T *)g.C + (int64_t)blockIdx.z * g.c_bs;

  uint16_t ra[A_PER], rb[B_PER];
  auto swz = [](int row, int k) {
    const int chunk = ((k >> 3) + (row >> 1)) & 3;
    return row * BK + chunk * 8 + (k & 7);
  };
  auto load_tiles = [&](int64_t k0) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      const int e = tid + 256 * j;
      // lanes walk k first when A is k-contiguous, m first otherwise
      int m, k;
      if (g.a_cs == 1) { k = e % BK; m = e / BK; }
      else { m = e % BM; k = e / BM; }
      const int64_t gm = m0 + m, gk = k0 + k;
      ra[j] = (gm < g.M && gk < g.K) ? A[gm * g.a_rs + gk * g.a_cs]
                                     : (uint16_t)0;
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int e = tid + 256 * j;
      int n, k;
      if (g.b_rs == 1) { k = e % BK; n = e / BK; }
      else { n = e % BN; k = e / BN; }
      const int64_t gn = n0 + n, gk = k0 + k;
      rb[j] = (gn < g.N && gk < g.K) ? B[gk * g.b_rs + gn * g.b_cs]
                                     : (uint16_t)0;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      const int e = tid + 256 * j;
      int m, k;
      if (g.a_cs == 1) { k = e % BK; m = e / BK; }
      else { m = e % BM; k = e / BM; }
      As[swz(m, k)] = ra[j];
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int e = tid + 256 * j;
      int n, k;
      if (g.b_rs == 1) { k = e % BK; n = e / BK; }
      else { n = e % BN; k = e / BN; }
      Bs[swz(n, k)] = rb[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = (int)((g.K + BK - 1) / BK);
  load_tiles(0);
  store_tiles();
  __syncthreads();
  const int lrow = lane & 31, lh = lane >> 5;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) load_tiles((int64_t)(t + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      frag a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WM + 32 * i + lrow;
        a[i] = *reinterpret_cast<const frag *>(&As[swz(row, ks + 8 * lh)]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WN + 32 * j + lrow;
        b[j] = *reinterpret_cast<const frag *>(&Bs[swz(row, ks + 8 * lh)]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = Half16<T>::mma(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
    if (t + 1 < nt) {
      store_tiles();
      __syncthreads();
    }
  }

  const T *bias = (const T *)g.bias;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + wn * WN + 32 * j + (lane & 31);
      if (col >= g.N) continue;
      const float bv = bias ? to_f<T>(bias[col]) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row =
            m0 + wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        float v = g.alpha * acc[i][j][r] + bv;
        T *cp = C + row * g.ldc + col;
        if (g.beta != 0.f) v += g.beta * to_f<T>(*cp);
        if (g.act == 1) v = fmaxf(v, 0.f);
        *cp = from_f<T>(v);
      }
    }
}

// ---------------------------------------------------------------------------
// fp16 / bf16 fast path: 128 x 128 x 64 tiles, 16-byte staging
// ---------------------------------------------------------------------------
// Each operand tile (128 rows of the operand's own dimension x 64 reduction
// indices, 16 KB) is fetched with 16-byte buffer loads (four per thread),
// parked in registers under the MFMAs of the previous tile and written to one
// of two LDS buffers with 16-byte stores: one barrier per 64-deep step.
//
// The LDS image depends on which index is contiguous in memory:
//   * reduction index contiguous ("KC"): image [row][64 k], 128-byte rows;
//     a lane's 32x32x16 fragment (8 consecutive k of one row) is one
//     ds_read_b128.  Chunk c of row r sits at c ^ ((r >> 1) & 7): any 16 rows
//     with distinct r & 15 hit 16 distinct 16-byte slots of the 64 banks.
//   * row index contiguous: image [k][128 rows], 256-byte rows, read with the
//     transposing ds_read_b64_tr_b16 (a 16-lane group fetches a 4 k x 16 row
//     block and each lane receives its row's 4 k values), two reads per
//     fragment.  Chunk c of k-row r sits at c ^ (((r & 3) << 2) | ((r >> 2) & 3)).
// Rows past the matrix edge read zeros (or neighbouring data) through the
// buffer resource and only feed outputs that are never stored; reduction
// indices past K are forced to zero.
namespace h16 {

constexpr int BK = 64;
constexpr int kOobOff = (int)0x80000000;
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int row_off(int r, int ch) {
  return r * 128 + ((ch ^ ((r >> 1) & 7)) << 4);
}
template <int ROWS>
__device__ __forceinline__ int kmaj_off(int r, int ch) {
  // a k-row holds ROWS / 8 chunks: the swizzle stays inside the row
  constexpr int kMask = ROWS / 8 - 1;
  return r * (ROWS * 2) +
         ((ch ^ ((((r & 3) << 2) | ((r >> 2) & 3)) & kMask)) << 4);
}

// One operand tile: ROWS x 64 elements staged by NT threads; element (i, k)
// of the matrix at base[i * rs + k * ks]; KC means ks == 1, otherwise rs == 1.
template <bool KC, int ROWS, int NT>
struct Operand {
  static constexpr int PER = ROWS * 8 / NT;  // 16-byte pieces per thread
  static constexpr int CPR = ROWS / 8;       // pieces per k-row ([k][row] image)
  rsrc_t rsrc;
  int rs, ks, row0, K;

  // byte offsets of this thread's pieces of the tile at k0
  __device__ __forceinline__ void plan(int tid, int k0, int *off) const {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (KC) {
        const int ch = tid & 7, r = (tid >> 3) + (NT / 8) * j;
        const int k = k0 + 8 * ch;
        off[j] = k < K ? ((row0 + r) * rs + k) * 2 : kOobOff;
      } else {
        const int ch = tid % CPR, kr = tid / CPR + (NT / CPR) * j;
        const int k = k0 + kr;
        off[j] = k < K ? (k * ks + row0 + 8 * ch) * 2 : kOobOff;
      }
    }
  }
  __device__ __forceinline__ void load(const int *off, u32x4 *v) const {
#pragma unroll
    for (int j = 0; j < PER; ++j)
      v[j] = __builtin_bit_cast(
          u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[j], 0, 0));
  }
  static __device__ __forceinline__ void store(int tid, char *img,
                                               const u32x4 *v) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int o = KC ? row_off((tid >> 3) + (NT / 8) * j, tid & 7)
                       : kmaj_off<ROWS>(tid / CPR + (NT / CPR) * j, tid % CPR);
      *reinterpret_cast<u32x4 *>(img + o) = v[j];
    }
  }
  // fragment of rows [r0, r0 + 32) for reduction step ks16 (16 indices)
  static __device__ __forceinline__ u32x4 frag(const char *img, int r0,
                                               int ks16, int lane) {
    if (KC) {
      return *reinterpret_cast<const u32x4 *>(
          img + row_off(r0 + (lane & 31), 2 * ks16 + (lane >> 5)));
    } else {
      const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
      const int ch = ((r0 + 16 * ((lane >> 4) & 1)) >> 3) + (p >> 1);
      const int kr = 16 * ks16 + 8 * (lane >> 5) + q;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4 *)(img + kmaj_off<ROWS>(kr, ch) + 8 * (p & 1)));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4 *)(img + kmaj_off<ROWS>(kr + 4, ch) + 8 * (p & 1)));
      typedef unsigned long long u64;
      u32x4 w;
      const u64 a = __builtin_bit_cast(u64, lo), b = __builtin_bit_cast(u64, hi);
      w.x = (unsigned)a; w.y = (unsigned)(a >> 32);
      w.z = (unsigned)b; w.w = (unsigned)(b >> 32);
      return w;
    }
  }
};

// ---- output tile through LDS ---------------------------------------------
// Elements per store of an output row: 8 (16 bytes) or 4 (8 bytes) when every
// row of C (and of the addend, which shares its layout) starts on such a
// boundary, 0 when only single elements can be stored.
__device__ __forceinline__ int store_width(const GemmArgs &g, const void *c) {
  const uintptr_t bits = reinterpret_cast<uintptr_t>(c) |
                         reinterpret_cast<uintptr_t>(g.addend) |
                         reinterpret_cast<uintptr_t>(g.bias);
  if (((g.ldc | g.c_bs | g.N) & 3) != 0 || (bits & 7) != 0) return 0;
  if (((g.ldc | g.c_bs | g.N) & 7) == 0 && (bits & 15) == 0) return 8;
  return 4;
}

// One pass of the output tile through LDS: the float accumulators of row
// blocks [i0, i0 + TI) of every wave go to a [rows][BN] float image (a wave's
// 32 lanes write 32 consecutive floats: conflict free), then each thread
// takes VEC consecutive columns of a row and stores
// C = act(alpha * tile + bias + beta * C + addend) in one access per operand,
// rounding once.
// the float accumulators of row blocks [i0, i0 + TI) of every wave -> img
template <int TM, int TN, int WVN, int TI>
__device__ __forceinline__ void stage_rows(float *img, f32x16 (&acc)[TM][TN],
                                           int i0, int wm, int wn, int lane) {
  constexpr int BN = 32 * TN * WVN;
#pragma unroll
  for (int ii = 0; ii < TI; ++ii)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row =
            (wm * TI + ii) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        img[row * BN + col] = acc[i0 + ii][j][r];
      }
    }
}

template <typename T, int TM, int TN, int WVM, int WVN, int TI, int VEC>
__device__ __forceinline__ void tile_pass(float *img, f32x16 (&acc)[TM][TN],
                                          int i0, T *C, const T *add,
                                          const T *bias, const GemmArgs &g,
                                          int m0, int n0, int wm, int wn,
                                          int tid) {
  constexpr int NT = 64 * WVM * WVN;
  constexpr int BN = 32 * TN * WVN;
  constexpr int ROWS = 32 * TI * WVM;  // rows staged per pass
  constexpr int PPR = BN / VEC;        // pieces per row
  using vec_t = typename std::conditional<VEC == 8, u32x4, uint2>::type;
  stage_rows<TM, TN, WVN, TI>(img, acc, i0, wm, wn, tid & 63);
  __syncthreads();
  for (int e = tid; e < ROWS * PPR; e += NT) {
    const int row = e / PPR, pc = e - row * PPR;
    // staged row -> tile row: wave row wm holds row blocks wm * TM + i
    const int w = row / (32 * TI), in = row - w * (32 * TI);
    const int64_t grow = m0 + (w * TM + i0) * 32 + in;
    const int64_t gcol = n0 + pc * VEC;
    if (grow >= g.M || gcol >= g.N) continue;
    float f[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k += 4) {
      const float4 q =
          *reinterpret_cast<const float4 *>(img + row * BN + pc * VEC + k);
      f[k] = g.alpha * q.x; f[k + 1] = g.alpha * q.y;
      f[k + 2] = g.alpha * q.z; f[k + 3] = g.alpha * q.w;
    }
    const int64_t at = grow * g.ldc + gcol;
    T t[VEC];
    if (bias) {
      const vec_t bw = *reinterpret_cast<const vec_t *>(bias + gcol);
      __builtin_memcpy(t, &bw, sizeof(bw));
#pragma unroll
      for (int k = 0; k < VEC; ++k) f[k] += to_f<T>(t[k]);
    }
    if (g.beta != 0.f) {
      const vec_t cw = *reinterpret_cast<const vec_t *>(C + at);
      __builtin_memcpy(t, &cw, sizeof(cw));
#pragma unroll
      for (int k = 0; k < VEC; ++k) f[k] += g.beta * to_f<T>(t[k]);
    }
    if (add) {
      const vec_t aw = *reinterpret_cast<const vec_t *>(add + at);
      __builtin_memcpy(t, &aw, sizeof(aw));
#pragma unroll
      for (int k = 0; k < VEC; ++k) f[k] += to_f<T>(t[k]);
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k)
      t[k] = from_f<T>(g.act == 1 ? fmaxf(f[k], 0.f) : f[k]);
    vec_t o;
    __builtin_memcpy(&o, t, sizeof(o));
    *reinterpret_cast<vec_t *>(C + at) = o;
  }
  __syncthreads();
}

// The same pass for the convolution over flattened positions: a piece of VEC
// columns is VEC consecutive positions of one image.
template <typename T, int TM, int TN, int WVM, int WVN, int TI, int VEC>
__device__ __forceinline__ void flat_tile_pass(
    float *img, f32x16 (&acc)[TM][TN], int i0, T *out, const T *add, int M,
    int HW, int n_img, int ppi, unsigned ppi_magic, int m0, int n0, int wm,
    int wn, int tid) {
  constexpr int NT = 64 * WVM * WVN;
  constexpr int BN = 32 * TN * WVN;
  constexpr int ROWS = 32 * TI * WVM;
  constexpr int PPR = BN / VEC;
  using vec_t = typename std::conditional<VEC == 8, u32x4, uint2>::type;
  stage_rows<TM, TN, WVN, TI>(img, acc, i0, wm, wn, tid & 63);
  __syncthreads();
  for (int e = tid; e < ROWS * PPR; e += NT) {
    const int row = e / PPR, pc = e - row * PPR;
    const int w = row / (32 * TI), in = row - w * (32 * TI);
    const int grow = m0 + (w * TM + i0) * 32 + in;
    const int col = n0 + pc * VEC;
    const int gp = col >> 3;
    const int n = (int)__umulhi((unsigned)gp, ppi_magic);
    const int p = (gp - n * ppi) * 8 + (col & 7);
    if (grow >= M || n >= n_img || p >= HW) continue;
    float f[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k += 4) {
      const float4 q =
          *reinterpret_cast<const float4 *>(img + row * BN + pc * VEC + k);
      f[k] = q.x; f[k + 1] = q.y; f[k + 2] = q.z; f[k + 3] = q.w;
    }
    const int64_t at = ((int64_t)n * M + grow) * HW + p;
    T t[VEC];
    if (add) {
      const vec_t aw = *reinterpret_cast<const vec_t *>(add + at);
      __builtin_memcpy(t, &aw, sizeof(aw));
#pragma unroll
      for (int k = 0; k < VEC; ++k) f[k] += to_f<T>(t[k]);
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) t[k] = from_f<T>(f[k]);
    vec_t o;
    __builtin_memcpy(&o, t, sizeof(o));
    *reinterpret_cast<vec_t *>(out + at) = o;
  }
  __syncthreads();
}

}  // namespace h16

// WVM x WVN waves, each computing TM x TN blocks of 32 x 32:
// 2 x 2 waves of 2 x 2 blocks = 128 x 128 tile (two workgroups per CU),
// 2 x 4 waves of 4 x 2 blocks = 256 x 256 tile (one workgroup of 8 waves per
// CU; half the LDS traffic per multiply, for problems with enough tiles).
template <typename T, bool A_KC, bool B_KC, int WVM, int WVN, int TM, int TN>
__global__ void __launch_bounds__(64 * WVM * WVN)
gemm_h16_fast_kernel(GemmArgs g, int ntm, int ntn) {
  using namespace h16;
  using frag = typename Half16<T>::frag;
  constexpr int NT = 64 * WVM * WVN;
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  using OpA = Operand<A_KC, BM, NT>;
  using OpB = Operand<B_KC, BN, NT>;
  constexpr int kImgA = BM * BK * 2, kImgB = BN * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2][kImgA + kImgB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WVN, wn = wid % WVN;

  // Workgroup ids go round-robin over the 8 XCDs: give XCD x the x-th eighth
  // of the tile list, walked in bands of 8 tile rows so that the workgroups
  // resident on one XCD share A and B panels through its L2.
  int pid_m, pid_n;
  {
    const int w = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (w >= ntm * ntn) return;
    const int band = 8 * ntn;
    const int first = (w / band) * 8;
    const int rows = min(8, ntm - first);
    const int in_band = w - (w / band) * band;
    pid_m = first + in_band % rows;
    pid_n = in_band / rows;
  }
  const int m0 = pid_m * BM, n0 = pid_n * BN;
  // reduction length of this batch entry (slices of a split reduction: the
  // last one may be shorter)
  const int K = g.k_total ? (int)max((int64_t)0,
                                     min(g.K, g.k_total -
                                                  (int64_t)blockIdx.z * g.K))
                          : (int)g.K;

  const T *Ab = (const T *)g.A + (int64_t)blockIdx.z * g.a_bs;
  const T *Bb = (const T *)g.B + (int64_t)blockIdx.z * g.b_bs;
  T *C = (T *)g.C + (int64_t)blockIdx.z * g.c_bs;
  const int64_t a_extent = ((g.M - 1) * g.a_rs + (g.K - 1) * g.a_cs + 1) * 2;
  const int64_t b_extent = ((g.K - 1) * g.b_rs + (g.N - 1) * g.b_cs + 1) * 2;
  OpA opa{__builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Ab), 0,
                                            (int)a_extent, 0x00020000),
          (int)g.a_rs, (int)g.a_cs, m0, K};
  OpB opb{__builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Bb), 0,
                                            (int)b_extent, 0x00020000),
          (int)g.b_cs, (int)g.b_rs, n0, K};

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[OpA::PER], rb[OpB::PER];
  int oa[OpA::PER], ob[OpB::PER];
  const int nt = (K + BK - 1) / BK;
  opa.plan(tid, 0, oa);
  opb.plan(tid, 0, ob);
  opa.load(oa, ra);
  opb.load(ob, rb);
  OpA::store(tid, smem[0], ra);
  OpB::store(tid, smem[0] + kImgA, rb);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const char *as = smem[t & 1], *bs = smem[t & 1] + kImgA;
    if (t + 1 < nt) {
      opa.plan(tid, (t + 1) * BK, oa);
      opb.plan(tid, (t + 1) * BK, ob);
      opa.load(oa, ra);
      opb.load(ob, rb);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      frag a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = __builtin_bit_cast(
            frag, OpA::frag(as, (wm * TM + i) * 32, ks, lane));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = __builtin_bit_cast(
            frag, OpB::frag(bs, (wn * TN + j) * 32, ks, lane));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = Half16<T>::mma(a[i], b[j], acc[i][j]);
    }
    if (t + 1 < nt) {
      // the other buffer was last read before the previous barrier
      OpA::store(tid, smem[(t + 1) & 1], ra);
      OpB::store(tid, smem[(t + 1) & 1] + kImgA, rb);
    }
    __syncthreads();
  }

  const T *bias = (const T *)g.bias;
  // 16-bit output whose rows can be written in 8- or 16-byte pieces: the
  // tile goes through LDS so that a lane stores (and reads the addend /
  // previous C in) whole pieces instead of single 2-byte elements.
  const int vec = g.c_float ? 0 : h16::store_width(g, C);
  if (vec) {
    // the operand images hold half of the tile as floats: two passes
    constexpr int TI = (TM + 1) / 2;
    static_assert(32 * TI * WVM * BN * 4 <= (int)sizeof(smem),
                  "half of the output tile must fit the operand images");
    const T *add = g.addend ? (const T *)g.addend + (int64_t)blockIdx.z * g.c_bs
                            : nullptr;
    float *img = reinterpret_cast<float *>(&smem[0][0]);
    __syncthreads();  // every wave is done reading the operand images
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += TI) {
      if (vec == 8)
        h16::tile_pass<T, TM, TN, WVM, WVN, TI, 8>(img, acc, i0, C, add, bias,
                                                   g, m0, n0, wm, wn, tid);
      else
        h16::tile_pass<T, TM, TN, WVM, WVN, TI, 4>(img, acc, i0, C, add, bias,
                                                   g, m0, n0, wm, wn, tid);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + (wn * TN + j) * 32 + (lane & 31);
      if (col >= g.N) continue;
      const float bv = bias ? to_f<T>(bias[col]) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) +
                            4 * (lane >> 5);
        if (row >= g.M) continue;
        if (g.c_float) {
          ((float *)g.C)[(int64_t)blockIdx.z * g.c_bs + row * g.ldc + col] =
              acc[i][j][r];
          continue;
        }
        float v = g.alpha * acc[i][j][r] + bv;
        T *cp = C + row * g.ldc + col;
        if (g.beta != 0.f) v += g.beta * to_f<T>(*cp);
        if (g.addend)
          v += to_f<T>(((const T *)g.addend)[(int64_t)blockIdx.z * g.c_bs +
                                            row * g.ldc + col]);
        if (g.act == 1) v = fmaxf(v, 0.f);
        *cp = from_f<T>(v);
      }
    }
}

template <typename T, int WVM, int WVN, int TM, int TN>
static void launch_h16_shape(const GemmArgs &g, bool a_kc, bool b_kc,
                             int64_t batch, hipStream_t s) {
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  const int ntm = (int)((g.M + BM - 1) / BM), ntn = (int)((g.N + BN - 1) / BN);
  const dim3 grid((unsigned)(((int64_t)ntm * ntn + 7) / 8 * 8), 1,
                  (unsigned)batch);
#define SA_H16_CASE(AK, BK_)                                                 \
  if (a_kc == AK && b_kc == BK_) {                                           \
    hipLaunchKernelGGL(                                                      \
        (gemm_h16_fast_kernel<T, AK, BK_, WVM, WVN, TM, TN>), grid,          \
        dim3(64 * WVM * WVN), 0, s, g, ntm, ntn);                            \
    return;                                                                  \
  }
  SA_H16_CASE(true, true)
  SA_H16_CASE(true, false)
  SA_H16_CASE(false, true)
  SA_H16_CASE(false, false)
#undef SA_H16_CASE
}

static const int kGemmTile = [] {
  const char *v = getenv("SA_GEMM_TILE");
  return v && *v ? atoi(v) : 0;
}();

// 16-bit GEMM tile shape: -1 chosen by the rule below; 0 128x128 (4 waves),
// 1 256x256, 2 256x128, 3 128x256 (8 waves), 4 64x128 (4 waves) for every
// problem -- what a tuner sets around one call
static int &h16_tile_override() {
  static int v = -1;
  return v;
}

template <typename T>
static void launch_h16_fast(const GemmArgs &g, bool a_kc, bool b_kc,
                            int64_t batch, hipStream_t s) {
  switch (h16_tile_override()) {
    case 0: return launch_h16_shape<T, 2, 2, 2, 2>(g, a_kc, b_kc, batch, s);
    case 1: return launch_h16_shape<T, 2, 4, 4, 2>(g, a_kc, b_kc, batch, s);
    case 2: return launch_h16_shape<T, 2, 4, 4, 1>(g, a_kc, b_kc, batch, s);
    case 3: return launch_h16_shape<T, 2, 4, 2, 2>(g, a_kc, b_kc, batch, s);
    case 4: return launch_h16_shape<T, 1, 4, 2, 1>(g, a_kc, b_kc, batch, s);
    default: break;
  }
  // 256 x 256 tiles (one workgroup per CU at a time) when their rounds over
  // the 256 CUs are at least 80 % full
  const int64_t t256 = ((g.M + 255) / 256) * ((g.N + 255) / 256) * batch;
  bool big = t256 * 5 >= (t256 + 255) / 256 * 256 * 4;
  if (kGemmTile == 128) big = false;
  if (kGemmTile == 256) big = true;
  if (big) launch_h16_shape<T, 2, 4, 4, 2>(g, a_kc, b_kc, batch, s);
  else launch_h16_shape<T, 2, 2, 2, 2>(g, a_kc, b_kc, batch, s);
}

// 64-row tiles (1 x 4 waves of 2 x 1 blocks = 64 x 128) for products with at
// most 64 rows, e.g. 1x1 convolutions with 64 output channels
template <typename T>
static void launch_h16_auto(const GemmArgs &g, bool a_kc, bool b_kc,
                            int64_t batch, hipStream_t s) {
  if (g.M <= 64) launch_h16_shape<T, 1, 4, 2, 1>(g, a_kc, b_kc, batch, s);
  else launch_h16_fast<T>(g, a_kc, b_kc, batch, s);
}

// ---------------------------------------------------------------------------
// 16-bit 3x3 / stride 1 / pad 1 convolution as a nine-tap GEMM per image
// ---------------------------------------------------------------------------
// out[n] (M x HW) = sum over taps (r, s) of  Wt[tap] (M x Kc) * shift(act[n], r, s)
// where shift() is act[n] (Kc x HW, positions contiguous) read (r - 1) rows and
// (s - 1) columns away, zero outside the image.  The reduction loop of the
// tiled 16-bit kernel simply runs over (tap, 64-channel block): the weight
// tile comes from the tap's slice of the re-laid-out filter [tap][M][Kc]
// (reduction index contiguous, ds_read_b128 fragments) and the activation tile
// is the same [channel][position] block for every tap, fetched at a different
// byte offset.  A one-column shift is a 2-byte offset: such pieces are read as
// five aligned dwords and funnel-shifted into four.  Positions whose tap falls
// outside the image are cleared with a per-thread element mask (a thread's
// pieces always cover the same eight positions, so the four border bit sets
// are computed once).  16-bit LDS images, transposing LDS reads for the
// position-contiguous activations, float accumulation, 16-bit output.
struct ConvTapArgs {
  const void *wt;      // [9][M][Kc]
  const void *act;     // [n][Kc][H * W]
  const void *addend;  // [n][M][H * W] or null
  void *out;           // [n][M][H * W]
  int M, Kc, H, W;
};

template <typename T, int WVM, int WVN, int TM, int TN>
__global__ void __launch_bounds__(64 * WVM * WVN)
conv3x3_h16_kernel(ConvTapArgs a, int ntm, int ntn) {
  using namespace h16;
  using frag = typename Half16<T>::frag;
  constexpr int NT = 64 * WVM * WVN;
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  using OpA = Operand<true, BM, NT>;
  using OpB = Operand<false, BN, NT>;
  constexpr int kImgA = BM * BK * 2, kImgB = BN * BK * 2;
  constexpr int CPR = OpB::CPR, PERB = OpB::PER;
  __shared__ __attribute__((aligned(16))) char smem[2][kImgA + kImgB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WVN, wn = wid % WVN;
  int pid_m, pid_n;
  {
    const int w = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (w >= ntm * ntn) return;
    pid_m = w % ntm;
    pid_n = w / ntm;
  }
  const int m0 = pid_m * BM, n0 = pid_n * BN;
  const int H = a.H, W = a.W, HW = H * W, Kc = a.Kc, M = a.M;

  const T *act = (const T *)a.act + (int64_t)blockIdx.z * Kc * HW;
  T *out = (T *)a.out + (int64_t)blockIdx.z * M * HW;
  const T *addend =
      a.addend ? (const T *)a.addend + (int64_t)blockIdx.z * M * HW : nullptr;
  OpA opa{__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.wt), 0,
                                            9 * M * Kc * 2, 0x00020000),
          Kc, 1, m0, Kc};
  const int limit_b = Kc * HW * 2;
  const rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T *>(act), 0, limit_b, 0x00020000);

  // the eight positions all of this thread's activation pieces cover
  const int chb = tid % CPR, krb = tid / CPR;
  const int pb = n0 + 8 * chb;
  unsigned top = 0, bottom = 0, left = 0, right = 0, beyond = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = pb + i;
    if (p >= HW) {
      beyond |= 1u << i;
    } else {
      const int h = p / W, w = p - h * W;
      if (h == 0) top |= 1u << i;
      if (h == H - 1) bottom |= 1u << i;
      if (w == 0) left |= 1u << i;
      if (w == W - 1) right |= 1u << i;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[OpA::PER], rb[PERB];
  int oa[OpA::PER];
  const int ntc = Kc / BK;
  const int steps = 9 * ntc;

  auto fetch = [&](int t) {
    const int tap = t / ntc, kc = (t - tap * ntc) * BK;
    const int r = tap / 3, sft = tap - 3 * r;
    opa.plan(tid, kc, oa);
#pragma unroll
    for (int j = 0; j < OpA::PER; ++j)
      if (oa[j] != kOobOff) oa[j] += tap * M * Kc * 2;
    opa.load(oa, ra);
    const unsigned inv = beyond | (r == 0 ? top : 0u) | (r == 2 ? bottom : 0u) |
                         (sft == 0 ? left : 0u) | (sft == 2 ? right : 0u);
    unsigned keep[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
      keep[d] = (((inv >> (2 * d)) & 1u) ? 0u : 0x0000ffffu) |
                (((inv >> (2 * d + 1)) & 1u) ? 0u : 0xffff0000u);
    const int shift = (r - 1) * W + (sft - 1);
#pragma unroll
    for (int j = 0; j < PERB; ++j) {
      const int kr = krb + (NT / CPR) * j;
      const int byte = ((kc + kr) * HW + pb + shift) * 2;
      // first aligned dword touched; an odd shift needs five dwords
      const int b0 = sft == 1 ? byte : byte - 2;
      const int ndw = sft == 1 ? 4 : 5;
      u32x4 lo;
      unsigned hi = 0;
      if (b0 >= 0 && b0 + 4 * ndw <= limit_b) {
        lo = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b0, 0, 0));
        if (sft != 1)
          hi = __builtin_bit_cast(
              unsigned,
              __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, b0 + 16, 0, 0));
      } else {
        // the piece hangs over an end of the image block: dword by dword, each
        // one range-checked on its own (offsets below zero are out of range)
        // (each offset is redirected on its own so that the loads cannot be
        // fused back into one wide load, which is range-checked as a whole)
        auto dw = [&](int o) {
          const int at = (o >= 0 && o + 4 <= limit_b) ? o : kOobOff;
          return __builtin_bit_cast(
              unsigned, __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, at, 0, 0));
        };
        lo.x = dw(b0); lo.y = dw(b0 + 4); lo.z = dw(b0 + 8); lo.w = dw(b0 + 12);
        if (sft != 1) hi = dw(b0 + 16);
      }
      u32x4 v = lo;
      if (sft != 1) {
        v.x = __builtin_amdgcn_alignbit(lo.y, lo.x, 16);
        v.y = __builtin_amdgcn_alignbit(lo.z, lo.y, 16);
        v.z = __builtin_amdgcn_alignbit(lo.w, lo.z, 16);
        v.w = __builtin_amdgcn_alignbit(hi, lo.w, 16);
      }
      v.x &= keep[0]; v.y &= keep[1]; v.z &= keep[2]; v.w &= keep[3];
      rb[j] = v;
    }
  };

  fetch(0);
  OpA::store(tid, smem[0], ra);
  OpB::store(tid, smem[0] + kImgA, rb);
  __syncthreads();
  for (int t = 0; t < steps; ++t) {
    const char *as = smem[t & 1], *bs = smem[t & 1] + kImgA;
    if (t + 1 < steps) fetch(t + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      frag fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = __builtin_bit_cast(
            frag, OpA::frag(as, (wm * TM + i) * 32, ks, lane));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = __builtin_bit_cast(
            frag, OpB::frag(bs, (wn * TN + j) * 32, ks, lane));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = Half16<T>::mma(fa[i], fb[j], acc[i][j]);
    }
    if (t + 1 < steps) {
      OpA::store(tid, smem[(t + 1) & 1], ra);
      OpB::store(tid, smem[(t + 1) & 1] + kImgA, rb);
    }
    __syncthreads();
  }

  if ((HW & 7) == 0 && (((int64_t)M * HW) & 7) == 0) {
    // rows of whole 16-byte pieces: the tile goes out through LDS
    constexpr int TI = (TM + 1) / 2;
    static_assert(32 * TI * WVM * BN * 4 <= (int)sizeof(smem),
                  "half of the output tile must fit the operand images");
    GemmArgs gg{};
    gg.M = M; gg.N = HW; gg.ldc = HW;
    gg.alpha = 1.f; gg.beta = 0.f; gg.act = 0;
    float *img = reinterpret_cast<float *>(&smem[0][0]);
    __syncthreads();  // every wave is done reading the operand images
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += TI)
      tile_pass<T, TM, TN, WVM, WVN, TI, 8>(img, acc, i0, out, addend, nullptr,
                                            gg, m0, n0, wm, wn, tid);
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
      if (col >= HW) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) +
                        4 * (lane >> 5);
        if (row >= M) continue;
        float v = acc[i][j][r];
        if (addend) v += to_f<T>(addend[(int64_t)row * HW + col]);
        out[(int64_t)row * HW + col] = from_f<T>(v);
      }
    }
}

// Filter re-layout for the nine-tap GEMM.  mode 0 (forward): w is [M][Kc][9],
// wt[tap][m][k] = w[m][k][tap].  mode 1 (backward-data): w is the forward
// filter [Kc][M][9] seen from dY's side, wt[tap][m][k] = w[k][m][8 - tap].
template <typename T>
__global__ void conv3x3_h16_prep_kernel(const T *__restrict__ w,
                                        T *__restrict__ wt, int M, int Kc,
                                        int mode) {
  const int64_t total = (int64_t)9 * M * Kc;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += step) {
    const int k = (int)(e % Kc);
    const int m = (int)((e / Kc) % M);
    const int tap = (int)(e / ((int64_t)Kc * M));
    wt[e] = mode == 0 ? w[((int64_t)m * Kc + k) * 9 + tap]
                      : w[((int64_t)k * M + m) * 9 + (8 - tap)];
  }
}

template <typename T, int WVM, int WVN, int TM, int TN>
static void launch_conv3x3_h16(const ConvTapArgs &a, int n_img, hipStream_t s) {
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.H * a.W + BN - 1) / BN;
  const dim3 grid((unsigned)(((int64_t)ntm * ntn + 7) / 8 * 8), 1,
                  (unsigned)n_img);
  hipLaunchKernelGGL((conv3x3_h16_kernel<T, WVM, WVN, TM, TN>), grid,
                     dim3(64 * WVM * WVN), 0, s, a, ntm, ntn);
}

// ---------------------------------------------------------------------------
// 16-bit 1x1 / 3x3 stride-1 same-size convolution over flattened positions
// ---------------------------------------------------------------------------
// The same tap GEMM as above for images of any size, small ones included:
// the GEMM's column index runs over the 8-position pieces of ALL images
// (`ppi` pieces per image, the last one of an image possibly partial), so a
// 14 x 14 or 7 x 7 layer fills whole tiles instead of one ragged tile per
// image.  A piece's elements start at any 2-byte offset (odd image sizes, odd
// tap shifts): it is fetched as five aligned dwords and funnel-shifted when
// its first element sits in the upper half of a dword (VARPAR: decided per
// thread and tile; otherwise the parity is the tap's).  All loads are
// select-based, `DEPTH` reduction tiles are in flight in registers, and only
// the few tiles that can reach past an end of the tensors (first / last image,
// first / last channel block) take the dword-by-dword EDGE variant.
struct ConvFlatArgs {
  const void *wt;      // [taps][M][Kc]
  const void *act;     // [n][Kc][HW]
  const void *addend;  // [n][M][HW] or null
  void *out;           // [n][M][HW]
  int M, Kc, H, W, n_img, taps;
  int ppi;             // pieces of 8 positions per image
  unsigned ppi_magic;  // ceil(2^32 / ppi)
};

template <typename T, int WVM, int WVN, int TM, int TN, int DEPTH, bool VARPAR>
__global__ void __launch_bounds__(64 * WVM * WVN)
conv_flat_h16_kernel(ConvFlatArgs a, int ntm, int ntn) {
  using namespace h16;
  using frag = typename Half16<T>::frag;
  constexpr int NT = 64 * WVM * WVN;
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  using OpA = Operand<true, BM, NT>;
  using OpB = Operand<false, BN, NT>;
  constexpr int kImgA = BM * BK * 2, kImgB = BN * BK * 2;
  constexpr int CPR = OpB::CPR, PERB = OpB::PER, PERA = OpA::PER;
  __shared__ __attribute__((aligned(16))) char smem[2][kImgA + kImgB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WVN, wn = wid % WVN;
  int pid_m, pid_n;
  {
    const int w = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (w >= ntm * ntn) return;
    pid_m = w % ntm;
    pid_n = w / ntm;
  }
  const int m0 = pid_m * BM, n0 = pid_n * BN;
  const int H = a.H, W = a.W, HW = H * W, Kc = a.Kc, M = a.M;
  const int n_pieces = a.n_img * a.ppi;

  OpA opa{__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.wt), 0,
                                            a.taps * M * Kc * 2, 0x00020000),
          Kc, 1, m0, Kc};
  const int limit_b = a.n_img * Kc * HW * 2;
  const rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(a.act), 0, limit_b, 0x00020000);

  // the eight positions all of this thread's activation pieces cover
  const int chb = tid % CPR, krb = tid / CPR;
  const int gb = n0 / 8 + chb;
  const int nb = (int)__umulhi((unsigned)gb, a.ppi_magic);
  const int pb = (gb - nb * a.ppi) * 8;
  const bool live = nb < a.n_img;
  const int eb = nb * Kc * HW + pb;  // element offset of the piece in row 0
  unsigned top = 0, bottom = 0, left = 0, right = 0, beyond = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = pb + i;
    if (p >= HW || !live) {
      beyond |= 1u << i;
    } else {
      const int h = p / W, w = p - h * W;
      if (h == 0) top |= 1u << i;
      if (h == H - 1) bottom |= 1u << i;
      if (w == 0) left |= 1u << i;
      if (w == W - 1) right |= 1u << i;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[DEPTH][PERA], rb[DEPTH][PERB];
  const int ntc = Kc / BK;
  const int steps = a.taps * ntc;

  auto dw = [](const rsrc_t &rs, int o, int limit) {
    const int at = (o >= 0 && o + 4 <= limit) ? o : kOobOff;
    return __builtin_bit_cast(
        unsigned, __builtin_amdgcn_raw_buffer_load_b32(rs, at, 0, 0));
  };

  auto fetch = [&](int t, u32x4 *ra, u32x4 *rb, auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    const int tap = t / ntc, kc = (t - tap * ntc) * BK;
    const int r = a.taps == 9 ? tap / 3 : 1;
    const int sft = a.taps == 9 ? tap - 3 * r : 1;
    int oa[PERA];
    opa.plan(tid, kc, oa);
#pragma unroll
    for (int j = 0; j < PERA; ++j)
      oa[j] = oa[j] != kOobOff ? oa[j] + tap * M * Kc * 2 : kOobOff;
    opa.load(oa, ra);
    const unsigned inv = beyond | (r == 0 ? top : 0u) | (r == 2 ? bottom : 0u) |
                         (sft == 0 ? left : 0u) | (sft == 2 ? right : 0u);
    unsigned keep[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
      keep[d] = (((inv >> (2 * d)) & 1u) ? 0u : 0x0000ffffu) |
                (((inv >> (2 * d + 1)) & 1u) ? 0u : 0xffff0000u);
    const int shift = (r - 1) * W + (sft - 1);
    const bool want = inv != 0xffu;
    // without VARPAR every element offset has the parity of the tap's shift
    const bool tap_odd = sft != 1;
#pragma unroll
    for (int j = 0; j < PERB; ++j) {
      const int kr = krb + (NT / CPR) * j;
      const int e0 = eb + (kc + kr) * HW + shift;
      const bool odd = VARPAR ? (e0 & 1) != 0 : tap_odd;
      const int b0 = (e0 & ~1) * 2;
      const bool five = VARPAR || tap_odd;
      u32x4 lo;
      unsigned hi = 0;
      if (EDGE) {
        lo.x = dw(rsrc_b, want ? b0 : kOobOff, limit_b);
        lo.y = dw(rsrc_b, want ? b0 + 4 : kOobOff, limit_b);
        lo.z = dw(rsrc_b, want ? b0 + 8 : kOobOff, limit_b);
        lo.w = dw(rsrc_b, want ? b0 + 12 : kOobOff, limit_b);
        if (five) hi = dw(rsrc_b, want ? b0 + 16 : kOobOff, limit_b);
      } else {
        lo = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                       rsrc_b, want ? b0 : kOobOff, 0, 0));
        if (five)
          hi = __builtin_bit_cast(
              unsigned, __builtin_amdgcn_raw_buffer_load_b32(
                            rsrc_b, want ? b0 + 16 : kOobOff, 0, 0));
      }
      u32x4 v = lo;
      if (five) {
        u32x4 sh;
        sh.x = __builtin_amdgcn_alignbit(lo.y, lo.x, 16);
        sh.y = __builtin_amdgcn_alignbit(lo.z, lo.y, 16);
        sh.z = __builtin_amdgcn_alignbit(lo.w, lo.z, 16);
        sh.w = __builtin_amdgcn_alignbit(hi, lo.w, 16);
        v.x = odd ? sh.x : lo.x; v.y = odd ? sh.y : lo.y;
        v.z = odd ? sh.z : lo.z; v.w = odd ? sh.w : lo.w;
      }
      v.x &= keep[0]; v.y &= keep[1]; v.z &= keep[2]; v.w &= keep[3];
      rb[j] = v;
    }
  };
  // tiles that can reach past an end of the activations: workgroups holding
  // pieces of the first / last image, in the first / last channel block
  const bool wg_lo = n0 / 8 < a.ppi;
  const bool wg_hi = n0 / 8 + BN / 8 > n_pieces - a.ppi;
  auto fetch_tile = [&](int t, u32x4 *ra, u32x4 *rb) {
    const int kc = (t % ntc) * BK;
    if ((wg_lo && kc == 0) || (wg_hi && kc + BK >= Kc))
      fetch(t, ra, rb, std::true_type{});
    else
      fetch(t, ra, rb, std::false_type{});
  };

#pragma unroll
  for (int u = 0; u < DEPTH; ++u)
    if (u < steps) fetch_tile(u, ra[u], rb[u]);
  OpA::store(tid, smem[0], ra[0]);
  OpB::store(tid, smem[0] + kImgA, rb[0]);
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < steps; t += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int tt = t + u;
      if (tt >= steps) break;
      const char *as = smem[cur], *bs = smem[cur] + kImgA;
      if (tt + DEPTH < steps) fetch_tile(tt + DEPTH, ra[u], rb[u]);
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        frag fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = __builtin_bit_cast(
              frag, OpA::frag(as, (wm * TM + i) * 32, ks, lane));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = __builtin_bit_cast(
              frag, OpB::frag(bs, (wn * TN + j) * 32, ks, lane));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = Half16<T>::mma(fa[i], fb[j], acc[i][j]);
      }
      if (tt + 1 < steps) {
        OpA::store(tid, smem[cur ^ 1], ra[(u + 1) % DEPTH]);
        OpB::store(tid, smem[cur ^ 1] + kImgA, rb[(u + 1) % DEPTH]);
      }
      __syncthreads();
      cur ^= 1;
    }
  }

  T *out = (T *)a.out;
  const T *addend = (const T *)a.addend;
  // whole 16- or 8-byte pieces per lane when the image rows allow it
  const int vec = (HW & 7) == 0 ? 8 : ((HW & 3) == 0 ? 4 : 0);
  if (vec) {
    constexpr int TI = (TM + 1) / 2;
    static_assert(32 * TI * WVM * BN * 4 <= (int)sizeof(smem),
                  "half of the output tile must fit the operand images");
    float *img = reinterpret_cast<float *>(&smem[0][0]);
    __syncthreads();  // every wave is done reading the operand images
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += TI) {
      if (vec == 8)
        flat_tile_pass<T, TM, TN, WVM, WVN, TI, 8>(
            img, acc, i0, out, addend, M, HW, a.n_img, a.ppi, a.ppi_magic, m0,
            n0, wm, wn, tid);
      else
        flat_tile_pass<T, TM, TN, WVM, WVN, TI, 4>(
            img, acc, i0, out, addend, M, HW, a.n_img, a.ppi, a.ppi_magic, m0,
            n0, wm, wn, tid);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
    const int g = col >> 3;
    const int n = (int)__umulhi((unsigned)g, a.ppi_magic);
    const int p = (g - n * a.ppi) * 8 + (col & 7);
    if (n >= a.n_img || p >= HW) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) +
                        4 * (lane >> 5);
        if (row >= M) continue;
        const int64_t at = ((int64_t)n * M + row) * HW + p;
        float v = acc[i][j][r];
        if (addend) v += to_f<T>(addend[at]);
        out[at] = from_f<T>(v);
      }
  }
}

// Filter re-layout for conv_flat_h16_kernel with `taps` taps (1 or 9).
// mode 0 (forward): w is [M][Kc][taps], wt[tap][m][k] = w[m][k][tap].  mode 1
// (backward-data): w is the forward filter [Kc][M][taps] seen from dY's side,
// wt[tap][m][k] = w[k][m][taps - 1 - tap].
template <typename T>
__global__ void conv_flat_prep_kernel(const T *__restrict__ w,
                                      T *__restrict__ wt, int M, int Kc,
                                      int taps, int mode) {
  const int64_t total = (int64_t)taps * M * Kc;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += step) {
    const int k = (int)(e % Kc);
    const int m = (int)((e / Kc) % M);
    const int tap = (int)(e / ((int64_t)Kc * M));
    wt[e] = mode == 0 ? w[((int64_t)m * Kc + k) * taps + tap]
                      : w[((int64_t)k * M + m) * taps + (taps - 1 - tap)];
  }
}

template <typename T, int WVM, int WVN, int TM, int TN, int DEPTH>
static void launch_conv_flat(const ConvFlatArgs &a, bool varpar,
                             hipStream_t s) {
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  const int ntm = (a.M + BM - 1) / BM;
  const int ntn = (a.n_img * a.ppi * 8 + BN - 1) / BN;
  const dim3 grid((unsigned)(((int64_t)ntm * ntn + 7) / 8 * 8));
  if (varpar)
    hipLaunchKernelGGL(
        (conv_flat_h16_kernel<T, WVM, WVN, TM, TN, DEPTH, true>), grid,
        dim3(64 * WVM * WVN), 0, s, a, ntm, ntn);
  else
    hipLaunchKernelGGL(
        (conv_flat_h16_kernel<T, WVM, WVN, TM, TN, DEPTH, false>), grid,
        dim3(64 * WVM * WVN), 0, s, a, ntm, ntn);
}

// ---------------------------------------------------------------------------
// 16-bit weight gradient of 1x1 and 3x3 / stride 1 / same-size convolutions
// ---------------------------------------------------------------------------
// dW[m][c][tap] = sum over images n and positions p of
//                 dy[n][m][p] * x[n][c][p shifted by the tap]
// Per tap this is a GEMM whose reduction index (the position) is contiguous in
// BOTH operands, the layout the tiled 16-bit kernel reads fastest (16-byte
// pieces into a row-major LDS image, ds_read_b128 fragments).  The reduction
// runs over the flattened list of 8-position pieces of all images (`ppi`
// pieces per image, so images whose size is not a multiple of 64 positions
// waste no matrix work); a reduction tile is 8 consecutive pieces.  The x
// pieces of a shifted tap are fetched at the shifted byte offset (odd shifts
// as five aligned dwords funnel-shifted into four) and cleared where the tap
// leaves the image.  The reduction is cut into `splits` ranges; a workgroup
// (tile, tap, split) writes its float partial product to
// part[split][tap][M][Nc], summed and cast by wgrad_reduce_kernel.
struct WgradArgs {
  const void *dy;  // [n][M][HW]
  const void *x;   // [n][Nc][HW]
  float *part;     // [splits][taps][M][Nc]
  int M, Nc, H, W, n_img;
  int ppi;         // pieces of 8 positions per image
  int tiles;       // reduction tiles in total
  int per_split;   // reduction tiles per split
  int taps;        // 1 or 9
  int splits;
  unsigned ppi_magic, w_magic;  // ceil(2^32 / d): umulhi(v, magic) == v / d
};

template <typename T, int WVM, int WVN, int TM, int TN, int DEPTH, bool VARPAR>
__global__ void __launch_bounds__(64 * WVM * WVN)
conv_wgrad_h16_kernel(WgradArgs a, int ntm, int ntn) {
  using namespace h16;
  using frag = typename Half16<T>::frag;
  constexpr int NT = 64 * WVM * WVN;
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  using OpA = Operand<true, BM, NT>;
  using OpB = Operand<true, BN, NT>;
  constexpr int PERA = OpA::PER, PERB = OpB::PER;
  constexpr int kImgA = BM * BK * 2, kImgB = BN * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2][kImgA + kImgB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WVN, wn = wid % WVN;
  // Workgroup ids go round-robin over the 8 XCDs: XCD x takes the x-th eighth
  // of the work list, which is ordered split-major, so the workgroups that
  // read the same positions (all tiles and taps of one split) share an L2.
  int pid_m, pid_n, tap, split;
  {
    const int per_split = ntm * ntn * a.taps;
    const int w = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (w >= per_split * a.splits) return;
    split = w / per_split;
    const int in = w - split * per_split;
    tap = in / (ntm * ntn);
    const int tile = in - tap * (ntm * ntn);
    pid_m = tile % ntm;
    pid_n = tile / ntm;
  }
  const int m0 = pid_m * BM, n0 = pid_n * BN;
  const int H = a.H, W = a.W, HW = H * W, M = a.M, Nc = a.Nc;
  const int dr = a.taps == 9 ? tap / 3 - 1 : 0;
  const int ds = a.taps == 9 ? tap % 3 - 1 : 0;
  const int shift = dr * W + ds;

  const int limit_a = a.n_img * M * HW * 2, limit_b = a.n_img * Nc * HW * 2;
  const rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(a.dy), 0, limit_a, 0x00020000);
  const rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(a.x), 0, limit_b, 0x00020000);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DEPTH reduction tiles are in flight in registers: the loads of tile
  // t + DEPTH are issued while tile t is multiplied, so a workgroup keeps
  // streaming although one load takes about a microsecond
  u32x4 ra[DEPTH][PERA], rb[DEPTH][PERB];
  const int ch = tid & 7, r_first = tid >> 3;

  auto expand = [](unsigned bits, unsigned *keep) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
      keep[d] = (((bits >> (2 * d)) & 1u) ? 0x0000ffffu : 0u) |
                (((bits >> (2 * d + 1)) & 1u) ? 0xffff0000u : 0u);
  };
  // One dword, range-checked on its own; offsets are redirected one by one so
  // that the loads cannot be fused back into a wide load, which is
  // range-checked as a whole.
  auto dw = [](const rsrc_t &rs, int o, int limit) {
    const int at = (o >= 0 && o + 4 <= limit) ? o : kOobOff;
    return __builtin_bit_cast(
        unsigned, __builtin_amdgcn_raw_buffer_load_b32(rs, at, 0, 0));
  };

  // Loads of reduction tile t into one register stage.  Everything is
  // select-based (an unwanted piece is fetched at the out-of-range offset and
  // reads zero): divergent branches around the loads would make the wave
  // wait for each piece in turn.  EDGE is the variant for the few tiles at
  // the two ends of the tensors, where a shifted or overhanging piece may
  // start before the buffer or end behind it: it fetches dword by dword.
  auto fetch = [&](int t, u32x4 *ra, u32x4 *rb, auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    // this thread's piece of the reduction tile: image n, positions p .. p + 7
    const int g = t * 8 + ch;
    const int n = (int)__umulhi((unsigned)g, a.ppi_magic);
    const int p = (g - n * a.ppi) * 8;
    const bool live = n < a.n_img;
    const int tail = HW - p;  // valid positions in the piece when < 8
    unsigned keep_a[4], keep_b[4];
    const unsigned tailbits = tail >= 8 ? 0xffu : ((1u << tail) - 1u);
    unsigned bbits = tailbits;
    if (a.taps == 9) {
      const int h = (int)__umulhi((unsigned)p, a.w_magic);
      const int w0 = p - h * W;
      const int e_wrap = W - w0;  // first element on row h + 1
      const unsigned first = e_wrap >= 8 ? 0xffu : ((1u << e_wrap) - 1u);
      const unsigned second = 0xffu & ~first;
      const bool v1 = h < H && (unsigned)(h + dr) < (unsigned)H;
      const bool v2 = h + 1 < H && (unsigned)(h + 1 + dr) < (unsigned)H;
      bbits = (v1 ? first : 0u) | (v2 ? second : 0u);
      if (ds < 0)
        bbits &= ~((w0 == 0 ? 1u : 0u) | (e_wrap < 8 ? (1u << e_wrap) : 0u));
      if (ds > 0)
        bbits &= ~((e_wrap <= 8 ? (1u << (e_wrap - 1)) : 0u) |
                   (e_wrap + W - 1 < 8 ? (1u << (e_wrap + W - 1)) : 0u));
    }
    if (!live) bbits = 0u;
    expand(tailbits, keep_a);
    expand(bbits, keep_b);
    // one 8-element piece starting at element offset e0 (any parity: odd
    // starts are fetched as five aligned dwords and funnel-shifted; `odd` is
    // per thread with VARPAR, else the same for the whole tile)
    auto piece = [&](const rsrc_t &rs, int limit, int e0, bool want, bool odd,
                     bool five) {
      const int b0 = (e0 & ~1) * 2;
      u32x4 lo;
      unsigned hi = 0;
      if (EDGE) {
        lo.x = dw(rs, want ? b0 : kOobOff, limit);
        lo.y = dw(rs, want ? b0 + 4 : kOobOff, limit);
        lo.z = dw(rs, want ? b0 + 8 : kOobOff, limit);
        lo.w = dw(rs, want ? b0 + 12 : kOobOff, limit);
        if (five) hi = dw(rs, want ? b0 + 16 : kOobOff, limit);
      } else {
        lo = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                       rs, want ? b0 : kOobOff, 0, 0));
        if (five)
          hi = __builtin_bit_cast(
              unsigned, __builtin_amdgcn_raw_buffer_load_b32(
                            rs, want ? b0 + 16 : kOobOff, 0, 0));
      }
      u32x4 v = lo;
      if (five) {
        u32x4 sh;
        sh.x = __builtin_amdgcn_alignbit(lo.y, lo.x, 16);
        sh.y = __builtin_amdgcn_alignbit(lo.z, lo.y, 16);
        sh.z = __builtin_amdgcn_alignbit(lo.w, lo.z, 16);
        sh.w = __builtin_amdgcn_alignbit(hi, lo.w, 16);
        v.x = odd ? sh.x : lo.x; v.y = odd ? sh.y : lo.y;
        v.z = odd ? sh.z : lo.z; v.w = odd ? sh.w : lo.w;
      }
      return v;
    };
    const bool tap_odd = (shift & 1) != 0;
#pragma unroll
    for (int j = 0; j < PERA; ++j) {
      const int r = m0 + r_first + (NT / 8) * j;
      const int e0 = (n * M + r) * HW + p;
      u32x4 v = piece(rsrc_a, limit_a, e0, live && r < M,
                      VARPAR && (e0 & 1) != 0, VARPAR);
      v.x &= keep_a[0]; v.y &= keep_a[1]; v.z &= keep_a[2]; v.w &= keep_a[3];
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < PERB; ++j) {
      const int r = n0 + r_first + (NT / 8) * j;
      const int e0 = (n * Nc + r) * HW + p + shift;
      u32x4 v = piece(rsrc_b, limit_b, e0, bbits != 0u && r < Nc,
                      VARPAR ? (e0 & 1) != 0 : tap_odd, VARPAR || tap_odd);
      v.x &= keep_b[0]; v.y &= keep_b[1]; v.z &= keep_b[2]; v.w &= keep_b[3];
      rb[j] = v;
    }
  };
  // tiles whose pieces may reach W + 1 positions past an end of the tensors
  const int edge_lo = ((W + 23) / 8 + 7) / 8;                  // tiles [0, lo)
  const int edge_hi = (a.n_img * a.ppi - (W + 24) / 8 - 1) / 8;  // [hi, ...)
  auto fetch_tile = [&](int t, u32x4 *ra, u32x4 *rb) {
    if (t < edge_lo || t >= edge_hi) fetch(t, ra, rb, std::true_type{});
    else fetch(t, ra, rb, std::false_type{});
  };

  const int t0 = split * a.per_split;
  const int t1 = min(a.tiles, t0 + a.per_split);
#pragma unroll
  for (int u = 0; u < DEPTH; ++u)
    if (t0 + u < t1) fetch_tile(t0 + u, ra[u], rb[u]);
  if (t0 < t1) {
    OpA::store(tid, smem[0], ra[0]);
    OpB::store(tid, smem[0] + kImgA, rb[0]);
  }
  __syncthreads();
  int cur = 0;
  for (int t = t0; t < t1; t += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int tt = t + u;
      if (tt >= t1) break;
      // tile tt sits in LDS; its registers (slot u) take tile tt + DEPTH
      const char *as = smem[cur], *bs = smem[cur] + kImgA;
      if (tt + DEPTH < t1) fetch_tile(tt + DEPTH, ra[u], rb[u]);
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        frag fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = __builtin_bit_cast(
              frag, OpA::frag(as, (wm * TM + i) * 32, ks, lane));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = __builtin_bit_cast(
              frag, OpB::frag(bs, (wn * TN + j) * 32, ks, lane));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = Half16<T>::mma(fa[i], fb[j], acc[i][j]);
      }
      if (tt + 1 < t1) {
        // the other buffer was last read before the previous barrier
        OpA::store(tid, smem[cur ^ 1], ra[(u + 1) % DEPTH]);
        OpB::store(tid, smem[cur ^ 1] + kImgA, rb[(u + 1) % DEPTH]);
      }
      __syncthreads();
      cur ^= 1;
    }
  }

  float *part = a.part + ((int64_t)split * a.taps + tap) * M * Nc;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
      if (col >= Nc) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) +
                        4 * (lane >> 5);
        if (row >= M) continue;
        part[(int64_t)row * Nc + col] = acc[i][j][r];
      }
    }
}

// dw[m][c][tap] = sum over splits of part[split][tap][m][c], in dw's type.
// A block sums 32 consecutive elements; its 8 groups of 32 threads each take
// every eighth split (coalesced 128-byte reads) and meet in LDS.
template <typename T>
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float *__restrict__ part, int splits, int taps,
                    int64_t mn, T *__restrict__ dw) {
  __shared__ float sums[8][33];
  const int64_t total = mn * taps;
  const int ii = threadIdx.x & 31, si = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + ii;
  float v = 0.f;
  if (i < total)
    for (int sp = si; sp < splits; sp += 8) v += part[sp * total + i];
  sums[si][ii] = v;
  __syncthreads();
  if (si == 0 && i < total) {
#pragma unroll
    for (int k = 1; k < 8; ++k) v += sums[k][ii];
    const int64_t tap = i / mn, e = i - tap * mn;
    dw[e * taps + tap] = from_f<T>(v);
  }
}

template <typename T, int WVM, int WVN, int TM, int TN, int DEPTH>
static void launch_wgrad_h16(const WgradArgs &a, hipStream_t s) {
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.Nc + BN - 1) / BN;
  const int64_t wgs = (int64_t)ntm * ntn * a.taps * a.splits;
  const dim3 grid((unsigned)((wgs + 7) / 8 * 8)), block(64 * WVM * WVN);
  // an odd image size puts the rows of both operands at either parity
  if ((a.H * a.W) % 2 != 0)
    hipLaunchKernelGGL(
        (conv_wgrad_h16_kernel<T, WVM, WVN, TM, TN, DEPTH, true>), grid, block,
        0, s, a, ntm, ntn);
  else
    hipLaunchKernelGGL(
        (conv_wgrad_h16_kernel<T, WVM, WVN, TM, TN, DEPTH, false>), grid,
        block, 0, s, a, ntm, ntn);
}

// Tile of the weight-gradient kernel for an M x Nc filter matrix: rows, columns
static void wgrad_tile(int M, int Nc, int *bm, int *bn) {
  *bm = M <= 64 ? 64 : 128;
  *bn = Nc <= 64 ? 64 : 128;
}

template <int BM, int BN>
void launch_f32(const GemmArgs &g, int amode, int bmode, dim3 grid,
                hipStream_t s) {
#define SA_GEMM_CASE(AM, BMo)                                               \
  if (amode == AM && bmode == BMo) {                                        \
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AM, BMo>), grid, dim3(256), \
                       0, s, g);                                            \
    return;                                                                 \
  }
  SA_GEMM_CASE(LD_VEC_K, LD_VEC_MN)
  SA_GEMM_CASE(LD_VEC_K, LD_VEC_K)
  SA_GEMM_CASE(LD_VEC_MN, LD_VEC_MN)
  SA_GEMM_CASE(LD_VEC_MN, LD_VEC_K)
  SA_GEMM_CASE(LD_GENERIC, LD_GENERIC)
#undef SA_GEMM_CASE
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, LD_GENERIC, LD_GENERIC>), grid,
                     dim3(256), 0, s, g);
}

// Tile shapes of gemm_f32_tiles_kernel (indices 0..6) and, as index 7, the
// single-stage kernel with 128 x 128 tiles.  `rate`: speed of one CU working
// on tiles of that shape relative to the best one, from tools/
// bench_gemm_tiles.py on large products (the single-stage kernel keeps more
// workgroups resident and leads there; the 256-wide tiles leave one wave per
// SIMD and never won).
struct F32Tile {
  int bm, bn;
  float rate;
};
static const F32Tile kF32Tiles[] = {
    {64, 64, 0.84f},   {128, 64, 0.87f},  {64, 128, 0.92f}, {128, 128, 0.94f},
    {256, 128, 0.80f}, {256, 96, 0.85f},  {128, 256, 0.82f}, {128, 128, 1.0f}};
constexpr int kNumF32Tiles = sizeof(kF32Tiles) / sizeof(kF32Tiles[0]);
constexpr int kF32SingleStage = 7;

static int &f32_tile_override() {
  static int v = [] {
    const char *e = getenv("SA_GEMM_F32_TILE");
    return e && *e ? atoi(e) : -1;
  }();
  return v;
}

// The shape whose busiest CU finishes first: a CU hosts ceil(tiles / 256)
// tiles of bm x bn outputs at the shape's rate; edge tiles count whole.
static int f32_tile_for(int64_t M, int64_t N, int64_t batch) {
  const int forced = f32_tile_override();
  if (forced >= 0 && forced < kNumF32Tiles) return forced;
  int best = 0;
  double best_cost = 1e300;
  for (int i : {0, 1, 2, 3, kF32SingleStage}) {
    const F32Tile &t = kF32Tiles[i];
    if (i == kF32SingleStage && (M < 96 || N < 96)) continue;
    const int64_t tiles = ((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn) *
                          batch;
    const double per_cu = (double)((tiles + 255) / 256);
    const double cost = per_cu * t.bm * t.bn / t.rate;
    if (cost < best_cost * 0.999) {
      best_cost = cost;
      best = i;
    }
  }
  return best;
}

template <int WVM, int WVN, int TM, int TN>
static void launch_f32_tiles(const GemmArgs &g, int amode, int bmode,
                             int64_t batch, hipStream_t s) {
  constexpr int BM = 32 * TM * WVM, BN = 32 * TN * WVN;
  const dim3 grid(
      (unsigned)((((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM) + 7) / 8 * 8), 1,
      (unsigned)batch);
#define SA_GEMM_CASE(AM, BMo)                                                 \
  if (amode == AM && bmode == BMo) {                                          \
    hipLaunchKernelGGL(                                                       \
        (gemm_f32_tiles_kernel<WVM, WVN, TM, TN, AM, BMo>), grid, dim3(256), \
        0, s, g);                                                             \
    return;                                                                   \
  }
  SA_GEMM_CASE(LD_VEC_K, LD_VEC_MN)
  SA_GEMM_CASE(LD_VEC_K, LD_VEC_K)
  SA_GEMM_CASE(LD_VEC_MN, LD_VEC_MN)
  SA_GEMM_CASE(LD_VEC_MN, LD_VEC_K)
#undef SA_GEMM_CASE
  hipLaunchKernelGGL(
      (gemm_f32_tiles_kernel<WVM, WVN, TM, TN, LD_GENERIC, LD_GENERIC>), grid,
      dim3(256), 0, s, g);
}

static void launch_f32_auto(const GemmArgs &g, int amode, int bmode,
                            int64_t batch, hipStream_t s) {
  switch (f32_tile_for(g.M, g.N, batch)) {
    case 0: return launch_f32_tiles<2, 2, 1, 1>(g, amode, bmode, batch, s);
    case 1: return launch_f32_tiles<2, 2, 2, 1>(g, amode, bmode, batch, s);
    case 2: return launch_f32_tiles<2, 2, 1, 2>(g, amode, bmode, batch, s);
    case 3: return launch_f32_tiles<2, 2, 2, 2>(g, amode, bmode, batch, s);
    case 4: return launch_f32_tiles<2, 2, 4, 2>(g, amode, bmode, batch, s);
    case 5: return launch_f32_tiles<4, 1, 2, 3>(g, amode, bmode, batch, s);
    case 6: return launch_f32_tiles<2, 2, 2, 4>(g, amode, bmode, batch, s);
    default: {
      const dim3 grid(
          (unsigned)((((g.N + 127) / 128) * ((g.M + 127) / 128) + 7) / 8 * 8),
          1, (unsigned)batch);
      return launch_f32<128, 128>(g, amode, bmode, grid, s);
    }
  }
}

// Split-K epilogue: C = act(alpha * sum_s part[s] + beta * C + bias), part is
// [splits][M][N] float; dtype is C's and the bias's element type.
template <typename T>
__global__ void splitk_reduce_kernel(const float *__restrict__ part, int splits,
                                     int64_t M, int64_t N, float alpha,
                                     float beta, T *__restrict__ C,
                                     int64_t ldc, const T *__restrict__ bias,
                                     int act) {
  const int64_t total = M * N;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    float v = 0.f;
    for (int sp = 0; sp < splits; ++sp) v += part[sp * total + i];
    const int64_t r = i / N, c = i - r * N;
    v *= alpha;
    if (bias) v += to_f<T>(bias[c]);
    T *cp = C + r * ldc + c;
    if (beta != 0.f) v += beta * to_f<T>(*cp);
    if (act == 1) v = fmaxf(v, 0.f);
    *cp = from_f<T>(v);
  }
}

}  // namespace sa

using namespace sa;

int sa_gemm_splitk_reduce(int dtype, const float *part, int splits, int64_t M,
                          int64_t N, float alpha, float beta, void *C,
                          int64_t ldc, const void *bias, int act,
                          hipStream_t s) {
  if (M <= 0 || N <= 0 || splits <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((splitk_reduce_kernel<T>),
                       dim3(grid_for(M * N, 256)), dim3(256), 0, s, part,
                       splits, M, N, alpha, beta, (T *)C, ldc, (const T *)bias,
                       act);
  });
  return 0;
}

// The tiled 16-bit kernel needs both operands readable in aligned 16-byte
// pieces along their contiguous index and every byte offset within 31 bits.
static bool h16_fast_ok(const GemmArgs &g) {
  auto al16 = [](const void *p) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  };
  const bool a_kc = g.a_cs == 1, b_kc = g.b_rs == 1;
  // a 16-byte buffer load needs a 4-byte aligned address: two elements
  const bool a_ok = al16(g.A) && g.a_bs % 2 == 0 &&
                    (a_kc ? g.a_rs % 2 == 0
                          : (g.a_rs == 1 && g.a_cs % 2 == 0));
  const bool b_ok = al16(g.B) && g.b_bs % 2 == 0 &&
                    (b_kc ? g.b_cs % 2 == 0
                          : (g.b_cs == 1 && g.b_rs % 2 == 0));
  const int64_t lim = ((int64_t)1 << 30) - 1024;  // elements
  const int64_t a_ext = (g.M - 1) * g.a_rs + (g.K - 1) * g.a_cs;
  const int64_t b_ext = (g.K - 1) * g.b_rs + (g.N - 1) * g.b_cs;
  return a_ok && b_ok && g.K % 8 == 0 && g.K >= 64 && g.M >= 8 && g.N >= 8 &&
         a_ext >= 0 && b_ext >= 0 && a_ext + 256 * g.a_rs < lim &&
         b_ext + 256 * g.b_cs < lim && g.M < lim && g.N < lim;
}

int sa_gemm(int dtype, int64_t M, int64_t N, int64_t K, float alpha,
            const void *A, int64_t a_rs, int64_t a_cs, int64_t a_bs,
            const void *B, int64_t b_rs, int64_t b_cs, int64_t b_bs,
            float beta, void *C, int64_t ldc, int64_t c_bs, int64_t batch,
            const void *bias, int act, hipStream_t s) {
  if (M <= 0 || N <= 0 || batch <= 0) return 0;
  GemmArgs g{M, N, K, alpha, beta, A, a_rs, a_cs, a_bs, B, b_rs, b_cs, b_bs,
             C, ldc, c_bs, bias, act};
  // big tiles only when they still fill the 256 CUs
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128) * batch;
  const bool big = tiles128 >= 192 && M >= 96 && N >= 96;
  if (dtype == kF32) {
    auto al16 = [](const void *p) {
      return (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    };
    int amode = LD_GENERIC, bmode = LD_GENERIC;
    if (a_cs == 1 && (a_rs % 4) == 0 && (K % 4) == 0 && (a_bs % 4) == 0 &&
        al16(A))
      amode = LD_VEC_K;
    else if (a_rs == 1 && (a_cs % 4) == 0 && (M % 4) == 0 && (a_bs % 4) == 0 &&
             al16(A))
      amode = LD_VEC_MN;
    if (b_cs == 1 && (b_rs % 4) == 0 && (N % 4) == 0 && (b_bs % 4) == 0 &&
        al16(B))
      bmode = LD_VEC_MN;
    else if (b_rs == 1 && (b_cs % 4) == 0 && (K % 4) == 0 && (b_bs % 4) == 0 &&
             al16(B))
      bmode = LD_VEC_K;
    if (amode == LD_GENERIC || bmode == LD_GENERIC)
      amode = bmode = LD_GENERIC;
    if (f32_tile_override() != -2) {
      launch_f32_auto(g, amode, bmode, batch, s);
      return 0;
    }
    // SA_GEMM_F32_TILE=-2: the single-stage kernel with 128 / 64 tiles
    if (big) {
      dim3 grid((unsigned)((((N + 127) / 128) * ((M + 127) / 128) + 7) / 8 * 8),
                1, (unsigned)batch);
      launch_f32<128, 128>(g, amode, bmode, grid, s);
    } else {
      dim3 grid((unsigned)((((N + 63) / 64) * ((M + 63) / 64) + 7) / 8 * 8), 1,
                (unsigned)batch);
      launch_f32<64, 64>(g, amode, bmode, grid, s);
    }
    return 0;
  }
  if (dtype == kF16 || dtype == kBF16) {
    if (h16_fast_ok(g)) {
      const bool a_kc = a_cs == 1, b_kc = b_rs == 1;
      if (dtype == kF16) launch_h16_fast<__half>(g, a_kc, b_kc, batch, s);
      else launch_h16_fast<__hip_bfloat16>(g, a_kc, b_kc, batch, s);
      return 0;
    }
    if (big) {
      dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128),
                (unsigned)batch);
      if (dtype == kF16)
        hipLaunchKernelGGL((gemm_h16_kernel<__half, 128, 128>), grid,
                           dim3(256), 0, s, g);
      else
        hipLaunchKernelGGL((gemm_h16_kernel<__hip_bfloat16, 128, 128>), grid,
                           dim3(256), 0, s, g);
    } else {
      dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64),
                (unsigned)batch);
      if (dtype == kF16)
        hipLaunchKernelGGL((gemm_h16_kernel<__half, 64, 64>), grid, dim3(256),
                           0, s, g);
      else
        hipLaunchKernelGGL((gemm_h16_kernel<__hip_bfloat16, 64, 64>), grid,
                           dim3(256), 0, s, g);
    }
    return 0;
  }
  return -1;
}

// 16-bit operands, float output: part[b] = A[b] * B[b] for b < batch (the
// slices of a split reduction).  Returns -3 when the operand layout does not
// fit the tiled kernel (the caller then takes the unsplit path).
// float32 product cut into `splits` slices of `kc` reduction indices (the
// last one holds what is left of `k_total`): slice z writes its partial
// product to part + z * M * N.  Any k_total; kc a multiple of 4.
int sa_gemm_f32_slices(int64_t M, int64_t N, int64_t kc, int64_t k_total,
                       const void *A, int64_t a_rs, int64_t a_cs,
                       const void *B, int64_t b_rs, int64_t b_cs, float *part,
                       int64_t splits, hipStream_t s) {
  if (M <= 0 || N <= 0 || splits <= 0) return 0;
  if (kc <= 0 || kc % 4 != 0 || (splits - 1) * kc >= k_total) return -2;
  GemmArgs g{M, N, kc, 1.f, 0.f, A, a_rs, a_cs, kc * a_cs, B, b_rs, b_cs,
             kc * b_rs, part, N, M * N, nullptr, 0};
  g.k_total = k_total;
  auto al16 = [](const void *p) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  };
  // vector loads along k need every slice to end on a group of four
  int amode = LD_GENERIC, bmode = LD_GENERIC;
  if (a_cs == 1 && (a_rs % 4) == 0 && (k_total % 4) == 0 && al16(A))
    amode = LD_VEC_K;
  else if (a_rs == 1 && (a_cs % 4) == 0 && (M % 4) == 0 && al16(A))
    amode = LD_VEC_MN;
  if (b_cs == 1 && (b_rs % 4) == 0 && (N % 4) == 0 && al16(B))
    bmode = LD_VEC_MN;
  else if (b_rs == 1 && (b_cs % 4) == 0 && (k_total % 4) == 0 && al16(B))
    bmode = LD_VEC_K;
  if (amode == LD_GENERIC || bmode == LD_GENERIC)
    amode = bmode = LD_GENERIC;
  if (f32_tile_override() != -2) {
    launch_f32_auto(g, amode, bmode, splits, s);
    return 0;
  }
  dim3 grid((unsigned)((((N + 63) / 64) * ((M + 63) / 64) + 7) / 8 * 8), 1,
            (unsigned)splits);
  launch_f32<64, 64>(g, amode, bmode, grid, s);
  return 0;
}

void sa_set_gemm_h16_tile(int tile) { h16_tile_override() = tile; }
void sa_set_gemm_f32_tile(int tile) { f32_tile_override() = tile; }
int sa_get_gemm_f32_tile() { return f32_tile_override(); }

int sa_gemm_h16_partial(int dtype, int64_t M, int64_t N, int64_t K,
                        const void *A, int64_t a_rs, int64_t a_cs,
                        int64_t a_bs, const void *B, int64_t b_rs,
                        int64_t b_cs, int64_t b_bs, float *part, int64_t batch,
                        int64_t k_total, hipStream_t s) {
  if (dtype != kF16 && dtype != kBF16) return -1;
  if (M <= 0 || N <= 0 || batch <= 0) return 0;
  GemmArgs g{M, N, K, 1.f, 0.f, A, a_rs, a_cs, a_bs, B, b_rs, b_cs, b_bs,
             part, N, M * N, nullptr, 0};
  g.c_float = 1;
  if (!h16_fast_ok(g)) return -3;
  // unequal slices: K indices each, the last one what is left of k_total
  if (k_total > 0 && (k_total % 8 != 0 || (batch - 1) * K >= k_total))
    return -3;
  g.k_total = k_total;
  const bool a_kc = a_cs == 1, b_kc = b_rs == 1;
  if (dtype == kF16) launch_h16_fast<__half>(g, a_kc, b_kc, batch, s);
  else launch_h16_fast<__hip_bfloat16>(g, a_kc, b_kc, batch, s);
  return 0;
}

// 16-bit 1x1 convolution with stride 1 and no padding as one GEMM per image,
// on the tiled 16-bit kernel (16-byte staging, 16-bit LDS images, transposing
// LDS reads for the pixel-contiguous activations):
//   mode 0  y[n]  (K x HW) = w (K x C) * x[n] (C x HW)
//   mode 1  dx[n] (C x HW) = w^T (C x K) * dy[n] (K x HW)  [+ addend[n]]
// Returns -3 when the geometry does not fit (caller uses the implicit-GEMM
// convolution kernel).
int sa_conv1x1_h16(int dtype, int mode, int64_t n_img, int64_t C, int64_t K,
                   int64_t HW, const void *act, const void *w,
                   const void *addend, void *out, hipStream_t s) {
  if (dtype != kF16 && dtype != kBF16) return -1;
  if (n_img <= 0) return 0;
  GemmArgs g{};
  g.alpha = 1.f;
  g.beta = 0.f;
  g.bias = nullptr;
  g.act = 0;
  g.c_float = 0;
  g.addend = addend;
  g.B = act;
  g.b_cs = 1;
  g.b_rs = HW;
  g.C = out;
  g.ldc = HW;
  g.N = HW;
  g.A = w;
  g.a_bs = 0;
  if (mode == 0) {
    g.M = K; g.K = C;
    g.a_rs = C; g.a_cs = 1;
    g.b_bs = C * HW; g.c_bs = K * HW;
  } else {
    g.M = C; g.K = K;
    g.a_rs = 1; g.a_cs = C;
    g.b_bs = K * HW; g.c_bs = C * HW;
  }
  if (!h16_fast_ok(g)) return -3;
  const bool a_kc = g.a_cs == 1;
  if (dtype == kF16) launch_h16_auto<__half>(g, a_kc, false, n_img, s);
  else launch_h16_auto<__hip_bfloat16>(g, a_kc, false, n_img, s);
  return 0;
}

// 16-bit 3x3 / stride 1 / pad 1 convolution, forward (mode 0) and
// backward-data (mode 1), as a nine-tap GEMM per image.  `wt` is workspace of
// 9 * C * K elements for the re-laid-out filter.  Returns -3 when the geometry
// does not fit (caller uses the implicit-GEMM convolution kernel).
int sa_conv3x3_h16(int dtype, int mode, int n_img, int C, int K, int H, int W,
                   const void *act, const void *w, void *wt,
                   const void *addend, void *out, hipStream_t s) {
  if (dtype != kF16 && dtype != kBF16) return -1;
  if (n_img <= 0) return 0;
  const int M = mode == 0 ? K : C, Kc = mode == 0 ? C : K;
  auto al16 = [](const void *p) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  };
  const int64_t HW = (int64_t)H * W;
  if (Kc % 64 != 0 || M < 8 || W % 2 != 0 || H < 2 || W < 4 ||
      !al16(act) || !al16(out) || !al16(wt) || !al16(addend) ||
      (int64_t)std::max(M, Kc) * HW >= ((int64_t)1 << 29) ||
      (int64_t)9 * M * Kc >= ((int64_t)1 << 29))
    return -3;
  ConvTapArgs a{wt, act, addend, out, M, Kc, H, W};
  const int64_t total = (int64_t)9 * M * Kc;
  SA_DISPATCH_FLOAT(dtype, T, {
    if constexpr (!std::is_same<T, float>::value) {
      hipLaunchKernelGGL((conv3x3_h16_prep_kernel<T>),
                         dim3(grid_for(total, 256)), dim3(256), 0, s,
                         (const T *)w, (T *)wt, M, Kc, mode);
      if (M <= 64) launch_conv3x3_h16<T, 1, 4, 2, 1>(a, n_img, s);
      else launch_conv3x3_h16<T, 2, 2, 2, 2>(a, n_img, s);
    }
  });
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Split count of the 16-bit weight-gradient kernel for a K x C filter with
// `taps` taps over n_img images of H x W positions; `part` must hold
// splits * taps * K * C floats.  0 when the geometry does not fit the kernel
// (the caller then uses the implicit-GEMM convolution kernel).
int sa_conv_wgrad_h16_splits(int n_img, int C, int K, int H, int W, int taps) {
  const int64_t HW = (int64_t)H * W;
  // a piece of 8 positions may span two image rows, not three
  if ((taps != 1 && taps != 9) || n_img <= 0 || C < 8 || K < 8 || W < 7 ||
      HW < 49 ||
      (int64_t)n_img * std::max(C, K) * HW >= ((int64_t)1 << 29))
    return 0;
  static const int forced = [] {
    const char *v = getenv("SA_WGRAD_SPLITS");
    return v && *v ? atoi(v) : 0;
  }();
  int bm, bn;
  wgrad_tile(K, C, &bm, &bn);
  const int64_t base =
      (int64_t)((K + bm - 1) / bm) * ((C + bn - 1) / bn) * taps;
  const int64_t ppi = (HW + 7) / 8;
  const int64_t tiles = (n_img * ppi + 7) / 8;
  // Measured on ResNet-50's layers: about 576 workgroups in all is the best
  // trade between filling the CUs and the traffic of the partial products
  // (each split writes and re-reads a whole filter gradient); at least four
  // reduction tiles per workgroup, at most 24 MB of partial products.
  int64_t splits = std::max<int64_t>(1, 576 / base);
  splits = std::min(splits, std::max<int64_t>(1, tiles / 4));
  splits = std::min(
      splits, std::max<int64_t>(1, ((int64_t)6 << 20) / ((int64_t)taps * K * C)));
  if (forced > 0) splits = std::min<int64_t>(forced, tiles);
  const int64_t per = (tiles + splits - 1) / splits;
  return (int)((tiles + per - 1) / per);
}

// dw (K x C x taps, 16-bit) = weight gradient of a 1x1 (taps 1) or 3x3 / pad 1
// (taps 9) stride-1 convolution from dy [n][K][H][W] and x [n][C][H][W].
int sa_conv_wgrad_h16(int dtype, int n_img, int C, int K, int H, int W,
                      int taps, const void *dy, const void *x, float *part,
                      int splits, void *dw, hipStream_t s) {
  if (dtype != kF16 && dtype != kBF16) return -1;
  if (splits <= 0 ||
      splits != sa_conv_wgrad_h16_splits(n_img, C, K, H, W, taps))
    return -3;
  auto al16 = [](const void *p) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  };
  if (!al16(dy) || !al16(x) || !al16(part)) return -3;
  const int HW = H * W;
  WgradArgs a{};
  a.dy = dy; a.x = x; a.part = part;
  a.M = K; a.Nc = C; a.H = H; a.W = W; a.n_img = n_img;
  a.ppi = (HW + 7) / 8;
  a.tiles = (n_img * a.ppi + 7) / 8;
  a.per_split = (a.tiles + splits - 1) / splits;
  a.taps = taps;
  a.splits = splits;
  a.ppi_magic = (unsigned)((((uint64_t)1 << 32) + a.ppi - 1) / a.ppi);
  a.w_magic = (unsigned)((((uint64_t)1 << 32) + W - 1) / W);
  int bm, bn;
  wgrad_tile(K, C, &bm, &bn);
  SA_DISPATCH_FLOAT(dtype, T, {
    if constexpr (!std::is_same<T, float>::value) {
      if (bm == 64 && bn == 64) launch_wgrad_h16<T, 2, 2, 1, 1, 4>(a, s);
      else if (bm == 64) launch_wgrad_h16<T, 1, 4, 2, 1, 3>(a, s);
      else if (bn == 64) launch_wgrad_h16<T, 4, 1, 1, 2, 3>(a, s);
      else launch_wgrad_h16<T, 2, 2, 2, 2, 3>(a, s);
      hipLaunchKernelGGL((wgrad_reduce_kernel<T>),
                         dim3((unsigned)(((int64_t)taps * K * C + 31) / 32)),
                         dim3(256), 0, s, part, splits, taps, (int64_t)K * C,
                         (T *)dw);
    }
  });
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// 16-bit 1x1 (taps 1) or 3x3 / pad 1 (taps 9) stride-1 convolution over the
// flattened positions of all images, forward (mode 0) and backward-data
// (mode 1); any image size.  `wt` is workspace of taps * C * K elements for
// the re-laid-out filter (unused for a forward 1x1).  Returns -3 when the
// geometry does not fit (caller uses the implicit-GEMM convolution kernel).
int sa_conv_flat_h16(int dtype, int mode, int n_img, int C, int K, int H,
                     int W, int taps, const void *act, const void *w, void *wt,
                     const void *addend, void *out, hipStream_t s) {
  if (dtype != kF16 && dtype != kBF16) return -1;
  if (n_img <= 0) return 0;
  const int M = mode == 0 ? K : C, Kc = mode == 0 ? C : K;
  auto al16 = [](const void *p) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  };
  const int64_t HW = (int64_t)H * W;
  if ((taps != 1 && taps != 9) || Kc % 64 != 0 || M < 8 || H < 1 || W < 1 ||
      (taps == 9 && (H < 2 || W < 2)) || !al16(act) || !al16(out) ||
      !al16(wt) || !al16(w) || !al16(addend) ||
      (int64_t)n_img * std::max(M, Kc) * HW >= ((int64_t)1 << 29) ||
      (int64_t)taps * M * Kc >= ((int64_t)1 << 29))
    return -3;
  const bool direct = taps == 1 && mode == 0;  // w already is [M][Kc]
  ConvFlatArgs a{direct ? w : wt, act, addend, out, M, Kc, H, W, n_img, taps};
  a.ppi = (int)((HW + 7) / 8);
  a.ppi_magic = (unsigned)((((uint64_t)1 << 32) + a.ppi - 1) / a.ppi);
  const bool varpar = (HW % 2 != 0) || (W % 2 != 0);
  const int64_t cols = (int64_t)n_img * a.ppi * 8;
  auto wgs = [&](int bm, int bn) {
    return (int64_t)((M + bm - 1) / bm) * ((cols + bn - 1) / bn);
  };
  const int64_t total = (int64_t)taps * M * Kc;
  SA_DISPATCH_FLOAT(dtype, T, {
    if constexpr (!std::is_same<T, float>::value) {
      if (!direct)
        hipLaunchKernelGGL((conv_flat_prep_kernel<T>),
                           dim3(grid_for(total, 256)), dim3(256), 0, s,
                           (const T *)w, (T *)wt, M, Kc, taps, mode);
      // the largest tile that still gives every CU a workgroup and a bit
      if (M > 64 && wgs(128, 128) >= 320)
        launch_conv_flat<T, 2, 2, 2, 2, 3>(a, varpar, s);
      else if (M <= 64 && wgs(64, 128) >= 320)
        launch_conv_flat<T, 1, 4, 2, 1, 3>(a, varpar, s);
      else if (M > 64 && wgs(128, 64) >= 320)
        launch_conv_flat<T, 4, 1, 1, 2, 3>(a, varpar, s);
      else
        launch_conv_flat<T, 2, 2, 1, 1, 4>(a, varpar, s);
    }
  });
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
