// Fused scaled-dot-product attention for head dimension 64 on the CDNA4 matrix
// cores: forward (scores, additive mask, soft-max and the product with V in
// one kernel, the S x S score matrix never reaches memory) and backward
// (probabilities recomputed from the saved log-sum-exp).
//
//   O = softmax(scale * Q K^T + mask) V          per (batch, head)
//
// Q, K, V are read through (batch, head, row) element strides with the head
// dimension contiguous, so the (B, S, H, D) activations of a transformer are
// consumed as they are -- the (B, H, S, D) / (B, H, D, S) transposed views the
// graph describes are never materialised.  Outputs and gradients are written
// as contiguous (B, S, H, D).
//
// Layout of the computation (exact float32 v_mfma_f32_32x32x2_f32; 16-bit
// inputs are widened when staged).  The score tile is computed TRANSPOSED,
// S^T = K Q^T, so a lane owns one query column and its 16 accumulator
// registers are 16 keys: the soft-max reduction over keys runs in registers
// plus one exchange between the two lane halves.  A 32x32 accumulator tile X
// with the query on the lane is directly the B operand of a product that sums
// over its rows (keys): O^T += V^T X and dQ^T += K^T dS^T need no LDS round
// trip -- register r of the two lane halves supplies keys k_r and k_r + 4 of
// one 2-deep MFMA step, and the A operand reads the matching two rows of V
// (or K) from LDS.  Products that sum over the query (dV = P^T dO,
// dK = dS^T Q) pass P^T / dS^T through a per-wave LDS image once.
//
// The reference has no attention operator: its ONNX transformers run MatMul,
// Div, Add, Softmax, MatMul node by node (examples/onnx/bert/bert-squad.py).
#include "common.cuh"
#include "kernels.h"

namespace sa {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kHeadDim = 64;
constexpr int kPitch = 65;       // LDS row pitch (floats): column reads hit 32 banks
constexpr float kMasked = -1e30f;  // finite stand-in for minus infinity

struct AttnArgs {
  const void *q, *k, *v, *mask;
  int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss;
  int64_t m_sb, m_sh, m_sq, m_sk;
  void *out;           // (B, Sq, H, 64)
  float *lse;          // (B, H, Sq)
  const void *dout;    // (B, Sq, H, 64)
  float *delta;        // (B, H, Sq): rowsum(dO * O)
  void *dq, *dk, *dv;  // (B, S, H, 64)
  int B, H, Sq, Sk;
  float scale;
};

template <typename T> struct Ld4;
template <> struct Ld4<float> {
  static __device__ __forceinline__ float4 ld(const float *p) {
    return *reinterpret_cast<const float4 *>(p);
  }
};
template <> struct Ld4<__half> {
  static __device__ __forceinline__ float4 ld(const __half *p) {
    const uint2 w = *reinterpret_cast<const uint2 *>(p);
    __half e[4];
    __builtin_memcpy(e, &w, 8);
    return make_float4(__half2float(e[0]), __half2float(e[1]),
                       __half2float(e[2]), __half2float(e[3]));
  }
};
template <> struct Ld4<__hip_bfloat16> {
  static __device__ __forceinline__ float4 ld(const __hip_bfloat16 *p) {
    const uint2 w = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__builtin_bit_cast(float, w.x << 16),
                       __builtin_bit_cast(float, w.x & 0xffff0000u),
                       __builtin_bit_cast(float, w.y << 16),
                       __builtin_bit_cast(float, w.y & 0xffff0000u));
  }
};

// ROWS x 64 tile of a row-major operand -> float LDS image with pitch kPitch;
// rows at or beyond `limit` are zero
template <typename T, int ROWS>
__device__ __forceinline__ void stage_rows(float *dst, const T *base,
                                           int64_t row_stride, int first_row,
                                           int limit, int tid) {
#pragma unroll
  for (int e = tid; e < ROWS * 16; e += 256) {
    const int row = e >> 4, c = e & 15, gr = first_row + row;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gr < limit) v = Ld4<T>::ld(base + (int64_t)gr * row_stride + 4 * c);
    float *d = dst + row * kPitch + 4 * c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// one row of 64 elements -> the lane's 32 operand values: element t is
// row[2 t + half] (the two reduction indices of MFMA step t live in the two
// lane halves)
template <typename T>
__device__ __forceinline__ void load_row_operand(float *reg, const T *row,
                                                 bool ok, int half) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = Ld4<T>::ld(row + 4 * j);
    reg[2 * j] = half ? v.y : v.x;
    reg[2 * j + 1] = half ? v.w : v.z;
  }
}

__device__ __forceinline__ int acc_row(int r, int half) {
  return (r & 3) + 8 * (r >> 2) + 4 * half;
}

// MODE 0: forward.  MODE 1: gradient of Q (same walk over the keys, with the
// probabilities recomputed from the log-sum-exp).  One workgroup = 128 query
// rows of one (batch, head): 4 waves x 32 rows.
template <typename T, int MODE>
__global__ void __launch_bounds__(256) attn_query_kernel(AttnArgs a) {
  __shared__ float Ks[64 * kPitch];
  __shared__ float Vs[64 * kPitch];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int q = blockIdx.x * 128 + wid * 32 + r32;
  const bool q_ok = q < a.Sq;
  const T *qp = (const T *)a.q + b * a.q_sb + head * a.q_sh;
  const T *kp = (const T *)a.k + b * a.k_sb + head * a.k_sh;
  const T *vp = (const T *)a.v + b * a.v_sb + head * a.v_sh;
  const T *mp = a.mask ? (const T *)a.mask + b * a.m_sb + head * a.m_sh +
                             (int64_t)(q_ok ? q : 0) * a.m_sq
                       : nullptr;
  const int64_t orow = (((int64_t)b * a.Sq + (q_ok ? q : 0)) * a.H + head) *
                       kHeadDim;
  const int64_t srow = ((int64_t)b * a.H + head) * a.Sq + (q_ok ? q : 0);

  float qreg[32];
  load_row_operand<T>(qreg, qp + (int64_t)(q_ok ? q : 0) * a.q_ss, q_ok, hh);
  float doreg[MODE == 1 ? 32 : 1];
  float lse_q = 0.f, delta_q = 0.f;
  if (MODE == 1) {
    load_row_operand<T>(doreg, (const T *)a.dout + orow, q_ok, hh);
    lse_q = a.lse[srow];
    delta_q = a.delta[srow];
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  float m_run = kMasked, l_run = 0.f;

  for (int c0 = 0; c0 < a.Sk; c0 += 64) {
    __syncthreads();  // the previous chunk has been consumed by every wave
    stage_rows<T, 64>(Ks, kp, a.k_ss, c0, a.Sk, tid);
    stage_rows<T, 64>(Vs, vp, a.v_ss, c0, a.Sk, tid);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < 2; ++kt) {
      const int kb = c0 + 32 * kt;
      if (kb >= a.Sk) break;
      // S^T[key][query] = sum_d K[key][d] Q[query][d]
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const float *ka = Ks + (32 * kt + r32) * kPitch + hh;
#pragma unroll
      for (int t = 0; t < 32; ++t)
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * t], qreg[t], s, 0, 0,
                                                 0);
      float x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb + acc_row(r, hh);
        float v = s[r] * a.scale;
        if (mp) v += to_f<T>(mp[(int64_t)(key < a.Sk ? key : 0) * a.m_sk]);
        x[r] = key < a.Sk ? v : kMasked;
      }
      if (MODE == 0) {
        float mx = x[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, x[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb + acc_row(r, hh);
          x[r] = key < a.Sk ? __expf(x[r] - m_new) : 0.f;
          ls += x[r];
        }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc0[r] *= alpha;
          acc1[r] *= alpha;
        }
      } else {
        // dP^T[key][query] = sum_d V[key][d] dO[query][d]
        f32x16 dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
        const float *va = Vs + (32 * kt + r32) * kPitch + hh;
#pragma unroll
        for (int t = 0; t < 32; ++t)
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va[2 * t], doreg[t], dp, 0,
                                                    0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb + acc_row(r, hh);
          const float p = key < a.Sk ? __expf(x[r] - lse_q) : 0.f;
          x[r] = p * (dp[r] - delta_q) * a.scale;
        }
      }
      // acc^T[d][query] += sum_key M[key][d] X[key][query], M = V or K:
      // register r of lane half h is key (r & 3) + 8 (r >> 2) + 4 h
      const float *ma = (MODE == 0 ? Vs : Ks) + (32 * kt + 4 * hh) * kPitch +
                        r32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = ((r & 3) + 8 * (r >> 2)) * kPitch;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma[off], x[r], acc0, 0, 0,
                                                    0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma[off + 32], x[r], acc1,
                                                    0, 0, 0);
      }
    }
  }

  if (!q_ok) return;
  float inv = 1.f;
  if (MODE == 0) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (hh == 0) a.lse[srow] = m_run + __logf(fmaxf(l_tot, 1e-37f));
  }
  T *dst = (MODE == 0 ? (T *)a.out : (T *)a.dq) + orow;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = acc_row(r, hh);
    dst[d] = from_f<T>(acc0[r] * inv);
    dst[32 + d] = from_f<T>(acc1[r] * inv);
  }
}

// Gradients of K and V.  One workgroup = 128 keys of one (batch, head), 4
// waves x 32 keys, walking over the queries in tiles of 32.
template <typename T>
__global__ void __launch_bounds__(256) attn_key_kernel(AttnArgs a) {
  __shared__ float Qs[32 * kPitch];
  __shared__ float Ds[32 * kPitch];
  __shared__ float Lq[32], Dq[32];
  __shared__ float Xs[4][2][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int key0 = blockIdx.x * 128 + wid * 32;
  const int key = key0 + r32;
  const bool key_ok = key < a.Sk;
  const T *qp = (const T *)a.q + b * a.q_sb + head * a.q_sh;
  const T *kp = (const T *)a.k + b * a.k_sb + head * a.k_sh;
  const T *vp = (const T *)a.v + b * a.v_sb + head * a.v_sh;
  const T *dop = (const T *)a.dout + (int64_t)b * a.Sq * a.H * kHeadDim +
                 head * kHeadDim;
  const int64_t do_ss = (int64_t)a.H * kHeadDim;
  const T *mp = a.mask ? (const T *)a.mask + b * a.m_sb + head * a.m_sh
                       : nullptr;
  const int64_t sbase = ((int64_t)b * a.H + head) * a.Sq;

  float kreg[32], vreg[32];
  load_row_operand<T>(kreg, kp + (int64_t)(key_ok ? key : 0) * a.k_ss, key_ok,
                      hh);
  load_row_operand<T>(vreg, vp + (int64_t)(key_ok ? key : 0) * a.v_ss, key_ok,
                      hh);
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) dk0[r] = dk1[r] = dv0[r] = dv1[r] = 0.f;
  float *xp = Xs[wid][0], *xd = Xs[wid][1];

  for (int q0 = 0; q0 < a.Sq; q0 += 32) {
    __syncthreads();
    stage_rows<T, 32>(Qs, qp, a.q_ss, q0, a.Sq, tid);
    stage_rows<T, 32>(Ds, dop, do_ss, q0, a.Sq, tid);
    if (tid < 32) {
      const bool ok = q0 + tid < a.Sq;
      Lq[tid] = ok ? a.lse[sbase + q0 + tid] : 0.f;
      Dq[tid] = ok ? a.delta[sbase + q0 + tid] : 0.f;
    }
    __syncthreads();
    // S^T[key][query] and dP^T[key][query]: keys from registers, queries
    // (columns) from the staged tiles
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
    const float *qb = Qs + r32 * kPitch + hh;
    const float *db = Ds + r32 * kPitch + hh;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[t], qb[2 * t], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[t], db[2 * t], dp, 0, 0,
                                                0);
    }
    const int qi = q0 + r32;  // this lane's query column
    const float lse_q = Lq[r32], delta_q = Dq[r32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hh), kk = key0 + row;
      float v = s[r] * a.scale;
      if (mp)
        v += to_f<T>(mp[(int64_t)(qi < a.Sq ? qi : 0) * a.m_sq +
                        (int64_t)(kk < a.Sk ? kk : 0) * a.m_sk]);
      const float p = (qi < a.Sq && kk < a.Sk) ? __expf(v - lse_q) : 0.f;
      xp[row * 33 + r32] = p;
      xd[row * 33 + r32] = p * (dp[r] - delta_q) * a.scale;
    }
    __syncthreads();  // the images are read with the key on the lane
    // dV[key][d] += sum_q P^T[key][q] dO[q][d];  dK[key][d] += sum_q dS^T Q
    const float *pa = xp + r32 * 33 + hh, *da = xd + r32 * 33 + hh;
    const float *dob = Ds + hh * kPitch + r32, *qob = Qs + hh * kPitch + r32;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float pv = pa[2 * t], dsv = da[2 * t];
      const int off = 2 * t * kPitch;
      dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, dob[off], dv0, 0, 0, 0);
      dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, dob[off + 32], dv1, 0, 0,
                                                 0);
      dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(dsv, qob[off], dk0, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dsv, qob[off + 32], dk1, 0, 0,
                                                 0);
    }
  }
  // accumulator: column d on the lane, rows = keys
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kk = key0 + acc_row(r, hh);
    if (kk >= a.Sk) continue;
    const int64_t o = (((int64_t)b * a.Sk + kk) * a.H + head) * kHeadDim + r32;
    ((T *)a.dk)[o] = from_f<T>(dk0[r]);
    ((T *)a.dk)[o + 32] = from_f<T>(dk1[r]);
    ((T *)a.dv)[o] = from_f<T>(dv0[r]);
    ((T *)a.dv)[o + 32] = from_f<T>(dv1[r]);
  }
}

// delta[b, h, q] = sum_d dO[b, q, h, d] * O[b, q, h, d]; one wave per row
template <typename T>
__global__ void attn_delta_kernel(const T *__restrict__ dout,
                                  const T *__restrict__ out,
                                  float *__restrict__ delta, int B, int H,
                                  int Sq) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t rows = (int64_t)B * Sq * H;
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float v = to_f<T>(dout[row * kHeadDim + lane]) *
                  to_f<T>(out[row * kHeadDim + lane]);
  const float s = wave_sum(v);
  if (lane == 0) {
    const int64_t bq = row / H, h = row - bq * H;
    const int64_t b = bq / Sq, q = bq - b * Sq;
    delta[(b * H + h) * Sq + q] = s;
  }
}


// ---------------------------------------------------------------------------
// 16-bit operands on v_mfma_f32_32x32x16_{f16,bf16}
// ---------------------------------------------------------------------------
// The same walk with the operands kept in 16 bits: one MFMA covers 16
// reduction indices, the lane half g supplies indices 8 g .. 8 g + 7 of it.
//
//  * Products that sum over the head dimension (scores, dP) take one operand
//    from registers (the lane's own Q / dO / K / V row, four 16-byte loads) and
//    the other from a row-major LDS image with one 16-byte read.
//  * Products that sum over the rows of a 32x32 accumulator tile X (keys in the
//    query kernels, queries in the key kernel) use X as the B operand straight
//    from its registers: registers 8 m .. 8 m + 7 of lane half g, packed to 16
//    bits, are the 8 values of MFMA m, which thereby sums over rows
//    16 m + (j & 3) + 8 (j >> 2) + 4 g in slot (g, j).  The A operand supplies the
//    other matrix in the same slot order: it is staged TRANSPOSED in LDS
//    ([d][row]), so that a slot group of four rows is one 8-byte read.
//    No LDS round trip and no cross-lane exchange for X.
//  * The key kernel computes S = Q K^T (not S^T) so that its lane owns a key and
//    dK^T, dV^T accumulate exactly like O^T and dQ^T do in the query kernels.
namespace h16a {

constexpr int kRowPitch = 72;  // row-major images: 64 values + 8 (16-byte rows)
constexpr int kTrPitch = 68;   // transposed images: 8-byte reads of 32 rows hit
                               // 64 different banks
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <typename T> struct Frag;
template <> struct Frag<__half> {
  using type = f16x8;
  static __device__ __forceinline__ f32x16 mma(type a, type b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ type pack(const float *x) {
    type f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (_Float16)x[j];
    return f;
  }
};
template <> struct Frag<__hip_bfloat16> {
  using type = bf16x8;
  static __device__ __forceinline__ f32x16 mma(type a, type b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ type pack(const float *x) {
    u16 h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __hip_bfloat16 b = __float2bfloat16(x[j]);
      __builtin_memcpy(&h[j], &b, 2);
    }
    type f;
    __builtin_memcpy(&f, h, 16);
    return f;
  }
};

// A 64 x 64 tile of a row-major 16-bit operand on its way from memory to LDS:
// `load` issues the thread's two 16-byte global loads (the same eight columns
// of two adjacent rows; rows at or beyond `limit` are zero), `store` writes the
// row-major image `rm` (pitch kRowPitch) and / or the transposed image `tr`
// ([64][kTrPitch]).  Keeping the two phases apart lets the loads of the next
// tile fly while the current one is multiplied.  The transposed image is
// written in 32-bit words (two rows of one column): the 32 lanes of a
// half-wave write 32 consecutive words.  256 threads.
struct Tile16 {
  u32x4 v0, v1;

  template <typename T>
  __device__ __forceinline__ void load(const T *base, int64_t row_stride,
                                       int first_row, int limit, int tid) {
    const int rp = tid & 31, c = tid >> 5;
    const int g0 = first_row + 2 * rp;
    v0 = u32x4{0u, 0u, 0u, 0u};
    v1 = u32x4{0u, 0u, 0u, 0u};
    if (g0 < limit)
      v0 = *reinterpret_cast<const u32x4 *>(base + (int64_t)g0 * row_stride +
                                            8 * c);
    if (g0 + 1 < limit)
      v1 = *reinterpret_cast<const u32x4 *>(
          base + (int64_t)(g0 + 1) * row_stride + 8 * c);
  }

  template <bool RM, bool TR>
  __device__ __forceinline__ void store(u16 *rm, u16 *tr, int tid) const {
    const int rp = tid & 31, c = tid >> 5, r0 = 2 * rp;
    if (RM) {
      *reinterpret_cast<u32x4 *>(rm + r0 * kRowPitch + 8 * c) = v0;
      *reinterpret_cast<u32x4 *>(rm + (r0 + 1) * kRowPitch + 8 * c) = v1;
    }
    if (TR) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned a = v0[j], b = v1[j];
        // columns 8 c + 2 j and 8 c + 2 j + 1: (row r0 | row r0 + 1 << 16)
        *reinterpret_cast<unsigned *>(tr + (8 * c + 2 * j) * kTrPitch + r0) =
            (a & 0xffffu) | (b << 16);
        *reinterpret_cast<unsigned *>(tr + (8 * c + 2 * j + 1) * kTrPitch +
                                      r0) = (a >> 16) | (b & 0xffff0000u);
      }
    }
  }
};

// the lane's own row as the four 16-deep operands of a product over d
template <typename T>
__device__ __forceinline__ void load_row_frags(typename Frag<T>::type *f,
                                               const T *row, bool ok, int half) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) v = *reinterpret_cast<const u32x4 *>(row + 16 * t + 8 * half);
    f[t] = __builtin_bit_cast(typename Frag<T>::type, v);
  }
}

template <typename T>
__device__ __forceinline__ typename Frag<T>::type row_frag(const u16 *img,
                                                           int row, int t,
                                                           int half) {
  return *reinterpret_cast<const typename Frag<T>::type *>(
      img + row * kRowPitch + 16 * t + 8 * half);
}

// slot-ordered operand of MFMA m from a transposed image: rows
// r0 + 16 m + 4 g + {0..3} and + 8 of column d
template <typename T>
__device__ __forceinline__ typename Frag<T>::type tr_frag(const u16 *img,
                                                          int d, int r0, int m,
                                                          int half) {
  const u16 *p = img + d * kTrPitch + r0 + 16 * m + 4 * half;
  const uint2 lo = *reinterpret_cast<const uint2 *>(p);
  const uint2 hi = *reinterpret_cast<const uint2 *>(p + 8);
  const u32x4 w = {lo.x, lo.y, hi.x, hi.y};
  return __builtin_bit_cast(typename Frag<T>::type, w);
}

// accumulator pair (d = row, this lane's column) -> 64 contiguous outputs
template <typename T>
__device__ __forceinline__ void store_column(T *dst, const f32x16 &a0,
                                             const f32x16 &a1, float mul,
                                             int half) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    T lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lo[k] = from_f<T>(a0[4 * i + k] * mul);
      hi[k] = from_f<T>(a1[4 * i + k] * mul);
    }
    uint2 w;
    __builtin_memcpy(&w, lo, 8);
    *reinterpret_cast<uint2 *>(dst + 8 * i + 4 * half) = w;
    __builtin_memcpy(&w, hi, 8);
    *reinterpret_cast<uint2 *>(dst + 32 + 8 * i + 4 * half) = w;
  }
}

}  // namespace h16a

// MODE 0: forward, MODE 1: gradient of Q.  128 queries per workgroup.
template <typename T, int MODE>
__global__ void __launch_bounds__(256) attn16_query_kernel(AttnArgs a) {
  using namespace h16a;
  using F = Frag<T>;
  using frag = typename F::type;
  __shared__ __attribute__((aligned(16))) u16 Ks[64 * kRowPitch];
  __shared__ __attribute__((aligned(16))) u16 Vs[MODE == 1 ? 64 * kRowPitch : 8];
  __shared__ __attribute__((aligned(16))) u16 Mt[64 * kTrPitch];  // V^T | K^T
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int q = blockIdx.x * 128 + wid * 32 + r32;
  const bool q_ok = q < a.Sq;
  const T *qp = (const T *)a.q + b * a.q_sb + head * a.q_sh;
  const T *kp = (const T *)a.k + b * a.k_sb + head * a.k_sh;
  const T *vp = (const T *)a.v + b * a.v_sb + head * a.v_sh;
  const T *mp = a.mask ? (const T *)a.mask + b * a.m_sb + head * a.m_sh +
                             (int64_t)(q_ok ? q : 0) * a.m_sq
                       : nullptr;
  const int64_t orow = (((int64_t)b * a.Sq + (q_ok ? q : 0)) * a.H + head) *
                       kHeadDim;
  const int64_t srow = ((int64_t)b * a.H + head) * a.Sq + (q_ok ? q : 0);

  frag qf[4], dof[MODE == 1 ? 4 : 1];
  load_row_frags<T>(qf, qp + (int64_t)(q_ok ? q : 0) * a.q_ss, q_ok, hh);
  float lse_q = 0.f, delta_q = 0.f;
  if (MODE == 1) {
    load_row_frags<T>(dof, (const T *)a.dout + orow, q_ok, hh);
    lse_q = a.lse[srow];
    delta_q = a.delta[srow];
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  float m_run = kMasked, l_run = 0.f;

  // The forward kernel keeps the next chunk of K and V in flight while it
  // multiplies the current one; the dQ kernel has no registers to spare for
  // that at three waves per SIMD and loads each chunk when it needs it.
  Tile16 kt_, vt_;
  if (MODE == 0) {
    kt_.load<T>(kp, a.k_ss, 0, a.Sk, tid);
    vt_.load<T>(vp, a.v_ss, 0, a.Sk, tid);
  }
  for (int c0 = 0; c0 < a.Sk; c0 += 64) {
    __syncthreads();  // the previous chunk has been consumed by every wave
    if (MODE == 0) {
      kt_.store<true, false>(Ks, nullptr, tid);
      vt_.store<false, true>(nullptr, Mt, tid);
    } else {
      kt_.load<T>(kp, a.k_ss, c0, a.Sk, tid);
      vt_.load<T>(vp, a.v_ss, c0, a.Sk, tid);
      kt_.store<true, true>(Ks, Mt, tid);
      vt_.store<true, false>(Vs, nullptr, tid);
    }
    __syncthreads();
    if (MODE == 0 && c0 + 64 < a.Sk) {
      kt_.load<T>(kp, a.k_ss, c0 + 64, a.Sk, tid);
      vt_.load<T>(vp, a.v_ss, c0 + 64, a.Sk, tid);
    }
#pragma unroll 1
    for (int kt = 0; kt < 2; ++kt) {
      const int kb = c0 + 32 * kt;
      if (kb >= a.Sk) break;
      // S^T[key][query] = sum_d K[key][d] Q[query][d]
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        s = F::mma(row_frag<T>(Ks, 32 * kt + r32, t, hh), qf[t], s);
      float x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb + acc_row(r, hh);
        float v = s[r] * a.scale;
        if (mp) v += to_f<T>(mp[(int64_t)(key < a.Sk ? key : 0) * a.m_sk]);
        x[r] = key < a.Sk ? v : kMasked;
      }
      if (MODE == 0) {
        float mx = x[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, x[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb + acc_row(r, hh);
          x[r] = key < a.Sk ? __expf(x[r] - m_new) : 0.f;
          ls += x[r];
        }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc0[r] *= alpha;
          acc1[r] *= alpha;
        }
      } else {
        // dP^T[key][query] = sum_d V[key][d] dO[query][d]
        f32x16 dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          dp = F::mma(row_frag<T>(Vs, 32 * kt + r32, t, hh), dof[t], dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb + acc_row(r, hh);
          const float p = key < a.Sk ? __expf(x[r] - lse_q) : 0.f;
          x[r] = p * (dp[r] - delta_q) * a.scale;
        }
      }
      // acc^T[d][query] += sum_key M[key][d] X[key][query], M = V or K
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const frag xb = F::pack(x + 8 * m);
        acc0 = F::mma(tr_frag<T>(Mt, r32, 32 * kt, m, hh), xb, acc0);
        acc1 = F::mma(tr_frag<T>(Mt, 32 + r32, 32 * kt, m, hh), xb, acc1);
      }
    }
  }

  if (!q_ok) return;
  float inv = 1.f;
  if (MODE == 0) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (hh == 0) a.lse[srow] = m_run + __logf(fmaxf(l_tot, 1e-37f));
  }
  store_column<T>((MODE == 0 ? (T *)a.out : (T *)a.dq) + orow, acc0, acc1, inv,
                  hh);
}

// Gradients of K and V: 128 keys per workgroup, the lane owns a key, queries
// are walked in chunks of 64.
template <typename T>
__global__ void __launch_bounds__(256) attn16_key_kernel(AttnArgs a) {
  using namespace h16a;
  using F = Frag<T>;
  using frag = typename F::type;
  __shared__ __attribute__((aligned(16))) u16 Qs[64 * kRowPitch];
  __shared__ __attribute__((aligned(16))) u16 Ds[64 * kRowPitch];
  __shared__ __attribute__((aligned(16))) u16 Qt[64 * kTrPitch];
  __shared__ __attribute__((aligned(16))) u16 Dt[64 * kTrPitch];
  __shared__ float Lq[64], Dq[64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int key = blockIdx.x * 128 + wid * 32 + r32;
  const bool key_ok = key < a.Sk;
  const T *qp = (const T *)a.q + b * a.q_sb + head * a.q_sh;
  const T *kp = (const T *)a.k + b * a.k_sb + head * a.k_sh;
  const T *vp = (const T *)a.v + b * a.v_sb + head * a.v_sh;
  const T *dop = (const T *)a.dout + (int64_t)b * a.Sq * a.H * kHeadDim +
                 head * kHeadDim;
  const int64_t do_ss = (int64_t)a.H * kHeadDim;
  const T *mp = a.mask ? (const T *)a.mask + b * a.m_sb + head * a.m_sh +
                             (int64_t)(key_ok ? key : 0) * a.m_sk
                       : nullptr;
  const int64_t sbase = ((int64_t)b * a.H + head) * a.Sq;

  frag kf[4], vf[4];
  load_row_frags<T>(kf, kp + (int64_t)(key_ok ? key : 0) * a.k_ss, key_ok, hh);
  load_row_frags<T>(vf, vp + (int64_t)(key_ok ? key : 0) * a.v_ss, key_ok, hh);
  // a mask that does not depend on the query is one value per lane
  const bool mask_per_key = mp != nullptr && a.m_sq == 0;
  const float mask_key = mask_per_key ? to_f<T>(*mp) : 0.f;

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) dk0[r] = dk1[r] = dv0[r] = dv1[r] = 0.f;

  Tile16 qt_, dt_;  // the next chunk of Q and dO, in flight
  qt_.load<T>(qp, a.q_ss, 0, a.Sq, tid);
  dt_.load<T>(dop, do_ss, 0, a.Sq, tid);
  for (int q0 = 0; q0 < a.Sq; q0 += 64) {
    __syncthreads();
    qt_.store<true, true>(Qs, Qt, tid);
    dt_.store<true, true>(Ds, Dt, tid);
    if (tid < 64) {
      const bool ok = q0 + tid < a.Sq;
      Lq[tid] = ok ? a.lse[sbase + q0 + tid] : 0.f;
      Dq[tid] = ok ? a.delta[sbase + q0 + tid] : 0.f;
    }
    __syncthreads();
    if (q0 + 64 < a.Sq) {
      qt_.load<T>(qp, a.q_ss, q0 + 64, a.Sq, tid);
      dt_.load<T>(dop, do_ss, q0 + 64, a.Sq, tid);
    }
#pragma unroll 1
    for (int qt = 0; qt < 2; ++qt) {
      const int qb = q0 + 32 * qt;
      if (qb >= a.Sq) break;
      // S[query][key] and dP[query][key]: queries from the images, the key's
      // own K / V row from registers
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s = F::mma(row_frag<T>(Qs, 32 * qt + r32, t, hh), kf[t], s);
        dp = F::mma(row_frag<T>(Ds, 32 * qt + r32, t, hh), vf[t], dp);
      }
      float p[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * qt + acc_row(r, hh), qi = q0 + row;
        float v = s[r] * a.scale;
        if (mask_per_key)
          v += mask_key;
        else if (mp)
          v += to_f<T>(mp[(int64_t)(qi < a.Sq ? qi : 0) * a.m_sq]);
        p[r] = (qi < a.Sq && key_ok) ? __expf(v - Lq[row]) : 0.f;
        ds[r] = p[r] * (dp[r] - Dq[row]) * a.scale;
      }
      // dV^T[d][key] += sum_q dO[q][d] P[q][key]; dK^T += sum_q Q[q][d] dS
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const frag pb = F::pack(p + 8 * m), sb = F::pack(ds + 8 * m);
        dv0 = F::mma(tr_frag<T>(Dt, r32, 32 * qt, m, hh), pb, dv0);
        dv1 = F::mma(tr_frag<T>(Dt, 32 + r32, 32 * qt, m, hh), pb, dv1);
        dk0 = F::mma(tr_frag<T>(Qt, r32, 32 * qt, m, hh), sb, dk0);
        dk1 = F::mma(tr_frag<T>(Qt, 32 + r32, 32 * qt, m, hh), sb, dk1);
      }
    }
  }
  if (!key_ok) return;
  const int64_t o = (((int64_t)b * a.Sk + key) * a.H + head) * kHeadDim;
  store_column<T>((T *)a.dk + o, dk0, dk1, 1.f, hh);
  store_column<T>((T *)a.dv + o, dv0, dv1, 1.f, hh);
}

}  // namespace sa

using namespace sa;

static bool attn_ok(int dtype, const AttnArgs &a) {
  const uintptr_t al = dtype == kF32 ? 15 : 7;
  auto ok_ptr = [&](const void *p) {
    return p == nullptr || (reinterpret_cast<uintptr_t>(p) & al) == 0;
  };
  const int64_t s[] = {a.q_sb, a.q_sh, a.q_ss, a.k_sb, a.k_sh, a.k_ss,
                       a.v_sb, a.v_sh, a.v_ss};
  for (int64_t v : s)
    if (v % 4 != 0) return false;
  return a.B > 0 && a.H > 0 && a.Sq > 0 && a.Sk > 0 && ok_ptr(a.q) &&
         ok_ptr(a.k) && ok_ptr(a.v) && ok_ptr(a.out) && ok_ptr(a.dout) &&
         ok_ptr(a.dq) && ok_ptr(a.dk) && ok_ptr(a.dv) && a.H <= 65535 &&
         a.B <= 65535;
}

// the 16-bit MFMA kernels load rows in 16-byte pieces
static const bool kAttn16 = [] {
  const char *v = getenv("SINGA_ATTENTION_16BIT");
  return !(v && *v == '0');
}();

static bool attn16_ok(int dtype, const AttnArgs &a) {
  if (!kAttn16 || (dtype != kF16 && dtype != kBF16)) return false;
  const int64_t s[] = {a.q_sb, a.q_sh, a.q_ss, a.k_sb, a.k_sh, a.k_ss,
                       a.v_sb, a.v_sh, a.v_ss};
  for (int64_t v : s)
    if (v % 8 != 0) return false;
  const void *p[] = {a.q, a.k, a.v, a.out, a.dout, a.dq, a.dk, a.dv};
  for (const void *x : p)
    if (x != nullptr && (reinterpret_cast<uintptr_t>(x) & 15) != 0) return false;
  return true;
}

static AttnArgs attn_args(const AttnDesc *d) {
  AttnArgs a{};
  a.q = d->q; a.k = d->k; a.v = d->v; a.mask = d->mask;
  a.q_sb = d->q_stride[0]; a.q_sh = d->q_stride[1]; a.q_ss = d->q_stride[2];
  a.k_sb = d->k_stride[0]; a.k_sh = d->k_stride[1]; a.k_ss = d->k_stride[2];
  a.v_sb = d->v_stride[0]; a.v_sh = d->v_stride[1]; a.v_ss = d->v_stride[2];
  a.m_sb = d->m_stride[0]; a.m_sh = d->m_stride[1]; a.m_sq = d->m_stride[2];
  a.m_sk = d->m_stride[3];
  a.B = d->B; a.H = d->H; a.Sq = d->Sq; a.Sk = d->Sk;
  a.scale = d->scale;
  return a;
}

int sa_attention_fwd(int dtype, const AttnDesc *d, void *out, float *lse,
                     hipStream_t s) {
  if (d->D != kHeadDim) return -2;
  AttnArgs a = attn_args(d);
  a.out = out;
  a.lse = lse;
  if (!attn_ok(dtype, a)) return -2;
  const dim3 grid((a.Sq + 127) / 128, a.H, a.B);
  if (attn16_ok(dtype, a)) {
    if (dtype == kF16)
      hipLaunchKernelGGL((attn16_query_kernel<__half, 0>), grid, dim3(256), 0,
                         s, a);
    else
      hipLaunchKernelGGL((attn16_query_kernel<__hip_bfloat16, 0>), grid,
                         dim3(256), 0, s, a);
    return 0;
  }
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((attn_query_kernel<T, 0>), grid, dim3(256), 0, s, a);
  });
  return 0;
}

int sa_attention_bwd(int dtype, const AttnDesc *d, const void *out,
                     const void *dout, const float *lse, float *delta,
                     void *dq, void *dk, void *dv, hipStream_t s) {
  if (d->D != kHeadDim) return -2;
  AttnArgs a = attn_args(d);
  a.out = const_cast<void *>(out);
  a.dout = dout;
  a.lse = const_cast<float *>(lse);
  a.delta = delta;
  a.dq = dq; a.dk = dk; a.dv = dv;
  if (!attn_ok(dtype, a)) return -2;
  const int64_t rows = (int64_t)a.B * a.Sq * a.H;
  if (attn16_ok(dtype, a)) {
    const dim3 gk((a.Sk + 127) / 128, a.H, a.B), gq((a.Sq + 127) / 128, a.H, a.B);
    const dim3 gd((unsigned)((rows + 3) / 4));
    if (dtype == kF16) {
      using T = __half;
      hipLaunchKernelGGL((attn_delta_kernel<T>), gd, dim3(256), 0, s,
                         (const T *)dout, (const T *)out, delta, a.B, a.H, a.Sq);
      hipLaunchKernelGGL((attn16_key_kernel<T>), gk, dim3(256), 0, s, a);
      hipLaunchKernelGGL((attn16_query_kernel<T, 1>), gq, dim3(256), 0, s, a);
    } else {
      using T = __hip_bfloat16;
      hipLaunchKernelGGL((attn_delta_kernel<T>), gd, dim3(256), 0, s,
                         (const T *)dout, (const T *)out, delta, a.B, a.H, a.Sq);
      hipLaunchKernelGGL((attn16_key_kernel<T>), gk, dim3(256), 0, s, a);
      hipLaunchKernelGGL((attn16_query_kernel<T, 1>), gq, dim3(256), 0, s, a);
    }
    return 0;
  }
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((attn_delta_kernel<T>), dim3((unsigned)((rows + 3) / 4)),
                       dim3(256), 0, s, (const T *)dout, (const T *)out, delta,
                       a.B, a.H, a.Sq);
    hipLaunchKernelGGL((attn_key_kernel<T>),
                       dim3((a.Sk + 127) / 128, a.H, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL((attn_query_kernel<T, 1>),
                       dim3((a.Sq + 127) / 128, a.H, a.B), dim3(256), 0, s, a);
  });
  return 0;
}
