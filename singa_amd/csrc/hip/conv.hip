// 2-d convolution (NCHW, cross-correlation, dilation 1, groups) as implicit
// GEMM on the CDNA4 matrix cores: forward, backward-data and backward-weight
// share one tiled MFMA kernel and differ only in how the two GEMM operands are
// gathered and where the result tile goes.
//
//   forward    Y[k, (n,p,q)]   = sum_(c,r,s) W[k,(c,r,s)] * im2col(X)[(c,r,s),(n,p,q)]
//   bwd data   dX[c, (n,h,w)]  = sum_(k,r,s) W^T[c,(k,r,s)] * col2im(dY)[(k,r,s),(n,h,w)]
//   bwd weight dW[k, (c,r,s)]  = sum_(n,p,q) dY[k,(n,p,q)] * im2col(X)^T[(n,p,q),(c,r,s)]
//
// In NCHW the pixel index is the contiguous one, so pixels sit on the GEMM
// column (lane) dimension for forward / backward-data: gathers are coalesced
// along q and the 32x32 accumulator layout (lane = column) stores 128-byte
// runs of one output channel.  Backward-weight reduces over pixels; it walks
// them as the GEMM K dimension (again coalesced), splits that range over
// grid.z and combines the partial tiles with float atomics (two 128-byte
// segments per wave instruction).  No im2col buffer is ever materialised.
//
// Block = 256 threads (2 x 2 waves), tile BM x BN x 16, k-major LDS tiles,
// register-staged prefetch of the next tile under the current tile's MFMAs.
// float32 runs on v_mfma_f32_32x32x2_f32; fp16 / bf16 inputs are widened when
// staged and use the same kernel (float accumulate).
//
// Replaces cudnnConvolutionForward / BackwardData / BackwardFilter /
// BackwardBias of the reference (src/model/operation/convolution.cc:585-703).
#include "common.cuh"
#include "kernels.h"

namespace sa {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// n / d for n < 2^31 with a precomputed multiplier
struct FastDiv {
  uint32_t d, mul, shift;
  FastDiv() : d(1), mul(0), shift(0) {}
  explicit FastDiv(uint32_t div) : d(div) {
    shift = 0;
    while ((1u << shift) < div) ++shift;
    uint64_t one = 1;
    mul = (uint32_t)(((one << 32) * ((one << shift) - div)) / div + 1);
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const {
    return (__umulhi(n, mul) + n) >> shift;
  }
  __device__ __forceinline__ void divmod(uint32_t n, uint32_t &q,
                                         uint32_t &r) const {
    q = div(n);
    r = n - q * d;
  }
};

enum { CONV_FWD = 0, CONV_BWD_DATA = 1, CONV_BWD_WEIGHT = 2 };

struct ConvArgs {
  ConvGeom g;
  const void *a;  // fwd: W        bwd-data: W        bwd-weight: dY
  const void *b;  // fwd: X        bwd-data: dY       bwd-weight: X
  void *out;      // fwd: Y        bwd-data: dX       bwd-weight: dW (float)
  const void *bias;
  int M, Ngemm, Kgemm;  // GEMM extents per group
  int Cg, Kg, RS, PQ, HW;
  int splits, k_per_split;
  FastDiv div_rs, div_s, div_pq, div_q, div_hw, div_w, div_sh, div_sw;
};

template <typename T, int MODE, int BM, int BN>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(ConvArgs a) {
  constexpr int BK = 16;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int A_PER = BM * BK / 256, B_PER = BN * BK / 256;
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];

  const ConvGeom &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int grp = blockIdx.z / a.splits, split = blockIdx.z % a.splits;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const T *pa = (const T *)a.a;
  const T *pb = (const T *)a.b;

  int kbeg = 0, kend = a.Kgemm;
  if (MODE == CONV_BWD_WEIGHT) {
    kbeg = split * a.k_per_split;
    kend = min(a.Kgemm, kbeg + a.k_per_split);
  }

  // ---- per-thread constants of the gathers --------------------------------
  // forward / backward-data: the thread owns one pixel column of the B tile
  int bj_valid = 0, b_n = 0, b_y = 0, b_x = 0;
  // backward-weight: the thread owns B_PER (c, r, s) columns
  int bw_off[B_PER], bw_r[B_PER], bw_s[B_PER];
  if (MODE == CONV_FWD || MODE == CONV_BWD_DATA) {
    const int j = n0 + tid % BN;
    bj_valid = j < a.Ngemm;
    if (bj_valid) {
      uint32_t n, pix, y, x;
      if (MODE == CONV_FWD) {
        a.div_pq.divmod(j, n, pix);
        a.div_q.divmod(pix, y, x);
        b_y = y * g.stride_h - g.pad_h;
        b_x = x * g.stride_w - g.pad_w;
      } else {
        a.div_hw.divmod(j, n, pix);
        a.div_w.divmod(pix, y, x);
        b_y = y + g.pad_h;
        b_x = x + g.pad_w;
      }
      b_n = n;
    }
  } else {
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int j = n0 + tid / BK + (256 / BK) * i;
      if (j < a.Ngemm) {
        uint32_t c, rs, r, s;
        a.div_rs.divmod(j, c, rs);
        a.div_s.divmod(rs, r, s);
        bw_r[i] = (int)r - g.pad_h;
        bw_s[i] = (int)s - g.pad_w;
        bw_off[i] = ((grp * a.Cg + (int)c) * g.H + bw_r[i]) * g.W + bw_s[i];
      } else {
        // far outside the image: the bounds test rejects every tap
        bw_off[i] = 0;
        bw_r[i] = bw_s[i] = -(1 << 28);
      }
    }
  }

  float ra[A_PER], rb[B_PER];

  auto load_tiles = [&](int k0) {
    if (MODE == CONV_FWD) {
      // A = W[k][(c,r,s)], contiguous along kk: kk fastest over lanes
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int kk = k0 + tid % BK, m = m0 + tid / BK + (256 / BK) * i;
        ra[i] = (m < a.M && kk < kend)
                    ? to_f<T>(pa[(size_t)(grp * a.Kg + m) * a.Kgemm + kk])
                    : 0.f;
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int kk = k0 + tid / BN + (256 / BN) * i;
        float v = 0.f;
        if (bj_valid && kk < kend) {
          uint32_t c, rs, r, s;
          if (a.RS == 1) { c = kk; r = 0; s = 0; }
          else { a.div_rs.divmod(kk, c, rs); a.div_s.divmod(rs, r, s); }
          const int ih = b_y + (int)r, iw = b_x + (int)s;
          if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
            v = to_f<T>(pb[((size_t)(b_n * g.C + grp * a.Cg + c) * g.H + ih) *
                               g.W + iw]);
        }
        rb[i] = v;
      }
    } else if (MODE == CONV_BWD_DATA) {
      // A = W^T[c][(k,r,s)]: c fastest over lanes (stride RS in memory)
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + tid % BM, kk = k0 + tid / BM + (256 / BM) * i;
        float v = 0.f;
        if (m < a.M && kk < kend) {
          uint32_t k, rs;
          if (a.RS == 1) { k = kk; rs = 0; }
          else a.div_rs.divmod(kk, k, rs);
          v = to_f<T>(pa[((size_t)(grp * a.Kg + k) * a.Cg + m) * a.RS + rs]);
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int kk = k0 + tid / BN + (256 / BN) * i;
        float v = 0.f;
        if (bj_valid && kk < kend) {
          uint32_t k, rs, r, s;
          if (a.RS == 1) { k = kk; r = 0; s = 0; }
          else { a.div_rs.divmod(kk, k, rs); a.div_s.divmod(rs, r, s); }
          const int py = b_y - (int)r, px = b_x - (int)s;
          if (py >= 0 && px >= 0) {
            uint32_t p, pr, q, qr;
            a.div_sh.divmod(py, p, pr);
            a.div_sw.divmod(px, q, qr);
            if (pr == 0 && qr == 0 && (int)p < g.P && (int)q < g.Q)
              v = to_f<T>(pb[((size_t)(b_n * g.K + grp * a.Kg + k) * g.P + p) *
                                 g.Q + q]);
          }
        }
        rb[i] = v;
      }
    } else {
      // reduction index kk = (n, p, q), one per thread and tile
      const int kk = k0 + tid % BK;
      const bool kv = kk < kend;
      uint32_t n = 0, pq = 0, p = 0, q = 0;
      if (kv) {
        a.div_pq.divmod(kk, n, pq);
        a.div_q.divmod(pq, p, q);
      }
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + tid / BK + (256 / BK) * i;
        ra[i] = (kv && m < a.M)
                    ? to_f<T>(pa[(size_t)(n * g.K + grp * a.Kg + m) * a.PQ + pq])
                    : 0.f;
      }
      const int ih0 = p * g.stride_h, iw0 = q * g.stride_w;
      const size_t base = (size_t)n * g.C * a.HW + (size_t)ih0 * g.W + iw0;
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        float v = 0.f;
        if (kv) {
          const int ih = ih0 + bw_r[i], iw = iw0 + bw_s[i];
          if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
            v = to_f<T>(pb[base + bw_off[i]]);
        }
        rb[i] = v;
      }
    }
  };

  auto store_tiles = [&]() {
    if (MODE == CONV_FWD || MODE == CONV_BWD_WEIGHT) {
#pragma unroll
      for (int i = 0; i < A_PER; ++i)
        As[(tid % BK) * LDA + tid / BK + (256 / BK) * i] = ra[i];
    } else {
#pragma unroll
      for (int i = 0; i < A_PER; ++i)
        As[(tid / BM + (256 / BM) * i) * LDA + tid % BM] = ra[i];
    }
    if (MODE == CONV_BWD_WEIGHT) {
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        Bs[(tid % BK) * LDB + tid / BK + (256 / BK) * i] = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        Bs[(tid / BN + (256 / BN) * i) * LDB + tid % BN] = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = (kend - kbeg + BK - 1) / BK;
  if (nt > 0) {
    load_tiles(kbeg);
    store_tiles();
  }
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) load_tiles(kbeg + (t + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[i] = As[(ks + lk) * LDA + wm * WM + 32 * i + lrow];
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bv[j] = Bs[(ks + lk) * LDB + wn * WN + 32 * j + lrow];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j],
                                                           acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (t + 1 < nt) {
      store_tiles();
      __syncthreads();
    }
  }

  // ---- epilogue ---------------------------------------------------------------
  // lane holds GEMM column (lane & 31); register r holds GEMM row
  // (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) of its 32 x 32 block
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + 32 * j + (lane & 31);
    if (col >= a.Ngemm) continue;
    size_t col_off;
    int chan_stride;
    if (MODE == CONV_FWD) {
      uint32_t n, pix;
      a.div_pq.divmod(col, n, pix);
      col_off = (size_t)n * g.K * a.PQ + pix;
      chan_stride = a.PQ;
    } else if (MODE == CONV_BWD_DATA) {
      uint32_t n, pix;
      a.div_hw.divmod(col, n, pix);
      col_off = (size_t)n * g.C * a.HW + pix;
      chan_stride = a.HW;
    } else {
      col_off = col;
      chan_stride = a.Ngemm;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row =
            m0 + wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= a.M) continue;
        const float v = acc[i][j][r];
        if (MODE == CONV_FWD) {
          const int ch = grp * a.Kg + row;
          const float bv = a.bias ? to_f<T>(((const T *)a.bias)[ch]) : 0.f;
          ((T *)a.out)[col_off + (size_t)ch * chan_stride] = from_f<T>(v + bv);
        } else if (MODE == CONV_BWD_DATA) {
          const int ch = grp * a.Cg + row;
          ((T *)a.out)[col_off + (size_t)ch * chan_stride] = from_f<T>(v);
        } else {
          float *dst = (float *)a.out +
                       (size_t)(grp * a.Kg + row) * chan_stride + col_off;
          if (a.splits == 1) *dst = v;
          else atomicAdd(dst, v);
        }
      }
  }
}

// dW accumulated in float -> parameter dtype
template <typename T>
__global__ void cast_out_kernel(const float *__restrict__ in,
                                T *__restrict__ out, int64_t n) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    out[i] = from_f<T>(in[i]);
}

// db[k] = sum_{n, pq} dy[n, k, pq]: one block per channel
template <typename T>
__global__ void conv_bias_grad_kernel(const T *__restrict__ dy,
                                      T *__restrict__ db, int N, int K, int PQ) {
  __shared__ float smem[32];
  const int k = blockIdx.x;
  float acc = 0.f;
  for (int n = 0; n < N; ++n) {
    const T *p = dy + ((size_t)n * K + k) * PQ;
    for (int i = threadIdx.x; i < PQ; i += blockDim.x) acc += to_f<T>(p[i]);
  }
  acc = block_sum(acc, smem);
  if (threadIdx.x == 0) db[k] = from_f<T>(acc);
}

static ConvArgs make_args(const ConvGeom &g, int mode) {
  ConvArgs a;
  a.g = g;
  a.Cg = g.C / g.groups;
  a.Kg = g.K / g.groups;
  a.RS = g.R * g.S;
  a.PQ = g.P * g.Q;
  a.HW = g.H * g.W;
  a.splits = 1;
  a.k_per_split = 0;
  a.bias = nullptr;
  if (mode == CONV_FWD) {
    a.M = a.Kg; a.Ngemm = g.N * a.PQ; a.Kgemm = a.Cg * a.RS;
  } else if (mode == CONV_BWD_DATA) {
    a.M = a.Cg; a.Ngemm = g.N * a.HW; a.Kgemm = a.Kg * a.RS;
  } else {
    a.M = a.Kg; a.Ngemm = a.Cg * a.RS; a.Kgemm = g.N * a.PQ;
  }
  a.div_rs = FastDiv(a.RS);
  a.div_s = FastDiv(g.S);
  a.div_pq = FastDiv(a.PQ);
  a.div_q = FastDiv(g.Q);
  a.div_hw = FastDiv(a.HW);
  a.div_w = FastDiv(g.W);
  a.div_sh = FastDiv(g.stride_h);
  a.div_sw = FastDiv(g.stride_w);
  return a;
}

template <typename T, int MODE>
void launch_conv(ConvArgs &a, hipStream_t s) {
  const int64_t t128 = (int64_t)((a.M + 127) / 128) * ((a.Ngemm + 127) / 128) *
                       a.g.groups * a.splits;
  const bool big = a.M >= 96 && a.Ngemm >= 96 && t128 >= 256;
  if (big) {
    dim3 grid((a.Ngemm + 127) / 128, (a.M + 127) / 128, a.g.groups * a.splits);
    hipLaunchKernelGGL((conv_igemm_kernel<T, MODE, 128, 128>), grid, dim3(256),
                       0, s, a);
  } else {
    dim3 grid((a.Ngemm + 63) / 64, (a.M + 63) / 64, a.g.groups * a.splits);
    hipLaunchKernelGGL((conv_igemm_kernel<T, MODE, 64, 64>), grid, dim3(256), 0,
                       s, a);
  }
}

}  // namespace sa

using namespace sa;

static bool geom_ok(const ConvGeom *g) {
  if (g->groups < 1 || g->C % g->groups || g->K % g->groups) return false;
  const int64_t lim = (int64_t)1 << 31;
  return (int64_t)g->N * g->C * g->H * g->W < lim &&
         (int64_t)g->N * g->K * g->P * g->Q < lim;
}

int sa_conv_fwd(int dtype, const ConvGeom *g, const void *x, const void *w,
                const void *bias, void *y, hipStream_t s) {
  if (!geom_ok(g)) return -2;
  ConvArgs a = make_args(*g, CONV_FWD);
  a.a = w; a.b = x; a.out = y; a.bias = bias;
  SA_DISPATCH_FLOAT(dtype, T, { launch_conv<T, CONV_FWD>(a, s); });
  return 0;
}

int sa_conv_bwd_data(int dtype, const ConvGeom *g, const void *dy,
                     const void *w, void *dx, hipStream_t s) {
  if (!geom_ok(g)) return -2;
  ConvArgs a = make_args(*g, CONV_BWD_DATA);
  a.a = w; a.b = dy; a.out = dx;
  SA_DISPATCH_FLOAT(dtype, T, { launch_conv<T, CONV_BWD_DATA>(a, s); });
  return 0;
}

int sa_conv_bwd_weight(int dtype, const ConvGeom *g, const void *dy,
                       const void *x, void *dw, hipStream_t s) {
  if (!geom_ok(g)) return -2;
  if (dtype != kF32) return -3;  // half types accumulate via a float buffer
  ConvArgs a = make_args(*g, CONV_BWD_WEIGHT);
  a.a = dy; a.b = x; a.out = dw;
  // split the pixel reduction until the launch has >= ~512 blocks
  const int64_t tiles = (int64_t)((a.M + 63) / 64) * ((a.Ngemm + 63) / 64) *
                        g->groups;
  int64_t splits = 1;
  if (tiles < 512) splits = (512 + tiles - 1) / tiles;
  const int64_t max_splits = (a.Kgemm + 255) / 256;  // >= 16 k-tiles each
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int kps = (int)((a.Kgemm + splits - 1) / splits);
  kps = (kps + 15) / 16 * 16;
  a.k_per_split = kps;
  a.splits = (a.Kgemm + kps - 1) / kps;
  if (a.splits > 1)
    hipMemsetAsync(dw, 0, (size_t)g->K * a.Cg * a.RS * sizeof(float), s);
  launch_conv<float, CONV_BWD_WEIGHT>(a, s);
  return 0;
}

int sa_conv_bwd_bias(int dtype, const void *dy, void *db, int N, int K, int PQ,
                     hipStream_t s) {
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((conv_bias_grad_kernel<T>), dim3(K), dim3(256), 0, s,
                       (const T *)dy, (T *)db, N, K, PQ);
  });
  return 0;
}
