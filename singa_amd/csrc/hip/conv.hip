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
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

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
  // backward-data: tensor of dX's shape added to the result (the gradient that
  // reached the same input along another path); may alias `out`
  const void *addend;
  int M, Ngemm, Kgemm;  // GEMM extents per group
  int Cg, Kg, RS, PQ, HW;
  int splits, k_per_split;
  int mt, nt;      // tile counts along M and N
  int a_vec;       // A rows may be read four reduction indices at a time
  int ncls;        // backward-data: stride classes in this launch
  int tap_mask;    // filter has <= 63 taps: bounds tests use a per-pixel mask
  // backward-weight: GEMM columns are ordered (tap, channel) instead of
  // (channel, tap), and a tile's columns all belong to one tap: the bounds
  // test of a gathered input element is then the same for every column of
  // the thread and is made once per reduction tile
  int tap_major;
  // with tap_major and more than one tap: `out` is a scratch laid out
  // [tap][K][Cg] (channels contiguous, so a wave's atomics cover whole
  // 128-byte runs); untap_kernel moves it to dw[K][Cg][tap] afterwards
  int tap_scratch;
  int stagger;     // per-wave-slot issue priorities (see the kernel)
  // deterministic mode: split s stores its partial tile to out + s * slab
  // (plain stores, every element exactly once) instead of adding atomically
  int slabs;
  int64_t slab;
  int64_t a_elems, b_elems;  // elements of the two operand tensors
  FastDiv div_rs, div_s, div_pq, div_q, div_hw, div_w, div_cg;
  // Backward-data with stride (sh, sw) splits into sh * sw independent
  // stride-1 problems: input pixels with (h % sh, w % sw) = (ch, cw) receive
  // contributions only from the taps r = r0 + sh * rr, s = s0 + sw * ss, and
  // the output pixel hit by tap (rr, ss) is p = hh + off_h - rr,
  // q = ww + off_w - ss for h = ch + sh * hh, w = cw + sw * ww.  A class is a
  // dense convolution over its Hc x Wc sub-grid with an Rc x Sc sub-filter,
  // so no multiply is spent on the zeros of the dilated gradient.
  struct Cls {
    int ch, cw, r0, s0, Rc, Sc, Hc, Wc, off_h, off_w, Kgemm, Ngemm;
    FastDiv div_rcsc, div_sc, div_hcwc, div_wc;
  } cls[4];
};

// ---- buffer loads -----------------------------------------------------------
// Tile staging reads through buffer resources: a lane whose element lies
// outside the tile, the image or the reduction range is given a byte offset
// past the end of the resource and the hardware returns zero for it, so the
// gathers are straight-line code without per-element branches.
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int kOob = (int)0x80000000;  // beyond any resource (<= 2^31 - 1 bytes)

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, int64_t bytes) {
  const int n = bytes > 0x7fffffff ? 0x7fffffff : (bytes < 0 ? 0 : (int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n,
                                           0x00020000);
}

template <typename T> struct BufLd;
template <> struct BufLd<float> {
  static __device__ __forceinline__ float one(rsrc_t r, int off) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
  }
  static __device__ __forceinline__ void four(rsrc_t r, int off, float *o) {
    // the loaded vector is reinterpreted as a whole, never indexed directly
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f w = __builtin_bit_cast(
        v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = w.w;
  }
};
template <> struct BufLd<__half> {
  static __device__ __forceinline__ float cvt(unsigned short b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
  static __device__ __forceinline__ float one(rsrc_t r, int off) {
    return cvt(__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
  }
  static __device__ __forceinline__ void four(rsrc_t r, int off, float *o) {
    typedef _Float16 v4h __attribute__((ext_vector_type(4)));
    const v4h w = __builtin_bit_cast(
        v4h, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    o[0] = (float)w.x; o[1] = (float)w.y; o[2] = (float)w.z; o[3] = (float)w.w;
  }
};
template <> struct BufLd<__hip_bfloat16> {
  static __device__ __forceinline__ float one(rsrc_t r, int off) {
    const unsigned b = __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0);
    return __builtin_bit_cast(float, b << 16);
  }
  static __device__ __forceinline__ void four(rsrc_t r, int off, float *o) {
    typedef unsigned short v4s __attribute__((ext_vector_type(4)));
    const v4s w = __builtin_bit_cast(
        v4s, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    o[0] = __builtin_bit_cast(float, (unsigned)w.x << 16);
    o[1] = __builtin_bit_cast(float, (unsigned)w.y << 16);
    o[2] = __builtin_bit_cast(float, (unsigned)w.z << 16);
    o[3] = __builtin_bit_cast(float, (unsigned)w.w << 16);
  }
};

// 16-bit element types: a lane's fragment of the 32x32x16 MFMA is eight
// consecutive reduction indices of one row.  LDS tiles hold floats converted
// from the 16-bit inputs, so narrowing them back is exact.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Frag16;
template <> struct Frag16<__half> {
  using V = h16x8;
  static __device__ __forceinline__ unsigned two(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));
  }
  static __device__ __forceinline__ f32x16 mma(V a, V b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Frag16<__hip_bfloat16> {
  using V = b16x8;
  static __device__ __forceinline__ unsigned two(float lo, float hi) {
    return (__builtin_bit_cast(unsigned, lo) >> 16) |
           (__builtin_bit_cast(unsigned, hi) & 0xffff0000u);
  }
  static __device__ __forceinline__ f32x16 mma(V a, V b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Frag16<float> {  // never instantiated for MFMA
  using V = u32x4;
  static __device__ __forceinline__ unsigned two(float, float) { return 0; }
  static __device__ __forceinline__ f32x16 mma(V, V, f32x16 c) { return c; }
};

// eight floats at p[0], p[ld], ..., p[7 * ld] -> one packed fragment
template <typename T>
__device__ __forceinline__ typename Frag16<T>::V frag16(const float *p, int ld) {
  u32x4 w;
  w.x = Frag16<T>::two(p[0], p[ld]);
  w.y = Frag16<T>::two(p[2 * ld], p[3 * ld]);
  w.z = Frag16<T>::two(p[4 * ld], p[5 * ld]);
  w.w = Frag16<T>::two(p[6 * ld], p[7 * ld]);
  return __builtin_bit_cast(typename Frag16<T>::V, w);
}

// Diagnostic build (-DSA_CONV_PROF): wave 0 of every workgroup stamps the
// shader clock around the phases of its life and adds the sums to the eight
// 64-bit words of its own record at ConvGeom::prof {prologue, first tile,
// plan + load issue, MFMA, wait + LDS store, barrier, epilogue, start time}.
#ifdef SA_CONV_PROF
__device__ __forceinline__ unsigned long long prof_stamp() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PROF_STAMP(v) const unsigned long long v = prof_stamp()
#define PROF_ADD(acc, a, b) acc += (b) - (a)
#else
#define PROF_STAMP(v)
#define PROF_ADD(acc, a, b)
#endif

// V1: both operands are plain matrices (1x1 filter, stride 1, no padding,
// H*W and the channel count multiples of 4): tiles are staged with 4-wide
// loads and no per-element index arithmetic.
//
// V4 (float, forward / backward-data, stride 1, row length a multiple of 4,
// at most one padding column per side): a thread gathers four adjacent
// pixels of one filter tap with a single 16-byte load -- in NCHW they are
// adjacent in memory whatever the tap's horizontal shift, and a 4-byte
// aligned address is all a dwordx4 buffer load needs.  A group that hangs
// over the left / right image border is loaded one pixel further in and
// shifted into place with the overhanging element zeroed.
template <typename T, int MODE, int BM, int BN, int BK, bool V1, bool V4 = false>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(ConvArgs a) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  // Row pitch of the k-major LDS tiles: rows written 4-wide need 16-byte
  // alignment (pitch % 4 == 0); rows filled by transposing stores (reduction
  // index over lanes) get an odd pitch, which spreads them over all banks.
  constexpr bool A_T = MODE != CONV_BWD_DATA, B_T = MODE == CONV_BWD_WEIGHT;
  constexpr int LDA = BM + (A_T ? 1 : 4), LDB = BN + (B_T ? 1 : 4);
  constexpr int A_PER = BM * BK / 256, B_PER = BN * BK / 256;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const ConvGeom &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  if (a.stagger) {
    // Waves that share a SIMD come from different workgroups that started
    // together and do identical work: left alone they stay in lockstep, all
    // in their MFMA phase (sharing the matrix pipe) and then all in their
    // load / store phase (pipe idle).  A distinct issue priority per wave
    // slot lets one wave at a time run its MFMA phase at the full rate, which
    // pulls the phases apart.
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 3;  // HW_ID.wave_id
    if (slot == 1) __builtin_amdgcn_s_setprio(1);
    else if (slot == 2) __builtin_amdgcn_s_setprio(2);
    else if (slot == 3) __builtin_amdgcn_s_setprio(3);
  }
  if constexpr (MODE == CONV_BWD_DATA) {
    if (g.side_zero) grid_zero_words(g.side_zero, g.side_zero_words);
  }
#ifdef SA_CONV_PROF
  unsigned long long pf_plan = 0, pf_mma = 0, pf_store = 0, pf_bar = 0;
#endif
  PROF_STAMP(pf_t0);

  // ---- tile assignment -------------------------------------------------------
  // Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with a
  // private L2.  The tile list is cut into 8 contiguous chunks, one per XCD,
  // ordered so that tiles reading the same pixels (forward / backward-data:
  // the M tiles of one pixel tile; backward-weight: the tiles of one
  // reduction split) are neighbours on one XCD: the shared operand is then
  // fetched from HBM once per XCD instead of once per tile.
  const int cls_i = MODE == CONV_BWD_DATA ? blockIdx.z % a.ncls : 0;
  const int grp = MODE == CONV_BWD_DATA ? blockIdx.z / a.ncls : blockIdx.z;
  const ConvArgs::Cls &bc = a.cls[cls_i];
  const int Ngemm = MODE == CONV_BWD_DATA ? bc.Ngemm : a.Ngemm;
  const int Kgemm = MODE == CONV_BWD_DATA ? bc.Kgemm : a.Kgemm;
  int mt_i, nt_i, split = blockIdx.y;
  {
    // XCD x works through the x-th eighth of the tile list
    const int w = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tiles = a.mt * a.nt;
    if (MODE == CONV_BWD_WEIGHT) {
      // split-major list: one XCD stays on one slice of the pixels
      if (w >= tiles * a.splits) return;
      split = w / tiles;
      const int tile = w - split * tiles;
      nt_i = tile / a.mt;
      mt_i = tile - nt_i * a.mt;
    } else {
      // pixel-tile-major list: the M tiles of one pixel tile are adjacent
      if (w >= tiles) return;
      nt_i = w / a.mt;
      mt_i = w - nt_i * a.mt;
      if (nt_i * BN >= Ngemm) return;
    }
  }
  const int m0 = mt_i * BM, n0 = nt_i * BN;
  const T *pa = (const T *)a.a;
  const T *pb = (const T *)a.b;

  int kbeg = 0, kend = Kgemm;
  if (a.splits > 1) {
    kbeg = split * a.k_per_split;
    kend = min(Kgemm, kbeg + a.k_per_split);
    if (kend <= kbeg) return;  // nothing to add to the zero-filled output
  }

  // ---- resources ---------------------------------------------------------------
  // Offsets into a resource are 32-bit byte offsets, so each resource starts
  // at the first image its tile can touch: forward / backward-data rebase B
  // once per workgroup, backward-weight rebases both operands per K tile.
  constexpr int SZ = (int)sizeof(T);
  const int64_t a_img = (int64_t)g.K * a.PQ;  // backward-weight A = dY
  const int64_t b_img =
      MODE == CONV_BWD_DATA ? (int64_t)g.K * a.PQ : (int64_t)g.C * a.HW;
  int n_base = 0;
  if (MODE == CONV_FWD) n_base = a.div_pq.div(n0);
  if (MODE == CONV_BWD_DATA) n_base = bc.div_hcwc.div(n0);
  rsrc_t rA = make_rsrc(pa, a.a_elems * SZ);
  rsrc_t rB = make_rsrc(pb + n_base * b_img, (a.b_elems - n_base * b_img) * SZ);

  // ---- per-thread constants of the gathers --------------------------------
  // forward / backward-data: the thread owns one pixel column of the B tile
  // (scalar path) or four adjacent ones (V1)
  int bj_valid = 0, b_y = 0, b_x = 0, b_base = 0;
  // backward-weight: the thread owns B_PER (c, r, s) columns
  int bw_off[B_PER], bw_r[B_PER], bw_s[B_PER];
  int tm_r = 0, tm_s = 0;   // tap-major order: the tile's tap, minus the padding
  constexpr int SZ_ = (int)sizeof(T);
  if (MODE == CONV_FWD || MODE == CONV_BWD_DATA) {
    const int j = (V1 || V4) ? n0 + 4 * (tid % (BN / 4)) : n0 + tid % BN;
    bj_valid = j < Ngemm;
    uint32_t n, pix, y, x;
    if (V1) {
      a.div_hw.divmod(j, n, pix);
      const int ch = MODE == CONV_FWD ? g.C : g.K;
      const int chg = MODE == CONV_FWD ? a.Cg : a.Kg;
      b_base = (((int)n - n_base) * ch + grp * chg) * a.HW + (int)pix;
    } else if (MODE == CONV_FWD) {
      a.div_pq.divmod(j, n, pix);
      a.div_q.divmod(pix, y, x);
      b_y = y * g.stride_h - g.pad_h;
      b_x = x * g.stride_w - g.pad_w;
      b_base = (((int)n - n_base) * g.C + grp * a.Cg) * a.HW + b_y * g.W + b_x;
    } else {
      bc.div_hcwc.divmod(j, n, pix);
      bc.div_wc.divmod(pix, y, x);
      b_y = y + bc.off_h;
      b_x = x + bc.off_w;
      b_base = (((int)n - n_base) * g.K + grp * a.Kg) * a.PQ + b_y * g.Q + b_x;
    }
  } else {
    if (a.tap_major) {
      // every column of the tile is (tap, c0 + ...): one tap, whole channels
      const uint32_t tap = a.div_cg.div(n0);
      const int c0 = n0 - (int)tap * a.Cg;
      uint32_t r, s;
      a.div_s.divmod(tap, r, s);
      tm_r = (int)r - g.pad_h;
      tm_s = (int)s - g.pad_w;
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int c = c0 + tid / BK + (256 / BK) * i;
        // byte offset of the channel plane; tiles never run past Cg here
        bw_off[i] = (grp * a.Cg + c) * a.HW * SZ_;
        bw_r[i] = bw_s[i] = 0;
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
        bw_r[i] = bw_s[i] = -(1 << 29);
      }
    }
    }
  }

  // Scalar-path elements of one wave share their reduction index: element i
  // of wave-row u sits at kk = k0 + u + step * i.  Once per K tile lane l
  // decomposes kk = k0 + l into filter tap and byte offset; elements then
  // fetch their entry with v_readlane, so the per-element cost is a handful
  // of vector instructions and no integer division.
  const int ub = __builtin_amdgcn_readfirstlane(tid / BN);
  const int ua = __builtin_amdgcn_readfirstlane(tid / BM);
  constexpr int kFar = 1 << 28;  // fails every bounds test
  if (!bj_valid) b_y = kFar;
  const int b_base_bytes = b_base * SZ;

  // Filters of up to 63 taps: bit t of inv_mask tells whether tap t falls
  // outside the image for this thread's pixel (bit 63 is always set and
  // stands for "beyond the reduction range").  The per-element bounds test
  // is then one 64-bit shift whose bit 0 is moved into the sign of the offset.
  uint64_t inv_mask = ~0ull;
  if (MODE == CONV_FWD && !V1 && !V4 && a.tap_mask) {
    for (int r = 0; r < g.R; ++r)
      for (int s2 = 0; s2 < g.S; ++s2)
        if ((unsigned)(b_y + r) < (unsigned)g.H &&
            (unsigned)(b_x + s2) < (unsigned)g.W)
          inv_mask &= ~(1ull << (r * g.S + s2));
  }
  if (MODE == CONV_BWD_DATA && !V1 && !V4 && a.tap_mask) {
    for (int r = 0; r < bc.Rc; ++r)
      for (int s2 = 0; s2 < bc.Sc; ++s2)
        if ((unsigned)(b_y - r) < (unsigned)g.P &&
            (unsigned)(b_x - s2) < (unsigned)g.Q)
          inv_mask &= ~(1ull << (r * bc.Sc + s2));
  }

  float ra[A_PER], rb[B_PER];
  int ao[A_PER], bo[B_PER];
  int bshift[B_PER / 4 > 0 ? B_PER / 4 : 1];  // V4: -1 / 0 / +1 per load

  auto store_a_rows_vec = [&](float *as) {
    constexpr int AKV = BK / 4, AROWS = 256 / AKV;
    const int kv = tid % AKV, mrow = tid / AKV;
#pragma unroll
    for (int i = 0; i < A_PER / 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        as[(4 * kv + e) * LDA + mrow + AROWS * i] = ra[4 * i + e];
  };

  // phase 1: byte offsets of every element of the next tile
  auto plan_tiles = [&](int k0) {
    int nb = 0;
    if (MODE == CONV_BWD_WEIGHT) {
      // all images of this K tile lie at or after image nb
      nb = a.div_pq.div(k0);
      rA = make_rsrc(pa + nb * a_img, (a.a_elems - nb * a_img) * SZ);
      rB = make_rsrc(pb + nb * b_img, (a.b_elems - nb * b_img) * SZ);
    }
    // ---- A ----
    if ((MODE == CONV_FWD || MODE == CONV_BWD_WEIGHT) && a.a_vec) {
      // A[m][kk] contiguous along kk (forward: W, backward-weight: dY)
      constexpr int AKV = BK / 4, AROWS = 256 / AKV;
      const int kv = tid % AKV, mrow = tid / AKV;
      const int kk = k0 + 4 * kv;
      int base;
      if (MODE == CONV_FWD) {
        base = grp * a.Kg * a.Kgemm + kk;
      } else {
        uint32_t n, pq;
        a.div_pq.divmod(kk, n, pq);
        base = (((int)n - nb) * g.K + grp * a.Kg) * a.PQ + (int)pq;
      }
      const int mstride = MODE == CONV_FWD ? a.Kgemm : a.PQ;
#pragma unroll
      for (int i = 0; i < A_PER / 4; ++i) {
        const int m = m0 + mrow + AROWS * i;
        ao[i] = (m < a.M && kk < kend) ? (base + m * mstride) * SZ : kOob;
      }
    } else if (MODE == CONV_FWD || MODE == CONV_BWD_WEIGHT) {
      const int kk = k0 + tid % BK;
      int base, mstride;
      if (MODE == CONV_FWD) {
        base = grp * a.Kg * a.Kgemm + kk;
        mstride = a.Kgemm;
      } else {
        uint32_t n, pq;
        a.div_pq.divmod(kk, n, pq);
        base = (((int)n - nb) * g.K + grp * a.Kg) * a.PQ + (int)pq;
        mstride = a.PQ;
      }
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + tid / BK + (256 / BK) * i;
        ao[i] = (m < a.M && kk < kend) ? (base + m * mstride) * SZ : kOob;
      }
    } else if (V1) {
      // backward-data, A[m][kk] = W[kk][m], contiguous along m
      constexpr int AMV = BM / 4, AROWS = 256 / AMV;
      const int m = m0 + 4 * (tid % AMV), krow = tid / AMV;
#pragma unroll
      for (int i = 0; i < A_PER / 4; ++i) {
        const int kk = k0 + krow + AROWS * i;
        ao[i] = (m < a.M && kk < kend)
                    ? ((grp * a.Kg + kk) * a.Cg + m) * SZ : kOob;
      }
    }
    // ---- B (and the gathered A of backward-data) ----
    if (MODE != CONV_BWD_WEIGHT && V1) {
      constexpr int BV = BN / 4, BROWS = 256 / BV;
      const int krow = tid / BV;
#pragma unroll
      for (int i = 0; i < B_PER / 4; ++i) {
        const int kk = k0 + krow + BROWS * i;
        bo[i] = (bj_valid && kk < kend) ? (b_base + kk * a.HW) * SZ : kOob;
      }
    } else if (MODE != CONV_BWD_WEIGHT && V4) {
      constexpr int BV = BN / 4, BROWS = 256 / BV;
      const int krow = tid / BV;
      // forward walks the input forwards from (b_y, b_x), backward-data
      // walks the output gradient backwards
      const int rows = MODE == CONV_FWD ? g.H : g.P;
      const int cols = MODE == CONV_FWD ? g.W : g.Q;
      const int plane = MODE == CONV_FWD ? a.HW : a.PQ;
      // the A operand of backward-data is gathered as in the scalar path
      if (MODE == CONV_BWD_DATA) {
        const int kk = k0 + lane;
        uint32_t k, rs, rr, ss;
        bc.div_rcsc.divmod(kk, k, rs);
        bc.div_sc.divmod(rs, rr, ss);
        const int t_a = ((grp * a.Kg + (int)k) * a.Cg * a.RS +
                         (bc.r0 + g.stride_h * (int)rr) * g.S + bc.s0 +
                         g.stride_w * (int)ss) * SZ;
        const int m = m0 + tid % BM;
        const int moff = m * a.RS * SZ;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          const int src = ua + (256 / BM) * i;
          ao[i] = (m < a.M && k0 + src < kend)
                      ? moff + __builtin_amdgcn_readlane(t_a, src) : kOob;
        }
      }
#pragma unroll
      for (int i = 0; i < B_PER / 4; ++i) {
        const int kk = k0 + krow + BROWS * i;
        uint32_t c, rs, r, s;
        if (MODE == CONV_FWD) {
          a.div_rs.divmod(kk, c, rs);
          a.div_s.divmod(rs, r, s);
        } else {
          bc.div_rcsc.divmod(kk, c, rs);
          bc.div_sc.divmod(rs, r, s);
        }
        const int dy = MODE == CONV_FWD ? (int)r : -(int)r;
        const int dx = MODE == CONV_FWD ? (int)s : -(int)s;
        const int x0 = b_x + dx;
        const int sh = x0 < 0 ? 1 : (x0 + 3 >= cols ? -1 : 0);
        const bool ok = kk < kend && (unsigned)(b_y + dy) < (unsigned)rows;
        bshift[i] = sh;
        bo[i] = ok ? (b_base + (int)c * plane + dy * cols + dx + sh) * SZ
                   : kOob;
      }
    } else if (MODE == CONV_FWD) {
      const int kk = k0 + lane;
      uint32_t c, rs, r, s;
      a.div_rs.divmod(kk, c, rs);
      a.div_s.divmod(rs, r, s);
      const int t_off = ((int)c * a.HW + (int)r * g.W + (int)s) * SZ;
      const int t_r = kk < kend ? (int)r : kFar, t_s = (int)s;
      if (a.tap_mask) {
        const int t_rs = kk < kend ? (int)rs : 63;
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
          const int src = ub + (256 / BN) * i;
          const uint64_t out = inv_mask >> __builtin_amdgcn_readlane(t_rs, src);
          bo[i] = (b_base_bytes + __builtin_amdgcn_readlane(t_off, src)) |
                  (int)((uint32_t)out << 31);
        }
      } else
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int src = ub + (256 / BN) * i;
        const int ih = b_y + __builtin_amdgcn_readlane(t_r, src);
        const int iw = b_x + __builtin_amdgcn_readlane(t_s, src);
        const bool ok = (unsigned)ih < (unsigned)g.H &&
                        (unsigned)iw < (unsigned)g.W;
        bo[i] = ok ? b_base_bytes + __builtin_amdgcn_readlane(t_off, src)
                   : kOob;
      }
    } else if (MODE == CONV_BWD_DATA) {
      const int kk = k0 + lane;
      uint32_t k, rs, rr, ss;
      bc.div_rcsc.divmod(kk, k, rs);
      bc.div_sc.divmod(rs, rr, ss);
      const int t_off = ((int)k * a.PQ - (int)rr * g.Q - (int)ss) * SZ;
      const int t_r = kk < kend ? -(int)rr : kFar, t_s = -(int)ss;
      const int t_a = ((grp * a.Kg + (int)k) * a.Cg * a.RS +
                       (bc.r0 + g.stride_h * (int)rr) * g.S + bc.s0 +
                       g.stride_w * (int)ss) * SZ;
      const int m = m0 + tid % BM;
      const int moff = m * a.RS * SZ;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int src = ua + (256 / BM) * i;
        ao[i] = (m < a.M && k0 + src < kend)
                    ? moff + __builtin_amdgcn_readlane(t_a, src) : kOob;
      }
      if (a.tap_mask) {
        const int t_rs = kk < kend ? (int)rs : 63;
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
          const int src = ub + (256 / BN) * i;
          const uint64_t out = inv_mask >> __builtin_amdgcn_readlane(t_rs, src);
          bo[i] = (b_base_bytes + __builtin_amdgcn_readlane(t_off, src)) |
                  (int)((uint32_t)out << 31);
        }
      } else
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int src = ub + (256 / BN) * i;
        const int py = b_y + __builtin_amdgcn_readlane(t_r, src);
        const int px = b_x + __builtin_amdgcn_readlane(t_s, src);
        const bool ok = (unsigned)py < (unsigned)g.P &&
                        (unsigned)px < (unsigned)g.Q;
        bo[i] = ok ? b_base_bytes + __builtin_amdgcn_readlane(t_off, src)
                   : kOob;
      }
    } else {
      // backward-weight B: reduction index kk = (n, p, q), one per thread
      const int kk = k0 + tid % BK;
      uint32_t n, pq, p, q;
      a.div_pq.divmod(kk, n, pq);
      a.div_q.divmod(pq, p, q);
      const int ih0 = kk < kend ? (int)p * g.stride_h : kFar;
      const int iw0 = q * g.stride_w;
      if (a.tap_major) {
        // one bounds test for all of the thread's columns; a rejected pixel
        // starts from an offset beyond every resource and the (positive,
        // < 2^31) channel offsets keep it there
        const int ih = ih0 + tm_r, iw = iw0 + tm_s;
        const bool ok = (unsigned)ih < (unsigned)g.H &&
                        (unsigned)iw < (unsigned)g.W;
        const unsigned pix = ok
            ? (unsigned)((((int)n - nb) * g.C * g.H + ih) * g.W + iw) * SZ
            : 0x80000000u;
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
          bo[i] = (int)(pix + (unsigned)bw_off[i]);
      } else {
      const int base = ((int)n - nb) * g.C * a.HW + (int)p * g.stride_h * g.W +
                       iw0;
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int ih = ih0 + bw_r[i], iw = iw0 + bw_s[i];
        const bool ok = (unsigned)ih < (unsigned)g.H &&
                        (unsigned)iw < (unsigned)g.W;
        bo[i] = ok ? (base + bw_off[i]) * SZ : kOob;
      }
      }
    }
  };

  // phase 2: all loads of the tile back to back
  auto issue_loads = [&]() {
    constexpr bool A4 = V1 && MODE == CONV_BWD_DATA;
    if (A4 || (MODE != CONV_BWD_DATA && a.a_vec)) {
#pragma unroll
      for (int i = 0; i < A_PER / 4; ++i) BufLd<T>::four(rA, ao[i], &ra[4 * i]);
    } else {
#pragma unroll
      for (int i = 0; i < A_PER; ++i) ra[i] = BufLd<T>::one(rA, ao[i]);
    }
    if (MODE != CONV_BWD_WEIGHT && (V1 || V4)) {
#pragma unroll
      for (int i = 0; i < B_PER / 4; ++i) BufLd<T>::four(rB, bo[i], &rb[4 * i]);
    } else {
#pragma unroll
      for (int i = 0; i < B_PER; ++i) rb[i] = BufLd<T>::one(rB, bo[i]);
    }
  };

  auto load_tiles = [&](int k0) {
    plan_tiles(k0);
    __builtin_amdgcn_sched_barrier(0);
    issue_loads();
    __builtin_amdgcn_sched_barrier(0);
  };

  auto store_tiles = [&](int buf) {
    float *as = As[buf], *bs = Bs[buf];
    if (MODE == CONV_FWD || MODE == CONV_BWD_WEIGHT) {
      if (a.a_vec) {
        store_a_rows_vec(as);
      } else {
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
          as[(tid % BK) * LDA + tid / BK + (256 / BK) * i] = ra[i];
      }
    } else if (V1) {
      constexpr int AMV = BM / 4, AROWS = 256 / AMV;
#pragma unroll
      for (int i = 0; i < A_PER / 4; ++i)
        *reinterpret_cast<float4 *>(
            &as[(tid / AMV + AROWS * i) * LDA + 4 * (tid % AMV)]) =
            make_float4(ra[4 * i], ra[4 * i + 1], ra[4 * i + 2], ra[4 * i + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < A_PER; ++i)
        as[(tid / BM + (256 / BM) * i) * LDA + tid % BM] = ra[i];
    }
    if (MODE == CONV_BWD_WEIGHT) {
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        bs[(tid % BK) * LDB + tid / BK + (256 / BK) * i] = rb[i];
    } else if (V1) {
      constexpr int BV = BN / 4, BROWS = 256 / BV;
#pragma unroll
      for (int i = 0; i < B_PER / 4; ++i)
        *reinterpret_cast<float4 *>(
            &bs[(tid / BV + BROWS * i) * LDB + 4 * (tid % BV)]) =
            make_float4(rb[4 * i], rb[4 * i + 1], rb[4 * i + 2], rb[4 * i + 3]);
    } else if (V4) {
      constexpr int BV = BN / 4, BROWS = 256 / BV;
#pragma unroll
      for (int i = 0; i < B_PER / 4; ++i) {
        const float v0 = rb[4 * i], v1 = rb[4 * i + 1], v2 = rb[4 * i + 2],
                    v3 = rb[4 * i + 3];
        const int sh = bshift[i];
        // sh = +1: the group was loaded one pixel to the right of where it
        // belongs (its first pixel is padding); sh = -1: one to the left
        const float o0 = sh > 0 ? 0.f : (sh < 0 ? v1 : v0);
        const float o1 = sh > 0 ? v0 : (sh < 0 ? v2 : v1);
        const float o2 = sh > 0 ? v1 : (sh < 0 ? v3 : v2);
        const float o3 = sh > 0 ? v2 : (sh < 0 ? 0.f : v3);
        *reinterpret_cast<float4 *>(
            &bs[(tid / BV + BROWS * i) * LDB + 4 * (tid % BV)]) =
            make_float4(o0, o1, o2, o3);
      }
    } else {
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        bs[(tid / BN + (256 / BN) * i) * LDB + tid % BN] = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // double-buffered LDS: tile t+1 is fetched into registers under the MFMAs
  // of tile t and written to the other buffer, one barrier per tile
  const int nt = (kend - kbeg + BK - 1) / BK;
  PROF_STAMP(pf_t1);
  if (nt > 0) {
    load_tiles(kbeg);
    store_tiles(0);
  }
  __syncthreads();
  PROF_STAMP(pf_t2);
  const int lrow = lane & 31, lk = lane >> 5;
  for (int t = 0; t < nt; ++t) {
    const float *as = As[t & 1], *bs = Bs[t & 1];
    PROF_STAMP(pf_s0);
    const bool more = t + 1 < nt;
    if (more) load_tiles(kbeg + (t + 1) * BK);
    PROF_STAMP(pf_s1);
    if constexpr (std::is_same<T, float>::value) {
      // operands of k-step s+1 are fetched from LDS before the MFMAs of
      // k-step s are issued: a wave alone on its SIMD keeps the matrix pipe
      // busy instead of waiting for each pair of reads
      float av[2][TM], bv[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[0][i] = as[lk * LDA + wm * WM + 32 * i + lrow];
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bv[0][j] = bs[lk * LDB + wn * WN + 32 * j + lrow];
#pragma unroll
      for (int ks = 0; ks < BK; ks += 2) {
        const int cur = (ks >> 1) & 1, nxt = cur ^ 1;
        if (ks + 2 < BK) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            av[nxt][i] = as[(ks + 2 + lk) * LDA + wm * WM + 32 * i + lrow];
#pragma unroll
          for (int j = 0; j < TN; ++j)
            bv[nxt][j] = bs[(ks + 2 + lk) * LDB + wn * WN + 32 * j + lrow];
        }
        // keep the issue order: next step's reads, then this step's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // 16-bit inputs: 32x32x16 MFMA, lane (row, half) holds reduction
      // indices 8 * half .. 8 * half + 7 of its row
#pragma unroll
      for (int ks = 0; ks < BK; ks += 16) {
        typename Frag16<T>::V av[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          av[i] = frag16<T>(
              &as[(ks + 8 * lk) * LDA + wm * WM + 32 * i + lrow], LDA);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bv[j] = frag16<T>(
              &bs[(ks + 8 * lk) * LDB + wn * WN + 32 * j + lrow], LDB);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = Frag16<T>::mma(av[i], bv[j], acc[i][j]);
      }
    }
    PROF_STAMP(pf_s2);
    if (more) store_tiles((t + 1) & 1);
    PROF_STAMP(pf_s3);
    __syncthreads();
    PROF_STAMP(pf_s4);
    PROF_ADD(pf_plan, pf_s0, pf_s1);
    PROF_ADD(pf_mma, pf_s1, pf_s2);
    PROF_ADD(pf_store, pf_s2, pf_s3);
    PROF_ADD(pf_bar, pf_s3, pf_s4);
  }
  PROF_STAMP(pf_t3);

  // ---- epilogue ---------------------------------------------------------------
  // lane holds GEMM column (lane & 31); register r holds GEMM row
  // (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) of its 32 x 32 block
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + 32 * j + (lane & 31);
    if (col >= Ngemm) continue;
    size_t col_off;
    int chan_stride;
    if (MODE == CONV_FWD) {
      uint32_t n, pix;
      a.div_pq.divmod(col, n, pix);
      col_off = (size_t)n * g.K * a.PQ + pix;
      chan_stride = a.PQ;
    } else if (MODE == CONV_BWD_DATA) {
      uint32_t n, pix, hh, ww;
      bc.div_hcwc.divmod(col, n, pix);
      bc.div_wc.divmod(pix, hh, ww);
      col_off = (size_t)n * g.C * a.HW +
                (bc.ch + g.stride_h * (int)hh) * g.W + bc.cw +
                g.stride_w * (int)ww;
      chan_stride = a.HW;
    } else if (a.tap_major) {
      const uint32_t tap = a.div_cg.div(col);
      const int c = col - (int)tap * a.Cg;
      if (a.tap_scratch) {
        // scratch[tap][k][c]
        col_off = (size_t)tap * g.K * a.Cg + c;
        chan_stride = a.Cg;
      } else {
        // column (tap, c) of row k is dw[k][c][tap]
        col_off = (size_t)c * a.RS + tap;
        chan_stride = a.Ngemm;
      }
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
          const float bv = (a.bias && split == 0)
                               ? to_f<T>(((const T *)a.bias)[ch]) : 0.f;
          const size_t off = col_off + (size_t)ch * chan_stride;
          // split reductions accumulate in a float buffer whatever T is
          if (a.splits == 1) ((T *)a.out)[off] = from_f<T>(v + bv);
          else if (a.slabs) ((float *)a.out)[split * a.slab + off] = v + bv;
          else atomicAdd((float *)a.out + off, v + bv);
        } else if (MODE == CONV_BWD_DATA) {
          const int ch = grp * a.Cg + row;
          const size_t off = col_off + (size_t)ch * chan_stride;
          if (a.splits == 1) {
            const float add =
                a.addend ? to_f<T>(((const T *)a.addend)[off]) : 0.f;
            ((T *)a.out)[off] = from_f<T>(v + add);
          } else if (a.slabs) {
            ((float *)a.out)[split * a.slab + off] = v;
          } else {
            atomicAdd((float *)a.out + off, v);
          }
        } else {
          float *dst = (float *)a.out +
                       (size_t)(grp * a.Kg + row) * chan_stride + col_off;
          if (a.splits == 1) *dst = v;
          else if (a.slabs) dst[split * a.slab] = v;
          else atomicAdd(dst, v);
        }
      }
  }
  // ---- batch-norm statistics of the tile just written -----------------------
  if (MODE == CONV_FWD) {
    if (a.g.stats != nullptr && a.splits == 1) {
      const int Ktot = g.K;
      float *stats = a.g.stats + (size_t)(blockIdx.x & (kStatCopies - 1)) * 2 * Ktot;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float sm[16], sq[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row =
              m0 + wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int ch = grp * a.Kg + min(row, a.M - 1);
          const float bv = a.bias ? to_f<T>(((const T *)a.bias)[ch]) : 0.f;
          const float sh = a.g.stats_shift[ch];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + 32 * j + (lane & 31);
            // the value as the output tensor holds it
            const float y = to_f<T>(from_f<T>(acc[i][j][r] + bv)) - sh;
            if (col < Ngemm) {
              s1 += y;
              s2 += y * y;
            }
          }
          sm[r] = s1;
          sq[r] = s2;
        }
        half_wave_sums16(sm, lane);
        half_wave_sums16(sq, lane);
        const int idx = (lane >> 1) & 15;
        const int row =
            m0 + wm * WM + 32 * i + (idx & 3) + 8 * (idx >> 2) + 4 * (lane >> 5);
        if ((lane & 1) == 0 && row < a.M) {
          const int ch = grp * a.Kg + row;
          atomicAdd(stats + ch, sm[0]);
          atomicAdd(stats + Ktot + ch, sq[0]);
        }
      }
    }
  }
#ifdef SA_CONV_PROF
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PROF_STAMP(pf_t4);
    if (tid == 0 && g.prof) {
      // one private record per workgroup (slot 0 counts the workgroups)
      const unsigned long long slot = atomicAdd(g.prof, 1ull);
      unsigned long long *p = g.prof + 8 + 8 * slot;
      p[0] = pf_t1 - pf_t0;
      p[1] = pf_t2 - pf_t1;
      p[2] = pf_plan;
      p[3] = pf_mma;
      p[4] = pf_store;
      p[5] = pf_bar;
      p[6] = pf_t4 - pf_t3;
      p[7] = pf_t0;
    }
  }
#endif
}

// split partial sums combined in float (+ addend) -> output dtype
template <typename T>
__global__ void cast_out_kernel(const float *__restrict__ in,
                                const T *__restrict__ addend,
                                T *__restrict__ out, int64_t n) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    out[i] = from_f<T>(in[i] + (addend ? to_f<T>(addend[i]) : 0.f));
}

// out[i] = addend[i] + slab 0 + slab 1 + ... in that order (deterministic mode)
template <typename T>
__global__ void slab_sum_kernel(const float *__restrict__ slabs, int splits,
                                int64_t n, const T *__restrict__ addend,
                                T *__restrict__ out) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step) {
    float v = addend ? to_f<T>(addend[i]) : 0.f;
    for (int s = 0; s < splits; ++s) v += slabs[(int64_t)s * n + i];
    out[i] = from_f<T>(v);
  }
}

// dw[k][c][tap] = scratch[tap][k][c], and the scratch is left holding zeros
// for the next weight-gradient launch that borrows it.  One thread per (k, c).
__global__ void untap_kernel(float *__restrict__ scratch,
                             float *__restrict__ dw, int64_t kc, int taps) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < kc;
       i += step) {
    float *dst = dw + i * taps;
    for (int t = 0; t < taps; ++t) {
      float *src = scratch + (int64_t)t * kc + i;
      dst[t] = *src;
      *src = 0.f;
    }
  }
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
  a.addend = nullptr;
  a.mt = a.nt = 1;
  a.ncls = 1;
  a.tap_mask = a.RS <= 63;
  a.tap_major = 0;
  a.tap_scratch = 0;
  a.stagger = 0;
  a.slabs = 0;
  a.slab = 0;
  a.div_cg = FastDiv(a.Cg);
  if (mode == CONV_FWD) {
    a.M = a.Kg; a.Ngemm = g.N * a.PQ; a.Kgemm = a.Cg * a.RS;
    a.a_vec = a.Kgemm % 4 == 0;
  } else if (mode == CONV_BWD_DATA) {
    a.M = a.Cg; a.Ngemm = g.N * a.HW; a.Kgemm = a.Kg * a.RS;
    a.a_vec = 0;
  } else {
    a.M = a.Kg; a.Ngemm = a.Cg * a.RS; a.Kgemm = g.N * a.PQ;
    a.a_vec = a.PQ % 4 == 0;  // splits start at multiples of 32
  }
  a.div_rs = FastDiv(a.RS);
  a.div_s = FastDiv(g.S);
  a.div_pq = FastDiv(a.PQ);
  a.div_q = FastDiv(g.Q);
  a.div_hw = FastDiv(a.HW);
  a.div_w = FastDiv(g.W);
  return a;
}

static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// Tuning knobs (read once): SA_CONV_SPLIT_TARGET = number of workgroups below
// which the reduction is split across workgroups, SA_CONV_TILE = 64 / 128
// forces a tile size.
static const int kSplitTarget = env_int("SA_CONV_SPLIT_TARGET", 1024);
static const int kForceTile = env_int("SA_CONV_TILE", 0);
static const int kStagger = env_int("SA_CONV_STAGGER", 1);
// SA_CONV_TAP_MAJOR=0: weight-gradient columns stay in (channel, tap) order
static const int kTapMajor = env_int("SA_CONV_TAP_MAJOR", 1);
// SA_CONV_V4=0 keeps the one-pixel-per-load gather for every geometry
static const int kUseV4 = env_int("SA_CONV_V4", 1);

static inline bool use_big_tile(const ConvArgs &a) {
  const int force = a.g.tile ? a.g.tile : kForceTile;
  if (force == 64) return false;
  if (force == 128) return true;
  const int64_t t128 = (int64_t)((a.M + 127) / 128) * ((a.Ngemm + 127) / 128) *
                       a.g.groups * a.ncls;
  return a.M >= 96 && a.Ngemm >= 96 && t128 >= 256;
}

// Split the reduction so that the launch has close to, but not more than,
// kSplitTarget workgroups (what fits the chip at once: a second, nearly empty
// round of workgroups would cost as much as the first), each keeping at least
// min_k reduction indices; results are then combined with float atomics on a
// zero-filled output.
static void plan_splits(ConvArgs &a, int64_t tiles, int min_k) {
  int64_t splits = 1;
  const int target = a.g.split_target ? a.g.split_target : kSplitTarget;
  if (tiles < target) splits = target / tiles;
  const int64_t max_splits = (a.Kgemm + min_k - 1) / min_k;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int kps = (int)((a.Kgemm + splits - 1) / splits);
  kps = (kps + 31) / 32 * 32;
  a.k_per_split = kps;
  a.splits = (a.Kgemm + kps - 1) / kps;
}

template <typename T, int MODE, int BM, int BN, int BK, bool V1,
          bool V4 = false>
void launch_tile(ConvArgs &a, hipStream_t s) {
  a.mt = (a.M + BM - 1) / BM;
  a.nt = (a.Ngemm + BN - 1) / BN;
  dim3 grid;
  if (MODE == CONV_BWD_WEIGHT)
    grid = dim3((unsigned)(((int64_t)a.mt * a.nt * a.splits + 7) / 8 * 8), 1,
                a.g.groups);
  else
    grid = dim3((unsigned)(((int64_t)a.mt * a.nt + 7) / 8 * 8), a.splits,
                a.g.groups * a.ncls);
  hipLaunchKernelGGL((conv_igemm_kernel<T, MODE, BM, BN, BK, V1, V4>), grid,
                     dim3(256), 0, s, a);
}

// a.Ngemm / a.Kgemm hold the largest per-class extents for backward-data.
// out_elems: size of the output tensor; clear: this launch may initialise the
// whole output (to zero or to the addend) when it splits its reduction.
template <typename T, int MODE>
void launch_conv(ConvArgs &a, int64_t out_elems, bool clear, hipStream_t s,
                 bool may_split = true) {
  const ConvGeom &g = a.g;
  const bool big = use_big_tile(a);
  const int tb = big ? 128 : 64;
  const int64_t tiles = (int64_t)((a.M + tb - 1) / tb) *
                        ((a.Ngemm + tb - 1) / tb) * g.groups * a.ncls;
  // 16-bit outputs combine their partial sums in the caller's float scratch
  constexpr bool kFloatOut =
      std::is_same<T, float>::value || MODE == CONV_BWD_WEIGHT;
  const bool via_scratch = !kFloatOut && clear && g.scratch != nullptr &&
                           g.scratch_elems >= out_elems;
  a.stagger = kStagger;
  const bool det = deterministic();
  if (det) {
    // as many slabs as the buffer holds; none: no split
    const int64_t room = g.det_scratch ? g.det_elems / out_elems : 0;
    if (may_split && room >= 2) {
      plan_splits(a, tiles, 256);
      if (a.splits > room) {
        int kps = (int)((a.Kgemm + room - 1) / room);
        kps = (kps + 31) / 32 * 32;
        a.k_per_split = kps;
        a.splits = (a.Kgemm + kps - 1) / kps;
      }
    }
  } else if (may_split && (kFloatOut || via_scratch)) {
    plan_splits(a, tiles, 256);
  }
  void *const out = a.out;
  const void *const addend = a.addend;
  const bool slabbed = det && a.splits > 1;
  const bool staged = !slabbed && a.splits > 1 && !kFloatOut;
  // Backward-weight with (tap, channel) column order.  With several taps the
  // result is accumulated in the caller's zero-filled scratch in
  // [tap][K][Cg] order and moved into place afterwards; without a scratch
  // the columns keep the (channel, tap) order.
  bool untap = false;
  if (slabbed) {
    // a class whose reduction is shorter than the others leaves slabs
    // untouched: they must read as zero
    a.slabs = 1;
    a.slab = out_elems;
    a.out = g.det_scratch;
    a.addend = nullptr;
    zero_async(g.det_scratch, (size_t)a.splits * out_elems * sizeof(float), s);
  } else if (MODE == CONV_BWD_WEIGHT && kTapMajor && !det && a.Cg % tb == 0) {
    if (a.RS == 1) {
      a.tap_major = 1;
    } else if (g.scratch != nullptr && g.scratch_elems >= out_elems) {
      a.tap_major = a.tap_scratch = 1;
      a.out = g.scratch;
      untap = true;
    }
  }
  if (untap || slabbed) {
    // the scratch holds zeros; the output is overwritten as a whole afterwards
  } else if (staged) {
    a.out = g.scratch;
    zero_async(g.scratch, (size_t)out_elems * sizeof(float), s);
  } else if (a.splits > 1 && clear) {
    // partial sums land with atomics: start from the addend, or from zero
    const size_t bytes = (size_t)out_elems * sizeof(float);
    if (MODE == CONV_BWD_DATA && a.addend && a.addend != a.out)
      copy_async(a.out, a.addend, bytes, s);
    else if (!(MODE == CONV_BWD_WEIGHT && g.out_zeroed))
      zero_async(a.out, bytes, s);
  }
  bool v1 = false;
  if (MODE != CONV_BWD_WEIGHT)
    v1 = a.RS == 1 && g.stride_h == 1 && g.stride_w == 1 && g.pad_h == 0 &&
         g.pad_w == 0 && a.HW % 4 == 0 && a.Cg % 4 == 0 && a.Kg % 4 == 0;
  bool v4 = false;
  if constexpr (MODE != CONV_BWD_WEIGHT && std::is_same<T, float>::value) {
    const int cols = MODE == CONV_FWD ? g.W : g.Q;       // gathered rows
    const int out_cols = MODE == CONV_FWD ? g.Q : g.W;   // tile rows
    v4 = !v1 && kUseV4 && g.stride_h == 1 && g.stride_w == 1 &&
         out_cols % 4 == 0 && cols >= 4 && g.pad_w <= 1 && g.S <= 3 &&
         g.S - 1 - g.pad_w <= 1 && a.ncls == 1;
  }
  if (MODE != CONV_BWD_WEIGHT && v1) {
    if (big) launch_tile<T, MODE, 128, 128, 16, true>(a, s);
    else launch_tile<T, MODE, 64, 64, 32, true>(a, s);
  } else if (v4) {
    if constexpr (MODE != CONV_BWD_WEIGHT && std::is_same<T, float>::value) {
      if (big) launch_tile<T, MODE, 128, 128, 16, false, true>(a, s);
      else launch_tile<T, MODE, 64, 64, 32, false, true>(a, s);
    }
  } else {
    if (big) launch_tile<T, MODE, 128, 128, 16, false>(a, s);
    else launch_tile<T, MODE, 64, 64, 32, false>(a, s);
  }
  if (slabbed) {
    using O = typename std::conditional<MODE == CONV_BWD_WEIGHT, float, T>::type;
    hipLaunchKernelGGL((slab_sum_kernel<O>), dim3(grid_for(out_elems, 256, 4)),
                       dim3(256), 0, s, (const float *)g.det_scratch, a.splits,
                       out_elems, (const O *)addend, (O *)out);
    a.out = out;
    a.addend = addend;
  }
  if (untap) {
    const int64_t kc = out_elems / a.RS;
    hipLaunchKernelGGL(untap_kernel, dim3(grid_for(kc, 256)), dim3(256), 0, s,
                       g.scratch, (float *)out, kc, a.RS);
    a.out = out;
  }
  if (staged) {
    hipLaunchKernelGGL((cast_out_kernel<T>), dim3(grid_for(out_elems, 256, 4)),
                       dim3(256), 0, s, (const float *)g.scratch,
                       (const T *)addend, (T *)out, out_elems);
    a.out = out;
  }
}

}  // namespace sa

using namespace sa;

static bool geom_ok(const ConvGeom *g) {
  if (g->groups < 1 || g->C % g->groups || g->K % g->groups) return false;
  if (g->stride_h < 1 || g->stride_w < 1) return false;
  const int64_t lim = (int64_t)1 << 31;
  return (int64_t)g->N * g->C * g->H * g->W < lim &&
         (int64_t)g->N * g->K * g->P * g->Q < lim;
}

static size_t elem_size(int dtype) { return dtype == kF32 ? 4 : 2; }

// A tile addresses its operands with 32-bit byte offsets from the first image
// it touches: `span` consecutive images (and the whole filter) must fit 2 GB.
static bool window_ok(int64_t image_elems, int64_t span, int64_t filter_elems,
                      size_t sz) {
  const int64_t lim = ((int64_t)1 << 31) - 1;
  return image_elems * span * (int64_t)sz < lim &&
         filter_elems * (int64_t)sz < lim;
}

// Per-channel sums of a finished tensor y [N][K][PQ], added to copy 0 of the
// statistics rows: grid (K, S), block (k, s) walks the images s, s + S, ...
template <typename T>
__global__ void __launch_bounds__(256)
chan_sums_kernel(const T *__restrict__ y, float *__restrict__ stats,
                 const float *__restrict__ shift, int N, int K, int PQ) {
  __shared__ float smem[32];
  const int k = blockIdx.x;
  const float sh = shift[k];
  float s1 = 0.f, s2 = 0.f;
  for (int n = blockIdx.y; n < N; n += gridDim.y) {
    const T *p = y + ((size_t)n * K + k) * PQ;
    for (int i = threadIdx.x; i < PQ; i += blockDim.x) {
      const float v = to_f<T>(p[i]) - sh;
      s1 += v;
      s2 += v * v;
    }
  }
  s1 = block_sum(s1, smem);
  s2 = block_sum(s2, smem);
  if (threadIdx.x == 0) {
    atomicAdd(stats + k, s1);
    atomicAdd(stats + K + k, s2);
  }
}

int sa_chan_sums(int dtype, const void *y, float *stats, const float *shift,
                 int N, int K, int PQ, hipStream_t s) {
  if (N <= 0 || K <= 0 || PQ <= 0) return 0;
  int split = 8192 / K;
  split = std::max(1, std::min(split, N));
  const int threads = PQ >= 256 ? 256 : 64;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((chan_sums_kernel<T>), dim3(K, split), dim3(threads), 0,
                       s, (const T *)y, stats, shift, N, K, PQ);
  });
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int sa_conv_fwd(int dtype, const ConvGeom *g, const void *x, const void *w,
                const void *bias, void *y, hipStream_t s) {
  if (!geom_ok(g)) return -2;
  if (g->stats != nullptr && g->stats_shift == nullptr) return -2;
  if (sa_dwconv_applies(g)) {
    const int rc = sa_dwconv_fwd(dtype, g, x, w, bias, y, s);
    if (rc != 0 || g->stats == nullptr) return rc;
    return sa_chan_sums(dtype, y, g->stats, g->stats_shift, g->N, g->K,
                        g->P * g->Q, s);
  }
  if (sa_conv3x3_applies(dtype, g, g->C))
    return sa_conv3x3_fwd(g, x, w, bias, y, s);
  ConvArgs a = make_args(*g, CONV_FWD);
  a.a = w; a.b = x; a.out = y; a.bias = bias;
  a.a_elems = (int64_t)g->K * a.Cg * a.RS;
  a.b_elems = (int64_t)g->N * g->C * a.HW;
  if (!window_ok((int64_t)g->C * a.HW, 128 / a.PQ + 2, a.a_elems,
                 elem_size(dtype)))
    return -2;
  const int64_t elems = (int64_t)g->N * g->K * g->P * g->Q;
  SA_DISPATCH_FLOAT(dtype, T, { launch_conv<T, CONV_FWD>(a, elems, true, s); });
  // a split reduction has no workgroup that sees a finished value
  if (g->stats != nullptr && a.splits > 1)
    return sa_chan_sums(dtype, y, g->stats, g->stats_shift, g->N, g->K,
                        g->P * g->Q, s);
  return 0;
}

int sa_conv_bwd_data(int dtype, const ConvGeom *g, const void *dy,
                     const void *w, void *dx, hipStream_t s) {
  return sa_conv_bwd_data_acc(dtype, g, dy, w, nullptr, dx, s);
}

// dx = addend + conv_bwd_data(dy, w); addend (of dx's shape, distinct from dx)
// may be null.  While dx is still unwritten the kernels read the addend
// themselves; once dx has been initialised from it (classes no tap reaches,
// or partial sums combined with atomics) later launches add onto dx in place.
int sa_conv_bwd_data_acc(int dtype, const ConvGeom *g, const void *dy,
                         const void *w, const void *addend, void *dx,
                         hipStream_t s) {
  if (!geom_ok(g)) return -2;
  if (addend == dx) return -2;
  if (sa_dwconv_applies(g)) {
    if (g->side_zero) zero_async(g->side_zero, (size_t)g->side_zero_words * 4, s);
    return sa_dwconv_bwd_data(dtype, g, dy, w, addend, dx, s);
  }
  if (sa_conv3x3_applies(dtype, g, g->K))
    return sa_conv3x3_bwd_data(g, dy, w, addend, dx, s);
  ConvArgs base = make_args(*g, CONV_BWD_DATA);
  base.a = w; base.b = dy; base.out = dx;
  base.a_elems = (int64_t)g->K * base.Cg * base.RS;
  base.b_elems = (int64_t)g->N * g->K * base.PQ;
  {
    const int hc = std::max(1, g->H / g->stride_h);
    const int wc = std::max(1, g->W / g->stride_w);
    if (!window_ok((int64_t)g->K * base.PQ, 128 / (hc * wc) + 2, base.a_elems,
                   elem_size(dtype)))
      return -2;
  }
  const int64_t elems = (int64_t)g->N * g->C * g->H * g->W;
  const size_t bytes = (size_t)elems * elem_size(dtype);
  // one dense sub-problem per (h % sh, w % sw) class; classes no tap reaches
  // (filter smaller than the stride) stay zero
  std::vector<ConvArgs::Cls> classes;
  bool untouched = false;
  for (int ch = 0; ch < g->stride_h; ++ch)
    for (int cw = 0; cw < g->stride_w; ++cw) {
      ConvArgs::Cls c;
      c.ch = ch; c.cw = cw;
      c.Hc = ch < g->H ? (g->H - ch + g->stride_h - 1) / g->stride_h : 0;
      c.Wc = cw < g->W ? (g->W - cw + g->stride_w - 1) / g->stride_w : 0;
      if (c.Hc == 0 || c.Wc == 0) continue;
      c.r0 = (ch + g->pad_h) % g->stride_h;
      c.s0 = (cw + g->pad_w) % g->stride_w;
      c.Rc = c.r0 < g->R ? (g->R - c.r0 + g->stride_h - 1) / g->stride_h : 0;
      c.Sc = c.s0 < g->S ? (g->S - c.s0 + g->stride_w - 1) / g->stride_w : 0;
      if (c.Rc == 0 || c.Sc == 0) { untouched = true; continue; }
      c.off_h = (ch + g->pad_h - c.r0) / g->stride_h;
      c.off_w = (cw + g->pad_w - c.s0) / g->stride_w;
      c.Kgemm = base.Kg * c.Rc * c.Sc;
      c.Ngemm = g->N * c.Hc * c.Wc;
      c.div_rcsc = FastDiv(c.Rc * c.Sc);
      c.div_sc = FastDiv(c.Sc);
      c.div_hcwc = FastDiv(c.Hc * c.Wc);
      c.div_wc = FastDiv(c.Wc);
      classes.push_back(c);
    }
  bool zeroed = false;  // dx holds its starting value (zero or the addend)
  if (untouched || classes.empty()) {
    if (addend)
      copy_async(dx, addend, bytes, s);
    else
      zero_async(dx, bytes, s);
    zeroed = true;
  }
  if (classes.empty() && g->side_zero)
    zero_async(g->side_zero, (size_t)g->side_zero_words * 4, s);
  for (size_t first = 0; first < classes.size(); first += 4) {
    ConvArgs a = base;
    if (first > 0) a.g.side_zero = nullptr;   // the first launch clears it
    a.ncls = (int)std::min<size_t>(4, classes.size() - first);
    a.Ngemm = a.Kgemm = 0;
    for (int i = 0; i < a.ncls; ++i) {
      a.cls[i] = classes[first + i];
      a.Ngemm = std::max(a.Ngemm, a.cls[i].Ngemm);
      a.Kgemm = std::max(a.Kgemm, a.cls[i].Kgemm);
    }
    // an earlier batch may already have written its classes: only the first
    // launch may clear the output for split accumulation
    const bool clear = !(zeroed || first > 0);
    // 16-bit outputs split only when one launch produces the whole tensor
    const bool may_split =
        dtype == kF32 ? (zeroed || first == 0)
                      : (!zeroed && classes.size() <= 4);
    a.addend = !addend ? nullptr : (zeroed ? (const void *)dx : addend);
    SA_DISPATCH_FLOAT(dtype, T, {
      launch_conv<T, CONV_BWD_DATA>(a, elems, clear, s, may_split);
    });
    if (a.splits > 1) zeroed = true;
  }
  return 0;
}

int sa_conv_bwd_weight(int dtype, const ConvGeom *g, const void *dy,
                       const void *x, void *dw, hipStream_t s) {
  if (!geom_ok(g)) return -2;
  if (sa_dwconv_bwd_weight_applies(g))
    return sa_dwconv_bwd_weight(dtype, g, dy, x, dw, s);
  ConvArgs a = make_args(*g, CONV_BWD_WEIGHT);
  a.a = dy; a.b = x; a.out = dw;
  a.a_elems = (int64_t)g->N * g->K * a.PQ;
  a.b_elems = (int64_t)g->N * g->C * a.HW;
  if (!window_ok(std::max((int64_t)g->K * a.PQ, (int64_t)g->C * a.HW),
                 32 / a.PQ + 2, 0, elem_size(dtype)))
    return -2;
  const int64_t elems = (int64_t)g->K * a.Cg * a.RS;
  SA_DISPATCH_FLOAT(dtype, T, {
    launch_conv<T, CONV_BWD_WEIGHT>(a, elems, true, s);
  });
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

// ---------------------------------------------------------------------------
// Stride-2 sampling for 1x1 / stride 2 convolutions
// ---------------------------------------------------------------------------
// A 1x1 convolution with stride 2 only ever sees x[.., 2h, 2w]: gathering
// those into a dense quarter-size tensor turns forward, data gradient and
// weight gradient into stride-1 problems for the GEMM-shaped kernels.  The
// data gradient goes back through the inverse (values at even positions,
// zero elsewhere), fused with the gradient that reached x along another path.
namespace sa {

template <typename T>
__global__ void subsample2_kernel(const T *__restrict__ x, T *__restrict__ xs,
                                  int64_t planes, int H, int W, int P, int Q) {
  const int64_t total = planes * P * Q;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    const int q = (int)(i % Q);
    const int64_t t = i / Q;
    const int p = (int)(t % P);
    const int64_t plane = t / P;
    xs[i] = x[(plane * H + 2 * p) * W + 2 * q];
  }
}

// dx[.., h, w] = addend[.., h, w] + (h, w both even ? dxs[.., h/2, w/2] : 0);
// one thread per pair of columns (W even)
template <typename T>
__global__ void upsample2_add_kernel(const T *__restrict__ dxs,
                                     const T *__restrict__ addend,
                                     T *__restrict__ dx, int64_t planes, int H,
                                     int W, int P, int Q) {
  const int W2 = W / 2;
  const int64_t total = planes * H * W2;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += step) {
    const int w2 = (int)(i % W2);
    const int64_t t = i / W2;
    const int h = (int)(t % H);
    const int64_t plane = t / H;
    const int64_t at = (plane * H + h) * W + 2 * w2;
    float a = 0.f, b = 0.f;
    if (addend) {
      a = to_f<T>(addend[at]);
      b = to_f<T>(addend[at + 1]);
    }
    if ((h & 1) == 0) a += to_f<T>(dxs[(plane * P + (h >> 1)) * Q + w2]);
    dx[at] = from_f<T>(a);
    dx[at + 1] = from_f<T>(b);
  }
}

}  // namespace sa

int sa_subsample2(int dtype, const void *x, void *xs, int64_t planes, int H,
                  int W, hipStream_t s) {
  const int P = (H + 1) / 2, Q = (W + 1) / 2;
  const int64_t total = planes * P * Q;
  if (total <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((subsample2_kernel<T>), dim3(grid_for(total, 256)),
                       dim3(256), 0, s, (const T *)x, (T *)xs, planes, H, W, P,
                       Q);
  });
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int sa_upsample2_add(int dtype, const void *dxs, const void *addend, void *dx,
                     int64_t planes, int H, int W, hipStream_t s) {
  if (W % 2 != 0) return -3;
  const int P = (H + 1) / 2, Q = (W + 1) / 2;
  const int64_t total = planes * H * (W / 2);
  if (total <= 0) return 0;
  SA_DISPATCH_FLOAT(dtype, T, {
    hipLaunchKernelGGL((upsample2_add_kernel<T>), dim3(grid_for(total, 256)),
                       dim3(256), 0, s, (const T *)dxs, (const T *)addend,
                       (T *)dx, planes, H, W, P, Q);
  });
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
