// 3x3, stride 1, pad 1 float32 convolution (forward and backward-data) as a
// direct MFMA kernel over halo tiles in LDS.
//
// A workgroup owns 64 output channels x TH output rows of one image.  For each
// chunk of 8 input channels it copies the (TH + 2) input rows -- raw rows, read
// with 16-byte loads, no im2col address arithmetic -- into an LDS image with a
// one-pixel halo ([channel][row][W + 2], zero borders), next to the matching
// weight slice stored [channel][tap][out channel].  In that image the nine
// taps of an output position q = row * (W + 2) + col are the nine constant
// offsets r * (W + 2) + s, so the B operand of v_mfma_f32_32x32x2_f32 (lane =
// output position) is one ds_read_b32 at a compile-time offset per tap, and
// the A operand (lane = output channel) one ds_read_b32 from the weight slice:
// the inner loop is LDS reads and MFMAs only.  Positions that fall on the two
// border columns are computed and dropped (2 / (W + 2) of the work).
//
// The next chunk is fetched from global memory into registers while the
// current one is multiplied, and written to the other LDS buffer afterwards:
// one barrier per chunk (144 MFMAs per wave).
//
// Backward-data is the same convolution of dY with the filter transposed and
// flipped; only the weight staging differs.
//
// Takes the place of cudnnConvolutionForward / BackwardData for the most
// common filter shape (reference src/model/operation/convolution.cc:585-660).
#include <algorithm>
#include <cstdlib>

#include "common.cuh"
#include "kernels.h"

namespace sa {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct Halo3Args {
  const float *in, *w, *bias, *addend;
  float *out;
  int N, Cin, Cout, H, tiles_h;
  // forward: batch-norm statistics of the output (see ConvGeom::stats)
  float *stats;
  const float *shift;
  // backward-data: words the grid clears on the side (ConvGeom::side_zero)
  uint32_t *side_zero;
  int64_t side_zero_words;
};

constexpr int kH3Chunk = 8;     // input channels per LDS stage
constexpr int kH3KPitch = 68;   // floats per (channel, tap) row of the weights

template <int W, int TH, int NTW>
struct Halo3Shape {
  static constexpr int Wp = W + 2;
  static constexpr int NT = 4 * NTW;              // 32-position tiles
  static constexpr int QN = NT * 32;              // positions computed
  static constexpr int VEC = W % 4 == 0 ? 4 : 2;  // floats per global load
  static constexpr int RowUnits = W / VEC;
  static constexpr int ChanUnits = (TH + 2) * RowUnits;
  static constexpr int Units = kH3Chunk * ChanUnits;
  static constexpr int NU = (Units + 255) / 256;  // units per thread
  // floats per channel image: every computed position reads up to 2 rows + 2
  static constexpr int CP = (QN + 2 * Wp + 2 + 31) / 32 * 32;
  static constexpr int XS = kH3Chunk * CP;
  static constexpr int WS = kH3Chunk * 9 * kH3KPitch;
  static_assert(TH * Wp <= QN, "tile rows exceed the computed positions");
  static_assert((TH + 2) * Wp <= CP, "halo rows exceed the channel image");
  static_assert(W % VEC == 0, "row length must be a whole number of loads");
};

template <int VEC> struct H3Vec;
template <> struct H3Vec<4> {
  using type = float4;
  static __device__ __forceinline__ void put(float *p, const float4 &v) {
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
  }
  static __device__ __forceinline__ float4 zero() {
    return make_float4(0.f, 0.f, 0.f, 0.f);
  }
};
template <> struct H3Vec<2> {
  using type = float2;
  static __device__ __forceinline__ void put(float *p, const float2 &v) {
    p[0] = v.x; p[1] = v.y;
  }
  static __device__ __forceinline__ float2 zero() { return make_float2(0.f, 0.f); }
};

// MODE 0: forward (w is [Cout][Cin][3][3]); MODE 1: backward-data (w is the
// forward filter [Cin][Cout][3][3] of this launch's roles, used flipped).
template <int W, int TH, int NTW, int MODE>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(Halo3Args a) {
  using S = Halo3Shape<W, TH, NTW>;
  using V = H3Vec<S::VEC>;
  using vec_t = typename V::type;
  constexpr int Wp = S::Wp, CP = S::CP, KP = kH3KPitch, CC = kH3Chunk;
  __shared__ float Xs[2][S::XS];
  __shared__ float Ws[2][S::WS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n31 = lane & 31, kh = lane >> 5;
  if constexpr (MODE == 1) {
    if (a.side_zero) grid_zero_words(a.side_zero, a.side_zero_words);
  }
  const int per_m = a.tiles_h * a.N;
  const int mt = blockIdx.x / per_m;
  const int rem = blockIdx.x - mt * per_m;
  const int img = rem / a.tiles_h;
  const int h0 = (rem - img * a.tiles_h) * TH;
  const int m0 = mt * 64;
  const int H = a.H;

  // ---- staging plan of this thread (the same for every chunk) --------------------
  int x_src[S::NU], x_dst[S::NU];
  bool x_ok[S::NU];
#pragma unroll
  for (int i = 0; i < S::NU; ++i) {
    const int u = tid + 256 * i;
    const int c = u / S::ChanUnits;
    const int r2 = u - c * S::ChanUnits;
    const int row = r2 / S::RowUnits;
    const int v = r2 - row * S::RowUnits;
    const int h = h0 + row - 1;
    x_ok[i] = u < S::Units && h >= 0 && h < H;
    x_src[i] = (c * H + h) * W + S::VEC * v;
    x_dst[i] = u < S::Units ? c * CP + row * Wp + 1 + S::VEC * v : -1;
  }
  const float *in_img = a.in + (size_t)img * a.Cin * H * W;

  // zero both images once: border columns and out-of-image rows stay zero
  for (int i = tid; i < 2 * S::XS; i += 256) (&Xs[0][0])[i] = 0.f;
  __syncthreads();

  vec_t xr[S::NU];
  float wr[MODE == 0 ? 18 : 24];

  auto fetch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < S::NU; ++i)
      xr[i] = x_ok[i] ? *reinterpret_cast<const vec_t *>(
                            in_img + (size_t)c0 * H * W + x_src[i])
                      : V::zero();
    if constexpr (MODE == 0) {
      const int k = tid >> 2, part = tid & 3;
      const bool ok = m0 + k < a.Cout;
      const float *src = a.w + ((size_t)(m0 + k) * a.Cin + c0) * 9 + part * 18;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float2 t = ok ? *reinterpret_cast<const float2 *>(src + 2 * i)
                            : make_float2(0.f, 0.f);
        wr[2 * i] = t.x;
        wr[2 * i + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < CC; ++kk) {
        const float *src = a.w + ((size_t)(c0 + kk) * a.Cout + m0) * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int j = tid + 256 * i;
          wr[kk * 3 + i] = (j < 576 && m0 + j / 9 < a.Cout) ? src[j] : 0.f;
        }
      }
    }
  };

  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < S::NU; ++i)
      if (x_dst[i] >= 0) V::put(&Xs[buf][x_dst[i]], xr[i]);
    if constexpr (MODE == 0) {
      const int k = tid >> 2, part = tid & 3;
#pragma unroll
      for (int i = 0; i < 18; ++i) Ws[buf][(part * 18 + i) * KP + k] = wr[i];
    } else {
#pragma unroll
      for (int kk = 0; kk < CC; ++kk)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int j = tid + 256 * i;
          if (j < 576)
            Ws[buf][(kk * 9 + 8 - j % 9) * KP + j / 9] = wr[kk * 3 + i];
        }
    }
  };

  f32x16 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int chunks = a.Cin / CC;
  fetch(0);
  stage(0);
  __syncthreads();
  for (int ci = 0; ci < chunks; ++ci) {
    const int buf = ci & 1;
    if (ci + 1 < chunks) fetch((ci + 1) * CC);
    const float *xs = &Xs[buf][kh * CP + wave * (NTW * 32) + n31];
    const float *ws = &Ws[buf][kh * 9 * KP + n31];
#pragma unroll
    for (int k2 = 0; k2 < CC / 2; ++k2) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float a0 = ws[(k2 * 18 + tap) * KP];
        const float a1 = ws[(k2 * 18 + tap) * KP + 32];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const float b =
              xs[2 * k2 * CP + j * 32 + (tap / 3) * Wp + (tap % 3)];
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][j], 0,
                                                           0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][j], 0,
                                                           0, 0);
        }
      }
    }
    if (ci + 1 < chunks) stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane = position, register = output channel ----------------------------
  float *out_img = a.out + (size_t)img * a.Cout * H * W;
  const float *add_img =
      a.addend ? a.addend + (size_t)img * a.Cout * H * W : nullptr;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int q = (wave * NTW + j) * 32 + n31;
    const int ro = q / Wp, co = q - ro * Wp;
    const bool ok = co < W && ro < TH && h0 + ro < H;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (ok && m < a.Cout) {
          const size_t e = ((size_t)m * H + h0 + ro) * W + co;
          float v = acc[i][j][r];
          if (a.bias) v += a.bias[m];
          if (add_img) v += add_img[e];
          out_img[e] = v;
        }
      }
  }

  // ---- batch-norm statistics of the tile just written -----------------------
  // Per wave: 16 rows x 32 positions per block, summed over the lanes of a
  // half with the register-exchanging butterfly; the four waves (same
  // channels, other positions) meet in LDS and one lane per channel adds the
  // workgroup's total to the statistics rows.
  if (MODE == 0) {
    if (a.stats != nullptr) {
      float *red = &Ws[0][0];  // [4 waves][2 sums][64 channels]
      __syncthreads();         // every wave is done with the weight images
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float sm[16], sq[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const int mc = min(m, a.Cout - 1);
          const float off = (a.bias ? a.bias[mc] : 0.f) - a.shift[mc];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            const int q = (wave * NTW + j) * 32 + n31;
            const int ro = q / Wp, co = q - ro * Wp;
            const bool ok = co < W && ro < TH && h0 + ro < H;
            const float y = acc[i][j][r] + off;
            if (ok) {
              s1 += y;
              s2 += y * y;
            }
          }
          sm[r] = s1;
          sq[r] = s2;
        }
        half_wave_sums16(sm, lane);
        half_wave_sums16(sq, lane);
        if ((lane & 1) == 0) {
          const int idx = (lane >> 1) & 15;
          const int ml = i * 32 + (idx & 3) + 8 * (idx >> 2) + 4 * kh;
          red[(wave * 2 + 0) * 64 + ml] = sm[0];
          red[(wave * 2 + 1) * 64 + ml] = sq[0];
        }
      }
      __syncthreads();
      if (tid < 128) {
        const int which = tid >> 6, ml = tid & 63;
        const float t = red[(0 * 2 + which) * 64 + ml] +
                        red[(1 * 2 + which) * 64 + ml] +
                        red[(2 * 2 + which) * 64 + ml] +
                        red[(3 * 2 + which) * 64 + ml];
        if (m0 + ml < a.Cout)
          atomicAdd(a.stats +
                        ((size_t)(blockIdx.x & (kStatCopies - 1)) * 2 + which) *
                            a.Cout +
                        m0 + ml,
                    t);
      }
    }
  }
}

static int env_flag(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

template <int W, int TH, int NTW>
static void launch_halo3(const Halo3Args &a, int mode, hipStream_t s) {
  Halo3Args b = a;
  b.tiles_h = (a.H + TH - 1) / TH;
  const unsigned grid =
      (unsigned)((int64_t)b.tiles_h * a.N * ((a.Cout + 63) / 64));
  if (mode == 0)
    hipLaunchKernelGGL((conv3x3_halo_kernel<W, TH, NTW, 0>), dim3(grid),
                       dim3(256), 0, s, b);
  else
    hipLaunchKernelGGL((conv3x3_halo_kernel<W, TH, NTW, 1>), dim3(grid),
                       dim3(256), 0, s, b);
}

}  // namespace sa

using namespace sa;

// SA_CONV_HALO=0 keeps every geometry on the implicit-GEMM kernel.
static const int kUseHalo = env_flag("SA_CONV_HALO", 1);

bool sa_conv3x3_applies(int dtype, const ConvGeom *g, int reduce_channels) {
  if (!kUseHalo || dtype != kF32) return false;
  if (g->R != 3 || g->S != 3 || g->stride_h != 1 || g->stride_w != 1 ||
      g->pad_h != 1 || g->pad_w != 1 || g->groups != 1)
    return false;
  if (g->H != g->P || g->W != g->Q) return false;
  if (reduce_channels % kH3Chunk != 0) return false;
  if ((int64_t)g->N * std::max(g->C, g->K) * g->H * g->W >= ((int64_t)1 << 31))
    return false;
  return g->W == 56 || g->W == 28 || g->W == 14;
}

static int run_halo3(const ConvGeom *g, const Halo3Args &a, int mode,
                     hipStream_t s) {
  switch (g->W) {
    case 56: launch_halo3<56, 4, 2>(a, mode, s); break;
    case 28: launch_halo3<28, 4, 1>(a, mode, s); break;
    case 14: launch_halo3<14, 7, 1>(a, mode, s); break;
    default: return -3;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int sa_conv3x3_fwd(const ConvGeom *g, const void *x, const void *w,
                   const void *bias, void *y, hipStream_t s) {
  Halo3Args a{};
  a.in = (const float *)x; a.w = (const float *)w;
  a.bias = (const float *)bias; a.addend = nullptr; a.out = (float *)y;
  a.N = g->N; a.Cin = g->C; a.Cout = g->K; a.H = g->H;
  a.stats = g->stats; a.shift = g->stats_shift;
  return run_halo3(g, a, 0, s);
}

int sa_conv3x3_bwd_data(const ConvGeom *g, const void *dy, const void *w,
                        const void *addend, void *dx, hipStream_t s) {
  Halo3Args a{};
  a.in = (const float *)dy; a.w = (const float *)w; a.bias = nullptr;
  a.addend = (const float *)addend; a.out = (float *)dx;
  a.N = g->N; a.Cin = g->K; a.Cout = g->C; a.H = g->H;
  a.side_zero = g->side_zero; a.side_zero_words = g->side_zero_words;
  return run_halo3(g, a, 1, s);
}
