// Host-callable launchers of every HIP kernel.  All functions enqueue on the
// given stream, never allocate or synchronise (so they can be captured into a
// hipGraph) and return 0 on success, a negative code for unsupported
// arguments.  Pointers are raw device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sa {

// Up to 8 collapsed dimensions with three stride sets (operand 0, operand 1,
// output); strides are in elements, 0 = broadcast.
struct Desc {
  int ndim;
  int64_t shape[8];
  int64_t s0[8];
  int64_t s1[8];
  int64_t s2[8];
};

enum UnaryOp : int {
  U_COPY = 0, U_ABS, U_EXP, U_LOG, U_SQRT, U_SQUARE, U_SIGN, U_CEIL, U_FLOOR,
  U_ROUND, U_ROUNDE, U_SIGMOID, U_TANH, U_RELU, U_ERF, U_COS, U_COSH, U_ACOS,
  U_ACOSH, U_SIN, U_SINH, U_ASIN, U_ASINH, U_TAN, U_ATAN, U_ATANH, U_SOFTPLUS,
  U_SOFTSIGN, U_CLAMP, U_ADD_S, U_SUB_S, U_MUL_S, U_DIV_S, U_POW_S, U_RDIV_S,
  U_LT_S, U_LE_S, U_GT_S, U_GE_S, U_EQ_S, U_MAX_S, U_MIN_S
};

enum BinaryOp : int {
  B_ADD = 0, B_SUB, B_MUL, B_DIV, B_POW, B_LT, B_LE, B_GT, B_GE, B_EQ, B_MAX,
  B_MIN, B_RELU_BWD
};

enum ReduceOp : int { R_SUM = 0, R_MEAN, R_MAX, R_MIN, R_ASUM, R_SUMSQ };

}  // namespace sa

// ---- elementwise.hip --------------------------------------------------------
int sa_unary(int op, int dt_in, int dt_out, const void *x, void *y,
             const sa::Desc *d, int64_t n, float a, float b, hipStream_t s);
int sa_binary(int op, int dtype, const void *x, const void *y, void *z,
              const sa::Desc *d, int64_t n, hipStream_t s);
int sa_fill(int dtype, void *y, int64_t n, float v, hipStream_t s);
int sa_axpy(int dtype, const float *alpha_ptr, float alpha, const void *x,
            void *y, int64_t n, hipStream_t s);

// ---- reduce.hip ---------------------------------------------------------------
// out[r] = reduce_c x[r, c] for a row-major (rows, cols) matrix
int sa_reduce_rows(int op, int dtype, const void *x, void *out, int64_t rows,
                   int64_t cols, hipStream_t s);
// out[c] = sum_r x[r, c] for a row-major (rows, cols) matrix; out is float32
// when out_f32 != 0.  `work` must hold rows_split * cols floats (see
// sa_reduce_cols_workspace).
int sa_reduce_cols(int dtype, const void *x, void *out, int64_t rows,
                   int64_t cols, hipStream_t s);
// The same in ONE launch for any number of row slices, with the result in
// float (out_typed 0) or in x's type: partial sums meet in `sums` (floats,
// zero on entry and on exit; `tickets` likewise), the last slice to arrive
// writes the output.  The scratch serves one stream at a time.  -3: scratch
// too small.
int sa_reduce_cols_fin(int dtype, const void *x, void *out, int out_typed,
                       int64_t rows, int64_t cols, float *sums,
                       int64_t sums_cap, unsigned *tickets,
                       int64_t tickets_cap, hipStream_t s);
// whole-tensor reduction to one element (op: R_SUM, R_ASUM, R_SUMSQ, R_MAX,
// R_MIN); `out` is written in the tensor's dtype, zero-initialised by the
// launcher.
int sa_reduce_all(int op, int dtype, const void *x, void *out, int64_t n,
                  hipStream_t s);

// ---- softmax_loss.hip -----------------------------------------------------------
int sa_softmax_fwd(int dtype, const void *x, void *y, int64_t rows,
                   int64_t cols, hipStream_t s);
int sa_softmax_bwd(int dtype, const void *dy, const void *y, void *dx,
                   int64_t rows, int64_t cols, hipStream_t s);
// loss[r] = -log p[r, t[r]] (int labels) or -sum_c w[r,c] log p[r,c]
int sa_cross_entropy_fwd(int dtype, const void *p, const void *t, int t_is_int,
                         int t_dtype, void *loss, int64_t rows, int64_t cols,
                         hipStream_t s);
// in place: p -= onehot(t)  (or p -= t)
int sa_softmax_ce_bwd(int dtype, void *p, const void *t, int t_is_int,
                      int t_dtype, int64_t rows, int64_t cols, hipStream_t s);
// p = softmax(x); loss[0] = mean_r CE(p[r], t[r])
int sa_softmax_ce_fused_fwd(int dtype, const void *x, const void *t,
                            int t_is_int, int t_dtype, void *p, void *loss,
                            int64_t rows, int64_t cols, hipStream_t s);
// dx = (p - onehot(t)) / rows
int sa_softmax_ce_fused_bwd(int dtype, const void *p, const void *t,
                            int t_is_int, int t_dtype, void *dx, int64_t rows,
                            int64_t cols, hipStream_t s);

// ---- rng.hip --------------------------------------------------------------------------
// state = int64[2] {seed, offset}; the kernel draws from the current offset
// and a trailing single-thread kernel advances it, so replays of a captured
// graph continue the stream.
int sa_rng_uniform(int dtype, void *y, int64_t n, float lo, float hi,
                   int64_t *state, hipStream_t s);
int sa_rng_normal(int dtype, void *y, int64_t n, float mean, float std,
                  int64_t *state, hipStream_t s);
int sa_rng_bernoulli(int dtype, void *y, int64_t n, float p, int64_t *state,
                     hipStream_t s);

// ---- gemm.hip ---------------------------------------------------------------------------
// C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b]; A is (M, K) with element
// strides (a_rs, a_cs), B is (K, N) with (b_rs, b_cs), C row-major with ldc.
int sa_gemm(int dtype, int64_t M, int64_t N, int64_t K, float alpha,
            const void *A, int64_t a_rs, int64_t a_cs, int64_t a_bs,
            const void *B, int64_t b_rs, int64_t b_cs, int64_t b_bs,
            float beta, void *C, int64_t ldc, int64_t c_bs, int64_t batch,
            const void *bias, int act, hipStream_t s);

// 16-bit A / B, float partial products part[b] = A[b] * B[b]; -3 when the
// operand layout does not fit the tiled kernel
int sa_gemm_f32_slices(int64_t M, int64_t N, int64_t kc, int64_t k_total,
                       const void *A, int64_t a_rs, int64_t a_cs,
                       const void *B, int64_t b_rs, int64_t b_cs, float *part,
                       int64_t splits, hipStream_t s);
int sa_gemm_h16_partial(int dtype, int64_t M, int64_t N, int64_t K,
                        const void *A, int64_t a_rs, int64_t a_cs,
                        int64_t a_bs, const void *B, int64_t b_rs,
                        int64_t b_cs, int64_t b_bs, float *part, int64_t batch, int64_t k_total,
                        hipStream_t s);
// epilogue of a reduction split over `splits` partial products [splits][M][N]
int sa_gemm_splitk_reduce(int dtype, const float *part, int splits, int64_t M,
                          int64_t N, float alpha, float beta, void *C,
                          int64_t ldc, const void *bias, int act,
                          hipStream_t s);

// 16-bit 1x1 / stride 1 / no padding convolution as one tiled GEMM per image:
// mode 0 forward (act = x, out = y), mode 1 backward-data (act = dy, out = dx,
// optional addend of dx's shape); -3 = geometry not supported
int sa_conv1x1_h16(int dtype, int mode, int64_t n_img, int64_t C, int64_t K,
                   int64_t HW, const void *act, const void *w,
                   const void *addend, void *out, hipStream_t s);
// 16-bit 3x3 / stride 1 / pad 1 convolution as a nine-tap GEMM per image:
// mode 0 forward (act = x, out = y), mode 1 backward-data (act = dy, out = dx,
// optional addend of dx's shape).  wt: workspace of 9 * C * K elements.
// Returns -3 when the geometry does not fit.
int sa_conv3x3_h16(int dtype, int mode, int n_img, int C, int K, int H, int W,
                   const void *act, const void *w, void *wt,
                   const void *addend, void *out, hipStream_t s);
// 16-bit 1x1 (taps 1) or 3x3 / pad 1 (taps 9) stride-1 convolution over the
// flattened positions of all images (any image size), forward (mode 0) and
// backward-data (mode 1, optional addend of dx's shape).  wt: workspace of
// taps * C * K elements.  Returns -3 when the geometry does not fit.
int sa_conv_flat_h16(int dtype, int mode, int n_img, int C, int K, int H,
                     int W, int taps, const void *act, const void *w, void *wt,
                     const void *addend, void *out, hipStream_t s);
// 16-bit weight gradient of a 1x1 (taps 1) or 3x3 / pad 1 (taps 9) stride-1
// convolution on the tiled 16-bit kernel with a split reduction.  _splits
// returns the split count (0 = geometry not supported); part is workspace of
// splits * taps * K * C floats; dw is [K][C][taps] in the operands' type.
int sa_conv_wgrad_h16_splits(int n_img, int C, int K, int H, int W, int taps);
int sa_conv_wgrad_h16(int dtype, int n_img, int C, int K, int H, int W,
                      int taps, const void *dy, const void *x, float *part,
                      int splits, void *dw, hipStream_t s);

// device-to-device copy / zero-fill as kernels (4-byte aligned pointers)
int sa_copy_bytes(void *dst, const void *src, int64_t bytes, hipStream_t s);
int sa_zero_bytes(void *dst, int64_t bytes, hipStream_t s);

// ---- layernorm.hip ------------------------------------------------------------------------
// x, y, gamma, beta share dtype; gamma / beta may be null; mean / rstd: float[rows]
int sa_layernorm_fwd(int dtype, const void *x, const void *gamma,
                     const void *beta, void *y, float *mean, float *rstd,
                     int64_t rows, int D, float eps, hipStream_t s);
int sa_layernorm_bwd(int dtype, const void *dy, const void *x,
                     const void *gamma, const float *mean, const float *rstd,
                     void *dx, float *dgamma, float *dbeta, int64_t rows, int D,
                     hipStream_t s);
int sa_layernorm_bwd_param(int dtype, const void *dy, const void *x,
                           const float *mean, const float *rstd, void *dgamma,
                           void *dbeta, int out_typed, float *sums,
                           int64_t sums_cap, unsigned *tickets,
                           int64_t tickets_cap, int64_t rows, int D,
                           hipStream_t s);
int sa_gelu_fwd(int dtype, const void *x, void *y, int64_t n, hipStream_t s);
int sa_gelu_bwd(int dtype, const void *dy, const void *x, void *dx, int64_t n,
                hipStream_t s);

// ---- conv.hip ---------------------------------------------------------------------------------
struct ConvGeom {
  int N, C, H, W;      // input
  int K, R, S;         // filters (K output channels, R x S window)
  int P, Q;            // output spatial size
  int stride_h, stride_w, pad_h, pad_w;
  int groups;
  // launch plan overrides, 0 = built-in heuristic: tile edge (64 or 128) and
  // the number of workgroups below which the reduction is split
  int tile = 0;
  int split_target = 0;
  // optional float buffer of at least scratch_elems elements: lets a 16-bit
  // forward / backward-data launch split its reduction (partial sums are
  // combined in float there and rounded once into the output).
  // Backward-weight: a buffer of the weight gradient's size that HOLDS ZEROS;
  // filters of more than one tap are accumulated there tap by tap and moved
  // into dw afterwards, and the buffer is handed back holding zeros.
  float *scratch = nullptr;
  int64_t scratch_elems = 0;
  // diagnostic builds only: eight 64-bit phase counters (see conv.hip)
  unsigned long long *prof = nullptr;
  // forward only: per-channel batch-norm statistics of the output, taken while
  // it is produced.  stats is float[kStatCopies][2][K], the launch ADDS
  // sum(y - shift[k]) and sum((y - shift[k])^2) over (n, p) to one of the
  // copies of [0][k] and [1][k]; shift is float[K].
  float *stats = nullptr;
  const float *stats_shift = nullptr;
  // backward-data only: a buffer of 32-bit words that the launch also clears
  // (the weight gradient of the same layer, which the weight-gradient launch
  // that follows accumulates into with atomics): saves that launch a
  // zero-fill pass of its own.  Cleared whenever the call returns 0.
  uint32_t *side_zero = nullptr;
  int64_t side_zero_words = 0;
  // backward-weight only: dw is known to hold zeros already
  bool out_zeroed = false;
  // deterministic mode: float buffer of det_elems elements for the partial
  // sums of a split reduction, one slab per split, summed in split order by
  // a second kernel.  Without it (or when it is too small for two slabs) a
  // deterministic launch does not split its reduction.
  float *det_scratch = nullptr;
  int64_t det_elems = 0;
};
// deterministic mode for every launcher of this library (see common.cuh)
void sa_set_deterministic(int on);
// float32 GEMM tile shape: -1 chosen per problem; 0..6 one shape of the
// two-stage kernel family (64x64, 128x64, 64x128, 128x128, 256x128, 256x96,
// 128x256), 7 the single-stage kernel with 128x128 tiles, for every problem
// (what a tuner sets around one call); -2 the single-stage kernel with its
// own 128 / 64 choice
void sa_set_gemm_f32_tile(int tile);
// 16-bit GEMM tile shape: -1 chosen per problem, 0 128x128, 1 256x256,
// 2 256x128, 3 128x256, 4 64x128
void sa_set_gemm_h16_tile(int tile);
int sa_get_gemm_f32_tile();
int sa_get_deterministic();
// copies of the statistics rows (spreads same-address atomics; one per XCD)
constexpr int kStatCopies = 8;
int sa_conv_fwd(int dtype, const ConvGeom *g, const void *x, const void *w,
                const void *bias, void *y, hipStream_t s);
int sa_conv_bwd_data(int dtype, const ConvGeom *g, const void *dy,
                     const void *w, void *dx, hipStream_t s);
// dx = addend + backward-data; addend has dx's shape and dtype, is a buffer
// other than dx, or null
int sa_conv_bwd_data_acc(int dtype, const ConvGeom *g, const void *dy,
                         const void *w, const void *addend, void *dx,
                         hipStream_t s);
// dtype is the type of dy and x; dw is always a float buffer (zero-filled by
// the launcher itself when the reduction is split)
int sa_conv_bwd_weight(int dtype, const ConvGeom *g, const void *dy,
                       const void *x, void *dw, hipStream_t s);
int sa_conv_bwd_bias(int dtype, const void *dy, void *db, int N, int K,
                     int PQ, hipStream_t s);
// xs[.., p, q] = x[.., 2p, 2q] over `planes` H x W images (xs is
// ceil(H/2) x ceil(W/2)), and its transpose fused with an addend (may be
// null): dx = addend + (values of dxs at even positions, zero elsewhere);
// W even, -3 otherwise.
int sa_subsample2(int dtype, const void *x, void *xs, int64_t planes, int H,
                  int W, hipStream_t s);
int sa_upsample2_add(int dtype, const void *dxs, const void *addend, void *dx,
                     int64_t planes, int H, int W, hipStream_t s);
// The same statistics as ConvGeom::stats from a finished tensor y [N][K][PQ]
// (for convolution kernels without the fused epilogue).
int sa_chan_sums(int dtype, const void *y, float *stats, const float *shift,
                 int N, int K, int PQ, hipStream_t s);

// ---- conv3x3.hip ------------------------------------------------------------------------------
// Direct halo-tile kernel for float32 3x3 / stride 1 / pad 1 convolutions
// (forward and backward-data); the sa_conv_* entry points route to it.
// reduce_channels: C for forward, K for backward-data.
bool sa_conv3x3_applies(int dtype, const ConvGeom *g, int reduce_channels);
int sa_conv3x3_fwd(const ConvGeom *g, const void *x, const void *w,
                   const void *bias, void *y, hipStream_t s);
int sa_conv3x3_bwd_data(const ConvGeom *g, const void *dy, const void *w,
                        const void *addend, void *dx, hipStream_t s);

// ---- depthwise.hip ----------------------------------------------------------------------------
// groups == C == K.  The sa_conv_* entry points route such geometries here
// themselves; the filter gradient has kernels for 3x3, 5x5 and 7x7 windows.
bool sa_dwconv_applies(const ConvGeom *g);
bool sa_dwconv_bwd_weight_applies(const ConvGeom *g);
int sa_dwconv_fwd(int dtype, const ConvGeom *g, const void *x, const void *w,
                  const void *bias, void *y, hipStream_t s);
int sa_dwconv_bwd_data(int dtype, const ConvGeom *g, const void *dy,
                       const void *w, const void *addend, void *dx,
                       hipStream_t s);
int sa_dwconv_bwd_weight(int dtype, const ConvGeom *g, const void *dy,
                         const void *x, void *dw, hipStream_t s);

// ---- batchnorm.hip ----------------------------------------------------------------------------------
// x, y: (N, C, HW) contiguous; statistics and affine parameters are float32.
int sa_bn_fwd_train(int dtype, const void *x, const float *scale,
                    const float *bias, float *rmean, float *rvar, float factor,
                    float eps, void *y, float *mean, float *invstd, int N,
                    int C, int HW, float *work, hipStream_t s);
int sa_bn_fwd_infer(int dtype, const void *x, const float *scale,
                    const float *bias, const float *rmean, const float *rvar,
                    float eps, void *y, int N, int C, int HW, hipStream_t s);
int sa_bn_bwd(int dtype, const void *dy, const void *x, const float *scale,
              const float *mean, const float *invstd, void *dx, float *dscale,
              float *dbias, int N, int C, int HW, float *work, hipStream_t s);
// Fused variants.  act: 0 none, 1 ReLU, 2 add `residual` then ReLU.  The
// backward masks dy by (output > 0) first; for act 2 it needs the saved output
// y and also writes the masked gradient to `dres` (gradient of the residual).
// one-pass training forward from statistics taken by the producer of x
// (ConvGeom::stats); stats: float[kStatCopies][2][C] followed by unsigned[C],
// all zero before the producer ran, cleared again by this launch
int sa_bn_fwd_train_sums(int dtype, const void *x, const float *scale,
                         const float *bias, float *rmean, float *rvar,
                         float factor, float eps, void *y, float *mean,
                         float *invstd, int N, int C, int HW, float *stats,
                         int act, const void *residual, hipStream_t s);
int sa_bn_fwd_train_act(int dtype, const void *x, const float *scale,
                        const float *bias, float *rmean, float *rvar,
                        float factor, float eps, void *y, float *mean,
                        float *invstd, int N, int C, int HW, float *work,
                        int act, const void *residual, hipStream_t s);
int sa_bn_fwd_infer_act(int dtype, const void *x, const float *scale,
                        const float *bias, const float *rmean,
                        const float *rvar, float eps, void *y, int N, int C,
                        int HW, int act, const void *residual, hipStream_t s);
int sa_bn_bwd_act(int dtype, const void *dy, const void *x, const void *y,
                  const float *scale, const float *bias, const float *mean,
                  const float *invstd, void *dx, void *dres, float *dscale,
                  float *dbias, int N, int C, int HW, float *work, int act,
                  hipStream_t s);
// number of floats sa_bn_* need in `work`
int64_t sa_bn_workspace(int N, int C, int HW);

// ---- pooling.hip ---------------------------------------------------------------------------------------
struct PoolGeom {
  int N, C, H, W, P, Q;
  int kh, kw, sh, sw, ph, pw;
  int is_max;
};
int sa_pool_fwd(int dtype, const PoolGeom *g, const void *x, void *y,
                hipStream_t s);
int sa_pool_bwd(int dtype, const PoolGeom *g, const void *dy, const void *x,
                const void *y, void *dx, hipStream_t s);

// ---- optim.hip --------------------------------------------------------------------------------------------
int sa_sgd_update(int dtype, void *p, const void *g, void *buf,
                  const float *lr, const float *momentum,
                  const float *dampening, const float *weight_decay,
                  int nesterov, float scale, int64_t n, hipStream_t s);

// The same update for `count` tensors in ceil(count / kSgdMaxTensors)
// launches; pointer arrays are host arrays (they are copied into the kernel
// arguments).  buf may be nullptr (no momentum) or hold nullptr entries.
namespace sa {
constexpr int kSgdMaxTensors = 48;
constexpr int kSgdChunk = 4096;
struct SgdBatch {
  void *p[kSgdMaxTensors];
  const void *g[kSgdMaxTensors];
  void *buf[kSgdMaxTensors];
  // mixed precision: float copy of a 16-bit parameter (buf is float then too)
  float *master[kSgdMaxTensors];
  int64_t n[kSgdMaxTensors];
  int chunk_start[kSgdMaxTensors + 1];
  int count;
};
}  // namespace sa
// master (null, or one float tensor per parameter): the step is taken on the
// float copy, with float momentum buffers, and rounded into the parameter
int sa_sgd_update_multi(int dtype, int count, void *const *p,
                        const void *const *g, void *const *buf,
                        float *const *master,
                        const int64_t *n, const float *lr,
                        const float *momentum, const float *dampening,
                        const float *weight_decay, int nesterov, float scale,
                        hipStream_t s);

// RMSProp (rule 0), AdaGrad (1), Adam (2) for `count` tensors; states share
// the parameter dtype; hyper-parameters are one-element float device arrays
// (a / b: rho or beta_1 / beta_2; step: Adam's step counter)
namespace sa {
struct AdaptBatch {
  void *p[kSgdMaxTensors];
  const void *g[kSgdMaxTensors];
  void *s1[kSgdMaxTensors];
  void *s2[kSgdMaxTensors];
  int64_t n[kSgdMaxTensors];
  int chunk_start[kSgdMaxTensors + 1];
  int count;
};
}  // namespace sa
int sa_adaptive_update_multi(int dtype, int rule, int count, void *const *p,
                             const void *const *g, void *const *s1,
                             void *const *s2, const int64_t *n,
                             const float *lr, const float *wd,
                             const float *eps, const float *a, const float *b,
                             const float *step, float scale, hipStream_t s);

// ---- index.hip -----------------------------------------------------------------------------------------------
// out[o, j, i] = x[o, idx[j], i]   (x: outer x n_in x inner)
int sa_index_select(int dtype, const void *x, const int *idx, void *out,
                    int64_t outer, int64_t n_in, int64_t n_out, int64_t inner,
                    hipStream_t s);
// out[o, idx[j], i] += dy[o, j, i]  (out zero-filled by the launcher)
int sa_index_add(int dtype, const void *dy, const int *idx, void *out,
                 int64_t outer, int64_t n_in, int64_t n_out, int64_t inner,
                 hipStream_t s);

// ---- rnn.hip ----------------------------------------------------------------------------------------------------
int sa_lstm_cell_fwd(int dtype, const void *gates, const void *c_prev, void *c,
                     void *h, void *acts, int64_t batch, int64_t hidden,
                     hipStream_t s);
int sa_lstm_cell_bwd(int dtype, const void *dh, const void *dc_next,
                     const void *acts, const void *c_prev, const void *c,
                     void *dgates, void *dc_prev, int64_t batch,
                     int64_t hidden, hipStream_t s);
// GRU step: gx / gh are the (batch, 3H) input and recurrent projections (gates
// r, z, n); saves the activated gates and gh_n + bRn for backward.
int sa_gru_cell_fwd(int dtype, const void *gx, const void *gh, const void *brn,
                    const void *h_prev, void *h, void *acts, void *ghn,
                    int64_t batch, int64_t hidden, hipStream_t s);
int sa_gru_cell_bwd(int dtype, const void *dh, const void *acts, const void *ghn,
                    const void *h_prev, void *dgx, void *dgh, void *dh_prev,
                    int64_t batch, int64_t hidden, hipStream_t s);
// d = dh * act'(h); mode 0 relu, 1 tanh
int sa_rnn_act_bwd(int dtype, int mode, const void *dh, const void *h, void *d,
                   int64_t total, hipStream_t s);
// Whole-sequence LSTM of one layer and direction in one launch (float32).
// gx: (T, B, 4H) input projection incl. biases, gates i, f, o, g; R: (4H, H).
// slabs: 2 * (H / 8) * B * H floats of workspace; bar: two zero-initialised
// words (arrival counter, give-up flag: non-zero after the launch means a
// barrier wait ran into its bound and the results are invalid).
struct LstmSeqArgs {
  const float *gx, *R, *h0, *c0;
  float *hseq, *cseq, *acts;            // (T, B, H), (T, B, H), (T, B, 4H)
  const float *dy, *dhy, *dcy;          // dy rows are dy_stride floats apart
  float *dgates, *dhx, *dcx, *slabs;
  unsigned *bar;
  int64_t dy_stride;
  int batch, hidden, steps, reverse;
};
int sa_lstm_seq_supported(int64_t batch, int64_t hidden);
int sa_lstm_seq(const LstmSeqArgs *a, int backward, hipStream_t s);

// ---- comm.hip -------------------------------------------------------------------------------------------------------
// many small tensors <-> one flat bucket: tensor i (counts[i] elements at
// ptrs[i]) occupies the bucket from the sum of the counts before it.  The
// entries are passed in the kernel argument block, kPackMax per launch.
namespace sa {
constexpr int kPackMax = 64;
struct PackBatch {
  void *ptr[kPackMax];
  int64_t offset[kPackMax];
  int64_t count[kPackMax];
};
}  // namespace sa
// ---- winograd.hip: transforms of F(m x m, 3 x 3), m = 2 or 4 ----------------------------------------------------
// All tensors of one call have the element type `dtype`; arithmetic is float.
// U[(m+2)^2][KO][CI] from w[K][C][3][3]; flip: rotated taps and exchanged
// channel roles (the data gradient's filter), KO = C, CI = K
int sa_wino_filter(int dtype, int m, const void *w, void *U, int K, int C,
                   int flip, hipStream_t s);
// V[(m+2)^2][C][T] from x[N][C][H][W], T = N * ceil(H / m) * ceil(W / m); with
// `slices` > 1 (a divisor of T) the layout is [(m+2)^2 * slices][C][T / slices]
int sa_wino_input(int dtype, int m, const void *x, void *V, int N, int C,
                  int H, int W, int slices, hipStream_t s);
// weight gradient: dM[(m+2)^2 * slices][K][T / slices] from dy[N][K][H][W], and
// dw[K][C][3][3] from dU[(m+2)^2 * slices][K][C] (summed over the slices)
int sa_wino_outgrad(int dtype, int m, const void *dy, void *dM, int N, int K,
                    int H, int W, int slices, hipStream_t s);
// (du_float: dU holds floats although dw is a 16-bit tensor)
int sa_wino_filtergrad(int dtype, int m, const void *dU, int du_float,
                       void *dw, int K, int C, int slices, hipStream_t s);
// sa_wino_input + sa_wino_filter, and sa_wino_input + sa_wino_outgrad, as one
// launch each (CI = C of w forward, K with flip)
int sa_wino_prepare(int dtype, int m, const void *x, void *V, int N, int CI,
                    int H, int W, const void *w, void *U, int K, int C,
                    int flip, hipStream_t s);
int sa_wino_prepare_grad(int dtype, int m, const void *x, void *V,
                         const void *dy, void *dM, int N, int C, int K, int H,
                         int W, int slices, hipStream_t s);
// y[N][K][H][W] = inverse transform of M[(m+2)^2][K][T] [+ bias[K]] [+ addend]
int sa_wino_output(int dtype, int m, const void *M, const void *bias,
                   const void *addend, void *y, int N, int K, int H, int W,
                   hipStream_t s);
// the whole route in one call: transforms + batched product + inverse
// transform; mode 0 forward (out = y, N x K), mode 1 data gradient (act = dy,
// out = dx, N x C); workspace: sa_conv3x3_winograd_workspace() elements
int64_t sa_conv3x3_winograd_workspace(int m, int N, int C, int K, int H,
                                      int W);
int sa_conv3x3_winograd(int dtype, int m, int mode, const void *act,
                        const void *w, const void *bias, const void *addend,
                        void *out, int N, int C, int K, int H, int W,
                        void *workspace, hipStream_t s);
// ---- xgmi.hip: collectives over IPC-mapped windows of the node's GPUs -------------------------------------------
namespace sa {
constexpr int kXgmiMaxRanks = 8;    // GPUs of one node
constexpr int kXgmiMaxBlocks = 64;  // workgroups of a collective kernel
constexpr int kXgmiThreads = 512;
struct XgmiPeers {
  char *data[kXgmiMaxRanks];      // every rank's window, as mapped here
  unsigned *flag[kXgmiMaxRanks];  // every rank's flag words [block][rank]
  unsigned *epoch;                // this rank's barrier count per workgroup
  unsigned *error;                // sticky word in mapped host memory
};
}  // namespace sa
int64_t sa_xgmi_signal_bytes();
int sa_xgmi_alloc(int64_t bytes, void **out);
int sa_xgmi_free(void *p);
int sa_xgmi_export(void *p, char *handle64);
int sa_xgmi_import(const char *handle64, void **out);
int sa_xgmi_close(void *p);
int sa_xgmi_error_word(void **host, void **device);
int sa_xgmi_error_word_free(void *host);
// In-place sum over the ranks of `bytes` bytes at byte offset `off` of every
// window; one_shot only up to sa_xgmi_one_shot_max_bytes().
int64_t sa_xgmi_one_shot_max_bytes();
int sa_xgmi_allreduce(int dtype, const sa::XgmiPeers *peers, int rank,
                      int world, int64_t off, int64_t bytes, int one_shot,
                      double timeout_s, hipStream_t s);
int sa_xgmi_allgather(const sa::XgmiPeers *peers, int rank, int world,
                      int64_t off, int64_t gathered_off, int64_t bytes,
                      double timeout_s, hipStream_t s);
int sa_xgmi_alltoall(const sa::XgmiPeers *peers, int rank, int world,
                     int64_t off, int64_t gathered_off, int64_t chunk_bytes,
                     double timeout_s, hipStream_t s);
int sa_xgmi_reduce_scatter(int dtype, const sa::XgmiPeers *peers, int rank,
                           int world, int64_t off, int64_t bytes,
                           double timeout_s, hipStream_t s);
int sa_pack(int src_dtype, int dst_dtype, int n_tensors, void *const *ptrs,
            const int64_t *counts, void *bucket, hipStream_t s);
int sa_unpack(int src_dtype, int dst_dtype, int n_tensors, void *const *ptrs,
              const int64_t *counts, const void *bucket, float scale,
              hipStream_t s);
// threshold sparsification helpers
// the k entries of largest magnitude of x[0..n), in no particular order;
// work holds sa_topk_abs_workspace() 32-bit words
int64_t sa_topk_abs_workspace();
int sa_topk_abs(const float *x, int64_t n, int64_t k, float *vals, int *idx,
                unsigned *work, hipStream_t s);
int sa_sparsify_threshold(const float *x, float thr, float *vals, int *idx,
                          int *count, int64_t n, int64_t cap, hipStream_t s);
int sa_scatter_add_sparse(const float *vals, const int *idx, const int *count,
                          int64_t max_count, float *dense, int64_t n,
                          hipStream_t s);
// Sparse exchange payload: int32 words [count | idx[cap] | float bits[cap]].
// Selection by threshold (|x| >= thr, count may exceed cap: only cap entries
// are stored) or the k largest magnitudes (count = k <= cap).
int sa_sparse_select(const float *x, int64_t n, float thr, int topk, int64_t k,
                     int *payload, int64_t cap, unsigned *work, hipStream_t s);
// dense[idx] += sign * value over `nseg` consecutive payloads of capacity cap
int sa_scatter_add_segments(const int *payloads, int nseg, int64_t cap,
                            float sign, float *dense, int64_t n, hipStream_t s);

// ---- attention.hip ------------------------------------------------------------
// softmax(scale * Q K^T + mask) V per (batch, head), head dimension 64.
// Q / K / V element (b, h, row, d) sits at base + b * stride[0] + h *
// stride[1] + row * stride[2] + d; the additive mask (may be null, same
// dtype) at mask + b * m_stride[0] + h * m_stride[1] + q * m_stride[2] +
// key * m_stride[3] (0 = broadcast).  Outputs and gradients are contiguous
// (B, S, H, 64); lse / delta are float (B, H, Sq).
struct AttnDesc {
  const void *q, *k, *v, *mask;
  int64_t q_stride[3], k_stride[3], v_stride[3], m_stride[4];
  int B, H, Sq, Sk, D;
  float scale;
};
int sa_attention_fwd(int dtype, const AttnDesc *d, void *out, float *lse,
                     hipStream_t s);
int sa_attention_bwd(int dtype, const AttnDesc *d, const void *out,
                     const void *dout, const float *lse, float *delta,
                     void *dq, void *dk, void *dv, hipStream_t s);

// ---- profile.hip --------------------------------------------------------------
// Device-side time stamps that can be captured into a hipGraph (timing events
// cannot): slots[i] = constant-rate wall clock; the closing stamp adds the
// seconds since slots[open] to acc[node] while gate[0] != 0.
int sa_stamp_open(unsigned long long *slots, int i, hipStream_t s);
int sa_stamp_close(const unsigned long long *slots, int open, float *acc,
                   int node, const int *gate, float sec_per_tick,
                   hipStream_t s);
// ticks per second of the clock the stamps read
double sa_wall_clock_hz();
