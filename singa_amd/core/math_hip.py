"""MI355X math backend: every primitive launches a hand-written HIP kernel from
``singa_amd._hip`` on the current torch stream.

Operands arrive as torch views (carrying shape, strides and storage offset of
the tensor); the methods translate them into raw pointers plus collapsed
stride descriptors, so transposed / broadcast views are consumed in place.
Temporary workspaces come from torch's caching allocator, which makes every
method safe to call while a hipGraph is being captured (allocations land in
the graph's private pool).

Contract parity: reference src/core/tensor/tensor_math.h:86-485 and
tensor_math_cuda.h (the CUDA specialisations this backend stands in for).
"""
import json
import math
import os

import torch

from . import dtypes

_U = {n: i for i, n in enumerate([
    "copy", "abs", "exp", "log", "sqrt", "square", "sign", "ceil", "floor",
    "round", "rounde", "sigmoid", "tanh", "relu", "erf", "cos", "cosh", "acos",
    "acosh", "sin", "sinh", "asin", "asinh", "tan", "atan", "atanh", "softplus",
    "softsign", "clamp", "add_s", "sub_s", "mul_s", "div_s", "pow_s", "rdiv_s",
    "lt_s", "le_s", "gt_s", "ge_s", "eq_s", "max_s", "min_s"])}
_B = {n: i for i, n in enumerate([
    "add", "sub", "mul", "div", "pow", "lt", "le", "gt", "ge", "eq", "max",
    "min", "relu_bwd"])}
_R = {"sum": 0, "mean": 1, "max": 2, "min": 3, "asum": 4, "sumsq": 5}

_CODE = {torch.float32: 0, torch.float16: 1, torch.int32: 2,
         torch.bfloat16: 7}


_FLOATS = (torch.float32, torch.float16, torch.bfloat16)
# workgroups a split reduction of a mid-sized product aims at (256 CUs, two
# resident 128 x 128 workgroups each); SINGA_GEMM_SPLIT_TARGET overrides
_SPLIT_TARGET = int(os.environ.get("SINGA_GEMM_SPLIT_TARGET", "512"))


def _code(t):
    try:
        return _CODE[t.dtype]
    except KeyError:
        raise TypeError("dtype %s has no HIP kernels" % t.dtype)


def _collapse(shape, strides_list):
    """Drop size-1 dims and merge neighbouring dims that are contiguous for
    every operand.  Returns (shape, [strides...])."""
    dims = [(s, tuple(st[i] for st in strides_list))
            for i, s in enumerate(shape) if s != 1]
    if not dims:
        return [1], [[0] for _ in strides_list]
    out = [dims[0]]
    for s, st in dims[1:]:
        ps, pst = out[-1]
        if all(a == b * s for a, b in zip(pst, st)):
            out[-1] = (ps * s, st)
        else:
            out.append((s, st))
    shp = [d[0] for d in out]
    sts = [[d[1][k] for d in out] for k in range(len(strides_list))]
    return shp, sts


def _is_linear(shp, st):
    return len(shp) == 1 and st[0] == 1


_CONV_AUTOTUNE = os.environ.get("SINGA_CONV_AUTOTUNE", "1") != "0"
# GEMM: measure (tile shape, reduction slices) per problem on first use; 0
# keeps the heuristic of the launcher
_GEMM_AUTOTUNE = os.environ.get("SINGA_GEMM_AUTOTUNE", "1") != "0"
_CONV16_AUTOTUNE = os.environ.get("SINGA_CONV16_AUTOTUNE", "1") != "0"
# float32 3x3 / stride 1 / pad 1 layers: time the Winograd routes (F(2x2, 3x3)
# and F(4x4, 3x3) transforms around a batched product, csrc/hip/winograd.hip)
# against the direct kernels per geometry; 0 keeps the direct kernels, 2 / 4
# force a transform where it applies (tests)
_CONV_WINOGRAD = os.environ.get("SINGA_CONV_WINOGRAD", "1")
_CONV_WINOGRAD_MAX_BYTES = int(os.environ.get("SINGA_CONV_WINOGRAD_MAX_MB",
                                              "4096")) << 20
# SINGA_DETERMINISTIC=1: bitwise reproducible training steps.  Reductions that
# are split over workgroups (convolution passes, column sums, whole-tensor
# sums, the soft-max loss) add their partial results in a fixed order instead
# of with float atomics, and convolution launch plans come from the heuristic
# instead of from timing.  Slower; not covered: depthwise filter gradients,
# LayerNorm parameter gradients, embedding / scatter gradients and the
# sparsified gradient exchange, which still accumulate with atomics.
_DETERMINISTIC = os.environ.get("SINGA_DETERMINISTIC", "0") == "1"
_DET_SCRATCH_ELEMS = int(os.environ.get("SINGA_DET_SCRATCH_MB", "512")) << 18


def _world_size():
    d = torch.distributed
    return d.get_world_size() if d.is_available() and d.is_initialized() \
        else 1


def _rank():
    d = torch.distributed
    return d.get_rank() if d.is_available() and d.is_initialized() else 0

_CONV1X1_GEMM = os.environ.get("SINGA_CONV1X1_GEMM", "1") != "0"
_CONV3X3_GEMM = os.environ.get("SINGA_CONV3X3_GEMM", "1") != "0"
_CONV_FLAT = os.environ.get("SINGA_CONV_FLAT", "1") != "0"
# 1x1 layers with at most this many positions per image take the kernel over
# flattened positions instead of one GEMM per image
_CONV_FLAT_MAX_HW = int(os.environ.get("SINGA_CONV_FLAT_MAX_HW", "399"))
# 3x3 layers need at least this many positions per image for that kernel (the
# 7 x 7 layers have too few columns to fill the chip with its tiles)
_CONV_FLAT_MIN_HW_3X3 = int(os.environ.get("SINGA_CONV_FLAT_MIN_HW_3X3", "64"))
_CONV_S2_GATHER = os.environ.get("SINGA_CONV_S2_GATHER", "1") != "0"
_CONV_WGRAD_GEMM = os.environ.get("SINGA_CONV_WGRAD_GEMM", "1") != "0"
# SINGA_LN_PARAM_ONE_LAUNCH=0: layer-norm parameter gradients through zeroed
# float buffers and atomics, cast afterwards (five launches instead of one)
_LN_PARAM_ONE_LAUNCH = os.environ.get("SINGA_LN_PARAM_ONE_LAUNCH", "1") != "0"
# column sums (bias gradients) in one launch instead of fill + sum + convert
_REDUCE_COLS_ONE_LAUNCH = os.environ.get("SINGA_REDUCE_COLS_ONE_LAUNCH",
                                         "1") != "0"
_COLS_SCRATCH = 1 << 16
# the backward-data launch of a layer also clears the layer's weight gradient
_CONV_SIDE_ZERO = os.environ.get("SINGA_CONV_SIDE_ZERO", "1") != "0"
_CONV3X3_GEMM_MIN_HW = int(os.environ.get("SINGA_CONV3X3_GEMM_MIN_HW", "2000"))

def _raw_stream(index):
    return torch._C._cuda_getCurrentRawStream(index)


class HipMath(object):
    name = "hip"

    def __init__(self, dev):
        self.dev = dev
        self._dev_index = dev.torch_device.index or 0
        self.k = dev.kernels
        HipMath._kernel_file = getattr(self.k, "__file__", "") or ""
        self.conv_prof = None
        self._wgrad_scratch = {}   # elements -> zero-filled float buffer
        self._det_scratch = None   # slabs of split reductions (deterministic mode)
        self.k.set_deterministic(1 if _DETERMINISTIC else 0)
        # zero-filled scratch of the one-launch column sums, per stream
        self._cols_scratch = {}

    def _column_sums(self, x, out, rows, cols):
        """out[c] = sum over rows of x[r, c]; x contiguous float / 16-bit,
        out contiguous float32 or of x's type.  One launch: partial sums of
        the row slices meet in a scratch buffer that the kernel leaves
        zeroed.  False when the shape exceeds the scratch."""
        if not _REDUCE_COLS_ONE_LAUNCH:
            return False
        stream = self._s()
        sums, tickets = self._zero_scratch()
        return self.k.reduce_cols_fin(
            _code(x), x.data_ptr(), out.data_ptr(),
            0 if out.dtype == torch.float32 else 1, rows, cols,
            sums.data_ptr(), sums.numel(), tickets.data_ptr(),
            tickets.numel(), stream)

    def _zero_scratch(self):
        """(float sums, int tickets) of the current stream: buffers that hold
        zeros between launches (every kernel that borrows them puts the zeros
        back), one pair per stream so that concurrent launches never share."""
        stream = self._s()
        scratch = self._cols_scratch.get(stream)
        if scratch is None:
            d = self.dev.torch_device
            scratch = (torch.zeros(_COLS_SCRATCH, dtype=torch.float32,
                                   device=d),
                       torch.zeros(_COLS_SCRATCH // 16, dtype=torch.int32,
                                   device=d))
            self._cols_scratch[stream] = scratch
        return scratch

    # ---- plumbing -----------------------------------------------------------
    def _s(self):
        # raw handle of the current stream of this device (follows stream
        # contexts and graph capture) without building a Stream object
        return _raw_stream(self._dev_index)

    def _tmp(self, n, dtype=torch.float32):
        return torch.empty(n, dtype=dtype, device=self.dev.torch_device)

    def _unary(self, op, x, out, a=0.0, b=0.0):
        n = out.numel()
        if n == 0:
            return
        if tuple(x.shape) != tuple(out.shape):
            x = x.expand(out.shape)
        shp, (sx, so) = _collapse(list(out.shape),
                                  [list(x.stride()), list(out.stride())])
        ci, co = _code(x), _code(out)
        if _is_linear(shp, sx) and _is_linear(shp, so):
            self.k.unary_contig(op, ci, co, x.data_ptr(), out.data_ptr(), n,
                                float(a), float(b), self._s())
        else:
            if len(shp) > 8:
                x = x.contiguous()
                return self._unary(op, x, out, a, b)
            self.k.unary_strided(op, ci, co, x.data_ptr(), out.data_ptr(), shp,
                                 sx, so, n, float(a), float(b), self._s())

    # ---- element-wise ----------------------------------------------------------
    def fill(self, out, value):
        if out.is_contiguous():
            self.k.fill(_code(out), out.data_ptr(), out.numel(), float(value),
                        self._s())
        else:
            # x * 0 + value over the strided view
            self._unary(_U["mul_s"], out, out, 0.0)
            self._unary(_U["add_s"], out, out, float(value))

    def copy(self, src, dst):
        if src.dtype not in _CODE or dst.dtype not in _CODE:
            # 8-bit / 64-bit types: data movement only.  A plain byte copy runs
            # as a kernel (a memcpy node inside a captured graph is not
            # reliably ordered against the kernels around it)
            if (src.dtype == dst.dtype and src.is_contiguous() and
                    dst.is_contiguous() and src.shape == dst.shape and
                    src.data_ptr() % 4 == 0 and dst.data_ptr() % 4 == 0):
                self.k.copy_bytes(dst.data_ptr(), src.data_ptr(),
                                  src.numel() * src.element_size(), self._s())
            else:
                dst.copy_(src)
            return
        self._unary(_U["copy"], src, dst)

    def unary(self, op, x, out):
        self._unary(_U[op], x, out)

    def clamp(self, x, lo, hi, out):
        self._unary(_U["clamp"], x, out, lo, hi)

    def binary(self, op, a, b, out):
        n = out.numel()
        if n == 0:
            return
        if a.dtype != out.dtype:
            a = self._cast(a, out.dtype)
        if b.dtype != out.dtype:
            b = self._cast(b, out.dtype)
        if tuple(a.shape) != tuple(out.shape):
            a = a.expand(out.shape)
        if tuple(b.shape) != tuple(out.shape):
            b = b.expand(out.shape)
        shp, (sa, sb, so) = _collapse(
            list(out.shape),
            [list(a.stride()), list(b.stride()), list(out.stride())])
        code = _code(out)
        if _is_linear(shp, sa) and _is_linear(shp, sb) and _is_linear(shp, so):
            self.k.binary_contig(_B[op], code, a.data_ptr(), b.data_ptr(),
                                 out.data_ptr(), n, self._s())
        else:
            if len(shp) > 8:
                return self.binary(op, a.contiguous(), b.contiguous(), out)
            self.k.binary_strided(_B[op], code, a.data_ptr(), b.data_ptr(),
                                  out.data_ptr(), shp, sa, sb, so, n,
                                  self._s())

    def _cast(self, x, tdtype):
        out = torch.empty(x.shape, dtype=tdtype, device=x.device)
        self._unary(_U["copy"], x, out)
        return out

    def binary_scalar(self, op, x, s, out):
        self._unary(_U[op + "_s"], x, out, float(s))

    def axpy(self, alpha, x, y):
        if (x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype and
                x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0 and
                x.dtype != torch.int32):
            if torch.is_tensor(alpha):
                if alpha.dtype != torch.float32:
                    alpha = self._cast(alpha, torch.float32)
                self.k.axpy(_code(y), alpha.data_ptr(), 0.0, x.data_ptr(),
                            y.data_ptr(), y.numel(), self._s())
            else:
                self.k.axpy(_code(y), 0, float(alpha), x.data_ptr(),
                            y.data_ptr(), y.numel(), self._s())
            return
        if torch.is_tensor(alpha):
            tmp = torch.empty(x.shape, dtype=y.dtype, device=y.device)
            self.binary("mul", x, alpha.reshape([1] * x.dim()), tmp)
        else:
            tmp = torch.empty(x.shape, dtype=y.dtype, device=y.device)
            self._unary(_U["mul_s"], x, tmp, float(alpha))
        self.binary("add", y, tmp, y)

    # ---- random -------------------------------------------------------------------
    def _rng(self, fn, out, a, b=None):
        tgt = out if out.is_contiguous() and out.dtype != torch.int32 else \
            torch.empty(out.shape, dtype=torch.float32, device=out.device)
        st = self.dev.rng_state.data_ptr()
        if b is None:
            fn(_code(tgt), tgt.data_ptr(), tgt.numel(), float(a), st, self._s())
        else:
            fn(_code(tgt), tgt.data_ptr(), tgt.numel(), float(a), float(b), st,
               self._s())
        if tgt is not out:
            self.copy(tgt, out)

    def uniform(self, out, low, high):
        self._rng(self.k.rng_uniform, out, low, high)

    def gaussian(self, out, mean, std):
        self._rng(self.k.rng_normal, out, mean, std)

    def bernoulli(self, out, p):
        self._rng(self.k.rng_bernoulli, out, p)

    # ---- reductions -------------------------------------------------------------------
    def _f32c(self, x):
        """contiguous float32 version of x (no copy when already so)"""
        if x.dtype == torch.float32 and x.is_contiguous():
            return x
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        self._unary(_U["copy"], x, out)
        return out

    def _reduce_all(self, op, x):
        xf = self._f32c(x)
        acc = self._tmp(1)
        self.k.reduce_all(_R[op], 0, xf.data_ptr(), acc.data_ptr(),
                          xf.numel(), self._s())
        return acc

    def sum_all(self, x, out):
        acc = self._reduce_all("sum", x)
        self.copy(acc.reshape(out.shape), out)

    def asum(self, x):
        return float(self._reduce_all("asum", x).item())

    def nrm2(self, x):
        return float(self._reduce_all("sumsq", x).item()) ** 0.5

    def sum_float(self, x):
        return float(self._reduce_all("sum", x).item())

    def _rows(self, op, x2d, out_flat):
        """reduce each row of a contiguous 2-d tensor"""
        rows, cols = x2d.shape
        if out_flat.dtype != x2d.dtype or not out_flat.is_contiguous():
            tmp = torch.empty(rows, dtype=x2d.dtype, device=x2d.device)
            self.k.reduce_rows(_R[op], _code(x2d), x2d.data_ptr(),
                               tmp.data_ptr(), rows, cols, self._s())
            self.copy(tmp, out_flat)
        else:
            self.k.reduce_rows(_R[op], _code(x2d), x2d.data_ptr(),
                               out_flat.data_ptr(), rows, cols, self._s())

    def _contig(self, x):
        if x.is_contiguous():
            return x
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        self._unary(_U["copy"], x, out)
        return out

    def sum_axis(self, x, axis, out):
        x = self._contig(x)
        rows, cols = x.shape
        o = out.reshape(-1)
        if axis == 1:
            self._rows("sum", x, o)
            return
        # 16-bit input is read as it is; sums are accumulated in float32
        xf = x if x.dtype in _FLOATS else self._f32c(x)
        if o.is_contiguous() and o.dtype in (torch.float32, xf.dtype) and \
                self._column_sums(xf, o, rows, cols):
            return
        if o.dtype == torch.float32 and o.is_contiguous():
            self.k.reduce_cols(_code(xf), xf.data_ptr(), o.data_ptr(), rows,
                               cols, self._s())
        else:
            tmp = self._tmp(cols)
            self.k.reduce_cols(_code(xf), xf.data_ptr(), tmp.data_ptr(), rows,
                               cols, self._s())
            self.copy(tmp, o)

    def row_max(self, x, out):
        self._rows("max", self._contig(x), out.reshape(-1))

    def reduce(self, op, x, axes, out_flat):
        nd = x.dim()
        axes = sorted(axes)
        kept = [i for i in range(nd) if i not in axes]
        rows = 1
        for i in kept:
            rows *= x.shape[i]
        cols = 1
        for i in axes:
            cols *= x.shape[i]
        o = out_flat[:rows]
        if axes == list(range(nd - len(axes), nd)):
            x2 = self._contig(x).reshape(rows, cols)
            self._rows(op, x2, o)
        elif op in ("sum", "mean") and axes == list(range(len(axes))):
            # leading axes: column sums of the (cols, rows) matrix
            x2 = (x if x.dtype in _FLOATS and x.is_contiguous()
                  else self._f32c(x)).reshape(cols, rows)
            if op == "sum" and o.is_contiguous() and \
                    o.dtype in (torch.float32, x2.dtype) and \
                    self._column_sums(x2, o, cols, rows):
                return
            tmp = o if (o.dtype == torch.float32 and o.is_contiguous()) \
                else self._tmp(rows)
            self.k.reduce_cols(_code(x2), x2.data_ptr(), tmp.data_ptr(), cols,
                               rows, self._s())
            if op == "mean":
                self._unary(_U["mul_s"], tmp, tmp, 1.0 / cols)
            if tmp is not o:
                self.copy(tmp, o)
        else:
            xp = x.permute(kept + axes)
            x2 = torch.empty(xp.shape, dtype=x.dtype, device=x.device)
            self._unary(_U["copy"], xp, x2)
            self._rows(op, x2.reshape(rows, cols), o)

    # ---- linear algebra -------------------------------------------------------------------
    def _mat(self, t):
        """2-d / 3-d view usable by the GEMM kernel (unit stride in one of
        the two matrix dims is not required; anything goes)."""
        return t

    def gemm(self, alpha, a, b, beta, c, bias=None, act=0):
        if a.dim() == 2:
            M, K = a.shape
            N = b.shape[1]
            batch = 1
            a_bs = b_bs = c_bs = 0
            a_rs, a_cs = a.stride()
            b_rs, b_cs = b.stride()
        else:
            batch, M, K = a.shape
            N = b.shape[2]
            if b.shape[0] != batch:
                b = b.expand(batch, b.shape[1], b.shape[2])
            a_bs, a_rs, a_cs = a.stride()
            b_bs, b_rs, b_cs = b.stride()
            if not c.is_contiguous():
                raise ValueError("GEMM output must be contiguous")
            c_bs = c.stride()[0]
        if c.stride()[-1] != 1:
            raise ValueError("GEMM output must be row-major")
        ldc = c.stride()[-2] if c.dim() >= 2 else N
        splits = self._gemm_splits(M, N, K) if batch == 1 and c.dtype in (
            torch.float32, torch.float16, torch.bfloat16) else 1
        if _GEMM_AUTOTUNE and c.dtype in (torch.float32, torch.float16,
                                          torch.bfloat16):
            plan = self._gemm_plan(alpha, a, b, beta, c, bias, act,
                                   (M, N, K, batch), splits)
            if plan is not None and self._gemm_planned(
                    plan, alpha, a, b, beta, c, bias, act, (M, N, K, batch)):
                return
        if splits > 1 and c.dtype != torch.float32 and (K // splits) % 64:
            splits = 1          # the 16-bit tiles are 64 deep
        if splits == 1 and batch == 1 and K % 8 == 0 and c.dtype in (
                torch.float16, torch.bfloat16):
            # 64-deep tiles: slices of a multiple of 64, the last one shorter
            n_sl, kc = self._gemm_uneven_slices(M, N, K, 64)
            if n_sl > 1:
                part = torch.empty(n_sl * M * N, dtype=torch.float32,
                                   device=c.device)
                if self.k.gemm_h16_partial(
                        _code(c), M, N, kc, a.data_ptr(), a_rs, a_cs,
                        kc * a_cs, b.data_ptr(), b_rs, b_cs, kc * b_rs,
                        part.data_ptr(), n_sl, self._s(), K) == 0:
                    self.k.gemm_splitk_reduce(
                        _code(c), part.data_ptr(), n_sl, M, N, float(alpha),
                        float(beta), c.data_ptr(), ldc,
                        bias.data_ptr() if bias is not None else 0, act,
                        self._s())
                    return
        if splits == 1 and batch == 1 and c.dtype == torch.float32:
            # no divisor of K fits (an odd reduction length): unequal slices
            n_sl, kc = self._gemm_uneven_slices(M, N, K)
            if n_sl > 1:
                part = torch.empty(n_sl * M * N, dtype=torch.float32,
                                   device=c.device)
                self.k.gemm_f32_slices(M, N, kc, K, a.data_ptr(), a_rs, a_cs,
                                       b.data_ptr(), b_rs, b_cs,
                                       part.data_ptr(), n_sl, self._s())
                self.k.gemm_splitk_reduce(
                    _code(c), part.data_ptr(), n_sl, M, N, float(alpha),
                    float(beta), c.data_ptr(), ldc,
                    bias.data_ptr() if bias is not None else 0, act,
                    self._s())
                return
        if splits > 1:
            # too few output tiles to occupy the chip: the reduction is cut
            # into `splits` equal slices computed as a batch of float partial
            # products, summed (and scaled / biased / activated) by one
            # epilogue kernel
            kc = K // splits
            part = torch.empty(splits * M * N, dtype=torch.float32,
                               device=c.device)
            if c.dtype == torch.float32:
                self.k.gemm(0, M, N, kc, 1.0, a.data_ptr(), a_rs, a_cs,
                            kc * a_cs, b.data_ptr(), b_rs, b_cs, kc * b_rs,
                            0.0, part.data_ptr(), N, M * N, splits, 0, 0,
                            self._s())
                done = True
            else:
                done = self.k.gemm_h16_partial(
                    _code(c), M, N, kc, a.data_ptr(), a_rs, a_cs, kc * a_cs,
                    b.data_ptr(), b_rs, b_cs, kc * b_rs, part.data_ptr(),
                    splits, self._s()) == 0
            if done:
                self.k.gemm_splitk_reduce(
                    _code(c), part.data_ptr(), splits, M, N, float(alpha),
                    float(beta), c.data_ptr(), ldc,
                    bias.data_ptr() if bias is not None else 0, act,
                    self._s())
                return
        self.k.gemm(_code(c), M, N, K, float(alpha), a.data_ptr(), a_rs, a_cs,
                    a_bs, b.data_ptr(), b_rs, b_cs, b_bs, float(beta),
                    c.data_ptr(), ldc, c_bs, batch,
                    bias.data_ptr() if bias is not None else 0, act, self._s())

    # ---- GEMM launch plans -------------------------------------------------
    # (tile shape, reduction slices) per problem -- element type, extents,
    # batch and which index of each operand is contiguous -- measured on the
    # real operands the first time the problem is met outside stream capture
    # and kept in the same store as the convolution plans.  Tile codes as in
    # csrc/hip/kernels.h: sa_set_gemm_f32_tile / sa_set_gemm_h16_tile.
    _gemm_plans = {}
    _GEMM_TILES_F32 = (-2, 7, 0, 1, 2, 3)
    _GEMM_TILES_H16 = (-1, 0, 1, 2, 3, 4)

    def _gemm_planned(self, plan, alpha, a, b, beta, c, bias, act, dims):
        """Run the product with ``plan``; False when the plan does not apply
        to these operands (a 16-bit layout the tiled kernel refuses)."""
        tile, splits = plan
        M, N, K, batch = dims
        if batch == 1:
            a_bs = b_bs = c_bs = 0
            a_rs, a_cs = a.stride()
            b_rs, b_cs = b.stride()
        else:
            a_bs, a_rs, a_cs = a.stride()
            b_bs, b_rs, b_cs = b.stride()
            c_bs = c.stride()[0]
        ldc = c.stride()[-2] if c.dim() >= 2 else N
        bp = bias.data_ptr() if bias is not None else 0
        f32 = c.dtype == torch.float32
        code = _code(c)
        setter = self.k.set_gemm_f32_tile if f32 else self.k.set_gemm_h16_tile
        setter(tile)
        try:
            if splits > 1:
                kc = K // splits
                part = torch.empty(splits * M * N, dtype=torch.float32,
                                   device=c.device)
                if f32:
                    self.k.gemm(0, M, N, kc, 1.0, a.data_ptr(), a_rs, a_cs,
                                kc * a_cs, b.data_ptr(), b_rs, b_cs,
                                kc * b_rs, 0.0, part.data_ptr(), N, M * N,
                                splits, 0, 0, self._s())
                elif self.k.gemm_h16_partial(
                        code, M, N, kc, a.data_ptr(), a_rs, a_cs, kc * a_cs,
                        b.data_ptr(), b_rs, b_cs, kc * b_rs, part.data_ptr(),
                        splits, self._s()) != 0:
                    return False
                self.k.gemm_splitk_reduce(
                    code, part.data_ptr(), splits, M, N, float(alpha),
                    float(beta), c.data_ptr(), ldc, bp, act, self._s())
            else:
                self.k.gemm(code, M, N, K, float(alpha), a.data_ptr(), a_rs,
                            a_cs, a_bs, b.data_ptr(), b_rs, b_cs, b_bs,
                            float(beta), c.data_ptr(), ldc, c_bs, batch, bp,
                            act, self._s())
        finally:
            setter(-1)
        return True

    def _gemm_plan(self, alpha, a, b, beta, c, bias, act, dims, rule_splits):
        """(tile, slices) for this product, or None to take the heuristic
        path."""
        M, N, K, batch = dims
        if M * N * K < (1 << 21) or _DETERMINISTIC:
            return None            # launch-bound: nothing to gain
        key = ("gemm", _code(c), M, N, K, batch,
               a.stride()[-1] == 1, a.stride()[-2] == 1,
               b.stride()[-1] == 1, b.stride()[-2] == 1)
        plan = self._gemm_plans.get(key)
        if plan is not None:
            return plan
        if _world_size() > 1 and _rank() != 0:
            return None            # rank 0's table arrives before the capture
        name = repr(key)
        stored = self._plan_store()[1].get(name)
        if stored is not None:
            plan = (int(stored[0]), int(stored[1]))
        elif not torch.cuda.is_current_stream_capturing() and beta == 0.0 \
                and K % 8 == 0:
            plan = self._measure_gemm_plan(alpha, a, b, c, bias, act, dims,
                                           rule_splits)
            self._persist_plan(name, plan)
        if plan is not None:
            self._gemm_plans[key] = plan
        return plan

    def _measure_gemm_plan(self, alpha, a, b, c, bias, act, dims,
                           rule_splits):
        M, N, K, batch = dims
        f32 = c.dtype == torch.float32
        depth = 4 if f32 else 64      # slices end on a whole reduction tile
        cuts = {1}
        if batch == 1:
            for sp in (rule_splits, 2 * rule_splits, rule_splits // 2, 2, 4):
                if sp > 1 and K % sp == 0 and (K // sp) % depth == 0 and \
                        K // sp >= 128 and \
                        sp * M * N * 4 <= (256 << 20):
                    cuts.add(sp)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream()
        best, best_t = None, None
        for sp in sorted(cuts):
            for tile in (self._GEMM_TILES_F32 if f32
                         else self._GEMM_TILES_H16):
                if f32 and tile == 7 and (M < 96 or N < 96):
                    continue
                plan = (tile, sp)

                def run(n):
                    e0.record(stream)
                    for _ in range(n):
                        if not self._gemm_planned(plan, alpha, a, b, 0.0, c,
                                                  bias, act, dims):
                            return None
                    e1.record(stream)
                    e1.synchronize()
                    return e0.elapsed_time(e1) / n
                t = run(2)                     # pages the code in as well
                if t is None:
                    continue
                if t < 0.15 and _world_size() <= 1:
                    # the host cannot launch this fast: time the kernels
                    # alone by replaying a captured burst (not while other
                    # ranks' collectives may be in flight on side streams)
                    burst = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(burst, capture_error_mode="thread_local"):
                        for _ in range(16):
                            self._gemm_planned(plan, alpha, a, b, 0.0, c,
                                               bias, act, dims)
                    burst.replay()
                    e0.record(stream)
                    burst.replay()
                    e1.record(stream)
                    e1.synchronize()
                    t = e0.elapsed_time(e1) / 16
                    del burst
                else:
                    t = run(max(3, min(16, int(2.0 / t))))
                if best_t is None or t < best_t * 0.98:
                    best, best_t = plan, t
        return best

    @staticmethod
    def _split_rule(M, N, K):
        """(most slices, least slice depth) for a product with too few output
        tiles to occupy the chip, or None: skinny products are counted in
        64 x 64 tiles and may be cut down to 64-deep slices, mid-sized ones in
        128 x 128 tiles with 512- or 1024-deep slices."""
        tiles64 = -(-M // 64) * -(-N // 64)
        tiles128 = -(-M // 128) * -(-N // 128)
        if tiles64 <= 64 and K >= 512:
            return min(32, 256 // tiles64), 64
        if tiles128 <= 96 and K >= 1024 and M >= 96 and N >= 96:
            return _SPLIT_TARGET // tiles128, 512
        if tiles128 < 384 and K >= 2048 and M >= 96 and N >= 96:
            # a weight gradient over many tokens: a few hundred tiles with a
            # long reduction each; aim at two workgroups per CU
            return (_SPLIT_TARGET + tiles128 // 2) // tiles128, 1024
        return None

    @classmethod
    def _gemm_splits(cls, M, N, K):
        """Number of equal reduction slices: the largest divisor of K within
        the rule's limits, with slices 4-aligned (for the vector loads)."""
        rule = cls._split_rule(M, N, K)
        if rule is None:
            return 1
        most, depth = rule
        best = 1
        for d in range(2, most + 1):
            if K % d == 0 and (K // d) >= depth and (K // d) % 4 == 0:
                best = d
        return best

    @classmethod
    def _gemm_uneven_slices(cls, M, N, K, align=4):
        """(slices, slice depth) for a reduction length without a suitable
        divisor (e.g. an odd one): slices of a multiple of ``align`` indices,
        the last one shorter."""
        rule = cls._split_rule(M, N, K)
        if rule is None:
            return 1, K
        most, depth = rule
        d = max(1, min(most, K // max(depth, align)))
        if d < 2:
            return 1, K
        kc = -(-K // d)
        kc += -kc % align
        n = -(-K // kc)
        return (n, kc) if n > 1 else (1, K)

    def gemv(self, alpha, a, v, beta, out):
        self.gemm(alpha, a, v.reshape(-1, 1), beta, out.reshape(-1, 1))

    def dgmm_rows(self, m, v, out):
        self.binary("mul", m, v.reshape(1, -1), out)

    def dgmm_cols(self, m, v, out):
        self.binary("mul", m, v.reshape(-1, 1), out)

    # ---- softmax / losses ---------------------------------------------------------------------
    def softmax(self, x, out):
        x = self._contig(x)
        self.k.softmax_fwd(_code(x), x.data_ptr(), out.data_ptr(), x.shape[0],
                           x.shape[1], self._s())

    def softmax_bwd(self, dy, y, dx):
        dy, y = self._contig(dy), self._contig(y)
        self.k.softmax_bwd(_code(y), dy.data_ptr(), y.data_ptr(),
                           dx.data_ptr(), y.shape[0], y.shape[1], self._s())

    def _target(self, t, batch):
        t = self._contig(t)
        is_int = 1 if t.numel() == batch else 0
        return t, is_int, _code(t)

    def cross_entropy_fwd(self, p, t, loss):
        p = self._contig(p)
        t, is_int, tc = self._target(t, p.shape[0])
        self.k.cross_entropy_fwd(_code(p), p.data_ptr(), t.data_ptr(), is_int,
                                 tc, loss.data_ptr(), p.shape[0], p.shape[1],
                                 self._s())

    def softmax_ce_bwd(self, p, t):
        t, is_int, tc = self._target(t, p.shape[0])
        self.k.softmax_ce_bwd(_code(p), p.data_ptr(), t.data_ptr(), is_int, tc,
                              p.shape[0], p.shape[1], self._s())

    def softmax_ce_fused_fwd(self, x, t, p, loss):
        x = self._contig(x)
        t, is_int, tc = self._target(t, x.shape[0])
        if x.dtype == torch.float32:
            self.k.softmax_ce_fused_fwd(0, x.data_ptr(), t.data_ptr(), is_int,
                                        tc, p.data_ptr(), loss.data_ptr(),
                                        x.shape[0], x.shape[1], self._s())
            return
        # 16-bit logits: soft-max in the tensor dtype, loss reduced in float32
        self.softmax(x, p)
        per = torch.empty(x.shape[0], dtype=x.dtype, device=x.device)
        self.cross_entropy_fwd(p, t, per)
        acc = self._reduce_all("sum", per)
        self._unary(_U["mul_s"], acc, loss.reshape(1), 1.0 / x.shape[0])

    def softmax_ce_fused_bwd(self, p, t, dx):
        t, is_int, tc = self._target(t, p.shape[0])
        self.k.softmax_ce_fused_bwd(_code(p), p.data_ptr(), t.data_ptr(),
                                    is_int, tc, dx.data_ptr(), p.shape[0],
                                    p.shape[1], self._s())

    # ---- convolution ---------------------------------------------------------------------------
    def _geom(self, xshape, wshape, yshape, stride, pad, groups):
        N, C, H, W = xshape
        K, _, R, S = wshape
        g = self.k.ConvGeom(N, C, H, W, K, R, S, yshape[2], yshape[3],
                            stride[0], stride[1], pad[0], pad[1], groups)
        if self.conv_prof is not None:   # diagnostic builds: phase counters
            g.set_prof(self.conv_prof.data_ptr())
        if _DETERMINISTIC:
            if self._det_scratch is None:
                self._det_scratch = torch.empty(
                    _DET_SCRATCH_ELEMS, dtype=torch.float32,
                    device=self.dev.torch_device)
            g.set_det_scratch(self._det_scratch.data_ptr(),
                              self._det_scratch.numel())
        return g

    # Launch-plan autotuning (the role of cuDNN's algorithm search in the
    # reference, convolution.cc:514-640): the first time a (pass, geometry,
    # dtype) is seen outside stream capture, every candidate tile edge / split
    # target is timed on the real operands and the fastest one is kept.
    # Plans persist in a JSON file keyed by GPU architecture and kernel build
    # (SINGA_CONV_PLAN_CACHE: path, or 0 for no file), so later processes --
    # and shapes first met under stream capture -- start from measured plans.
    # With several ranks only rank 0 consults the file or measures; the other
    # ranks run the eager first iteration on the heuristic and take rank 0's
    # whole table in ONE broadcast over the host control group before the step
    # is captured (``sync_conv_plans``), so all ranks replay identical kernels
    # and no collective is issued per geometry.  SINGA_CONV_AUTOTUNE=0 keeps
    # the heuristic.
    _conv_plans = {}
    _CONV_CANDIDATES = ((64, 512), (64, 1024), (64, 2048), (128, 256),
                        (128, 512), (128, 1024), (128, 2048))
    _plan_file = None          # (path, {key string: [tile, split]}) once read
    _kernel_file = ""          # path of the loaded kernel library

    @classmethod
    def _plan_store(cls):
        """The persisted plans of this architecture and kernel build."""
        if cls._plan_file is None:
            path = os.environ.get("SINGA_CONV_PLAN_CACHE")
            if path is None:
                path = os.path.join(
                    os.environ.get("XDG_CACHE_HOME",
                                   os.path.expanduser("~/.cache")),
                    "singa_amd", "conv_plans.json")
            plans = {}
            if path not in ("", "0"):
                try:
                    with open(path) as f:
                        blob = json.load(f)
                    if blob.get("build") == cls._build_id():
                        plans = dict(blob.get("plans", {}))
                except (OSError, ValueError):
                    plans = {}
            cls._plan_file = (path, plans)
        return cls._plan_file

    @staticmethod
    def _build_id():
        """Architecture plus size and time stamp of the kernel library: plans
        measured with other kernels are not reused."""
        so = HipMath._kernel_file
        try:
            st = os.stat(so)
            stamp = "%d-%d" % (st.st_size, int(st.st_mtime))
        except OSError:
            stamp = "unknown"
        arch = torch.cuda.get_device_properties(0).gcnArchName \
            if torch.cuda.is_available() else "cpu"
        return "%s/%s" % (arch, stamp)

    @classmethod
    def _persist_plan(cls, name, plan):
        path, plans = cls._plan_store()
        plans[name] = list(plan)
        if path in ("", "0"):
            return
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "w") as f:
                json.dump({"build": cls._build_id(), "plans": plans}, f)
            os.replace(tmp, path)
        except OSError:
            pass                      # a read-only home: keep the process cache

    def _measure_plan(self, g, launch):
        best, best_t = (0, 0), None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream()
        for cand in ((0, 0),) + self._CONV_CANDIDATES:
            g.tile, g.split_target = cand
            launch(g)                      # warm up / page in code
            e0.record(stream)
            for _ in range(3):
                launch(g)
            e1.record(stream)
            e1.synchronize()
            t = e0.elapsed_time(e1)
            if best_t is None or t < best_t * 0.97:   # prefer the default
                best, best_t = cand, t
        return best

    def _tuned(self, mode, g, key, launch):
        """Set g.tile / g.split_target for this launch.  ``launch(g)`` runs the
        convolution once (it only overwrites the output).  Returns True when
        timing launches were issued, i.e. the output has been written."""
        if key[4] > 1 and key[4] == key[0][1] == key[1][0]:
            return False  # depthwise kernels have no launch plan to choose
        full = (mode,) + key
        measured = False
        plan = self._conv_plans.get(full)
        if plan is None:
            # measured plans differ from run to run with the timing noise,
            # and with them the order in which partial sums are added
            if not _CONV_AUTOTUNE or _DETERMINISTIC:
                return False
            capturing = torch.cuda.is_current_stream_capturing()
            if _world_size() > 1 and _rank() != 0:
                return False  # rank 0's table arrives before the capture
            name = repr(full)
            stored = self._plan_store()[1].get(name)
            if stored is not None:
                plan = (int(stored[0]), int(stored[1]))
            elif not capturing:
                plan = self._measure_plan(g, launch)
                measured = True
                self._persist_plan(name, plan)
            if plan is None:
                return False  # first met under capture, nothing stored
            self._conv_plans[full] = plan
        g.tile, g.split_target = plan
        return measured

    def sync_conv_plans(self, comm):
        """Every rank adopts rank 0's launch plans (one host-side broadcast
        of the table; a no-op for a lone rank)."""
        if comm is None or _world_size() <= 1 or not _CONV_AUTOTUNE:
            return
        mine = {repr(k): list(v) for k, v in self._conv_plans.items()}
        mine.update({repr(k): list(v) for k, v in self._gemm_plans.items()})
        table = comm.broadcast_object(mine if _rank() == 0 else None)
        if _rank() != 0 and table:
            import ast
            for name, plan in table.items():
                key = ast.literal_eval(name)
                store = type(self)._gemm_plans if key[0] == "gemm" \
                    else type(self)._conv_plans
                store[key] = (int(plan[0]), int(plan[1]))

    def _split_scratch(self, g, out, rows, cols, groups):
        """16-bit outputs: lend the launcher a float buffer of the output's
        size when the launch is small enough for a split reduction to pay
        (fewer 64 x 64 tiles than the largest split target).  The returned
        tensor must stay referenced until the launch has been issued."""
        if out.dtype == torch.float32:
            return None
        tiles = -(-rows // 64) * -(-cols // 64) * groups
        if tiles >= 2048:
            return None
        buf = torch.empty(out.numel(), dtype=torch.float32, device=out.device)
        g.set_scratch(buf.data_ptr(), buf.numel())
        return buf

    @staticmethod
    def _pointwise16(x, w, stride, pad, groups):
        """16-bit 1x1 / stride 1 / unpadded convolution: a plain GEMM per
        image, served by the tiled 16-bit GEMM kernel."""
        return (_CONV1X1_GEMM and x.dtype in (torch.float16, torch.bfloat16)
                and w.shape[2] == 1 and w.shape[3] == 1 and groups == 1 and
                tuple(stride) == (1, 1) and tuple(pad) == (0, 0))

    @staticmethod
    def _window16(x, w, stride, pad, groups):
        """16-bit 3x3 / stride 1 / pad 1 convolution: nine shifted GEMMs per
        image accumulated inside one launch of the tiled 16-bit kernel."""
        return (_CONV3X3_GEMM and x.dtype in (torch.float16, torch.bfloat16)
                and w.shape[2] == 3 and w.shape[3] == 3 and groups == 1 and
                tuple(stride) == (1, 1) and tuple(pad) == (1, 1) and
                x.is_contiguous() and
                # small images leave the 128-position tiles mostly empty
                x.shape[2] * x.shape[3] >= _CONV3X3_GEMM_MIN_HW)

    def _conv3x3_h16(self, mode, act, w, addend, out):
        """Run the nine-tap GEMM; False when the geometry does not fit."""
        N, _, H, W = act.shape
        K, C = w.shape[0], w.shape[1]
        wt = torch.empty(9 * K * C, dtype=w.dtype, device=w.device)
        return self.k.conv3x3_h16(
            _code(act), mode, N, C, K, H, W, act.data_ptr(), w.data_ptr(),
            wt.data_ptr(), addend.data_ptr() if addend is not None else 0,
            out.data_ptr(), self._s())

    def _flat16(self, mode, act, w, addend, out, stride, pad, groups):
        """16-bit 1x1 or 3x3 / pad 1 stride-1 convolution (forward or
        backward-data) on the tap GEMM over the flattened positions of all
        images; any image size.  False when the geometry does not fit."""
        if not (_CONV_FLAT and groups == 1 and tuple(stride) == (1, 1) and
                act.dtype in (torch.float16, torch.bfloat16) and
                act.is_contiguous() and out.is_contiguous()):
            return False
        window = (w.shape[2], w.shape[3]) + tuple(pad)
        if window not in ((1, 1, 0, 0), (3, 3, 1, 1)):
            return False
        N, _, H, W = act.shape
        K, C = w.shape[0], w.shape[1]
        taps = w.shape[2] * w.shape[3]
        if taps == 9 and H * W < _CONV_FLAT_MIN_HW_3X3:
            return False
        wt = torch.empty(taps * K * C, dtype=w.dtype, device=w.device)
        return self.k.conv_flat_h16(
            _code(act), mode, N, C, K, H, W, taps, act.data_ptr(),
            w.data_ptr(), wt.data_ptr(),
            addend.data_ptr() if addend is not None else 0, out.data_ptr(),
            self._s())

    @staticmethod
    def _small_image(t):
        return t.shape[2] * t.shape[3] <= _CONV_FLAT_MAX_HW

    # ---- 16-bit stride-1 convolutions: which GEMM-shaped kernel -------------
    # Forward (mode 0: act = x, out = y) and backward-data (mode 1: act = dy,
    # out = dx) of a 1x1 or 3x3 / pad 1 layer can run on the tap GEMM over the
    # flattened positions of all images ("flat"), on one GEMM per image with
    # one of the 16-bit tile shapes ("pointwise", 1x1 only) or on the nine-tap
    # GEMM per image ("window", 3x3 only).  Which one is fastest depends on
    # image size, channel counts and batch: each applicable one is timed the
    # first time a geometry is met outside stream capture, the choice is kept
    # with the other launch plans (SINGA_CONV16_AUTOTUNE=0: a fixed order).
    _ROUTE_FLAT, _ROUTE_WINDOW, _ROUTE_POINTWISE = 0, 1, 2

    def _routes16(self, mode, act, w, addend, out, stride, pad, groups):
        """[((route, tile), run)] in the order of the fixed rule; run() ->
        True when the kernel took the geometry."""
        ref = act if mode == 0 else out          # the tensor with C channels
        N, C, H, W = ref.shape
        K = w.shape[0]
        code = _code(act)
        if addend is not None:
            assert addend.shape == out.shape and addend.dtype == out.dtype \
                and addend.data_ptr() != out.data_ptr()
        ad = addend.data_ptr() if addend is not None else 0

        def flat():
            return bool(self._flat16(mode, act, w, addend, out, stride, pad,
                                     groups))

        def window():
            return bool(self._conv3x3_h16(mode, act, w, addend, out))

        def pointwise(tile):
            def run():
                self.k.set_gemm_h16_tile(tile)
                try:
                    return bool(self.k.conv1x1_h16(
                        code, mode, N, C, K, H * W, act.data_ptr(),
                        w.data_ptr(), ad, out.data_ptr(), self._s()))
                finally:
                    self.k.set_gemm_h16_tile(-1)
            return run
        routes = []
        if mode == 0 and self._small_image(act):
            routes.append(((self._ROUTE_FLAT, -1), flat))
        if self._pointwise16(ref, w, stride, pad, groups):
            routes.append(((self._ROUTE_POINTWISE, -1), pointwise(-1)))
            if K > 64 if mode == 0 else C > 64:
                # (64 rows or fewer always run on the 64 x 128 tile)
                for tile in (0, 1, 2, 3):
                    routes.append(((self._ROUTE_POINTWISE, tile),
                                   pointwise(tile)))
        if self._window16(act, w, stride, pad, groups):
            routes.append(((self._ROUTE_WINDOW, -1), window))
        if not routes or routes[0][0][0] != self._ROUTE_FLAT:
            routes.append(((self._ROUTE_FLAT, -1), flat))
        return routes

    def _run_routes16(self, mode, act, w, addend, out, stride, pad, groups):
        routes = self._routes16(mode, act, w, addend, out, stride, pad, groups)
        key = ("c16", mode, tuple(act.shape), tuple(w.shape), tuple(stride),
               tuple(pad), _code(act), addend is not None)
        plan = self._conv_plans.get(key)
        if plan is None and _CONV16_AUTOTUNE and not _DETERMINISTIC and \
                (_world_size() <= 1 or _rank() == 0):
            name = repr(key)
            stored = self._plan_store()[1].get(name)
            if stored is not None:
                plan = (int(stored[0]), int(stored[1]))
            elif not torch.cuda.is_current_stream_capturing():
                plan = self._measure_routes16(routes)
                if plan is not None:
                    self._persist_plan(name, plan)
            if plan is not None:
                self._conv_plans[key] = plan
        if plan is not None:
            for which, run in routes:
                if which == tuple(plan):
                    if run():
                        return True
                    break
        for _, run in routes:
            if run():
                return True
        return False

    def _measure_routes16(self, routes):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream()
        best, best_t = None, None
        for which, run in routes:
            if not run():               # the kernel refuses the geometry
                continue
            if _world_size() <= 1:
                burst = torch.cuda.CUDAGraph()
                with torch.cuda.graph(burst,
                                      capture_error_mode="thread_local"):
                    for _ in range(8):
                        run()
                burst.replay()
                e0.record(stream)
                burst.replay()
                e1.record(stream)
                e1.synchronize()
                t = e0.elapsed_time(e1)
                del burst
            else:
                # collectives of this rank may be in flight on side streams:
                # no capture next to them, plain launches instead
                e0.record(stream)
                for _ in range(8):
                    run()
                e1.record(stream)
                e1.synchronize()
                t = e0.elapsed_time(e1)
            if best_t is None or t < best_t * 0.97:    # prefer the rule's
                best, best_t = which, t
        return best

    # ---- float32 Winograd routes ------------------------------------------------
    @staticmethod
    def _winograd_applies(act, w, out, stride, pad, groups):
        # the transformed image and the products are (m+2)^2 / m^2 times the
        # activation each: no route for layers whose two scratch tensors
        # would exceed SINGA_CONV_WINOGRAD_MAX_MB (4 GiB)
        if act.dim() == 4:
            tiles = act.shape[0] * (-(-act.shape[2] // 4)) * \
                (-(-act.shape[3] // 4))
            scratch = 36 * tiles * (act.shape[1] + out.shape[1]) * \
                act.element_size()
            if scratch > _CONV_WINOGRAD_MAX_BYTES:
                return False
        return (_CONV_WINOGRAD != "0" and groups == 1 and
                act.dtype in _FLOATS and w.dtype == act.dtype and
                out.dtype == act.dtype and
                tuple(w.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and
                tuple(pad) == (1, 1) and act.is_contiguous() and
                out.is_contiguous() and
                tuple(act.shape[2:]) == tuple(out.shape[2:]))

    def _winograd(self, m, mode, act, w, bias, addend, out):
        """out = conv3x3(act, w) (mode 0) or its data gradient (mode 1: act
        is dy, the filter is rotated and its channel roles exchanged) through
        F(m x m, 3 x 3): filter and input transforms, (m + 2)^2 products
        [KO x CI] x [CI x tiles] as one batched GEMM, output transform (which
        adds ``bias`` / ``addend``)."""
        N, CI, H, W = act.shape
        KO = out.shape[1]
        K, C = w.shape[0], w.shape[1]
        assert (KO, CI) == ((K, C) if mode == 0 else (C, K))
        P = (m + 2) * (m + 2)
        T = N * (-(-H // m)) * (-(-W // m))
        dt, code = act.dtype, _code(act)
        U = self._tmp(P * KO * CI, dt).view(P, KO, CI)
        V = self._tmp(P * CI * T, dt).view(P, CI, T)
        M = self._tmp(P * KO * T, dt).view(P, KO, T)
        s = self._s()
        self.k.wino_prepare(code, m, act.data_ptr(), V.data_ptr(), N, CI, H,
                            W, w.data_ptr(), U.data_ptr(), K, C, mode == 1, s)
        self.gemm(1.0, U, V, 0.0, M)
        self.k.wino_output(code, m, M.data_ptr(),
                           bias.data_ptr() if bias is not None else 0,
                           addend.data_ptr() if addend is not None else 0,
                           out.data_ptr(), N, KO, H, W, s)
        return True

    @staticmethod
    def _winograd_slices(P, K, C, T):
        """Into how many slices the tile sum of a Winograd weight gradient is
        cut: enough 64 x 64 output tiles for the chip, slices not shorter
        than 128 tiles."""
        tiles = P * (-(-K // 64)) * (-(-C // 64))
        best = 1
        for sl in (2, 4, 7, 8, 14, 16, 28, 32):
            if tiles * best >= 512:
                break
            if T % sl == 0 and T // sl >= 128:
                best = sl
        return best

    def _winograd_wgrad(self, m, dy, x, dw):
        """dw = weight gradient of the 3x3 / stride 1 / pad 1 convolution
        through F(m x m, 3 x 3): input transform of x, the transposed output
        transform of dy, (m + 2)^2 products [K x tiles] x [tiles x C] (the
        tile sum cut into slices that run as further batch entries), and the
        transposed filter transform, which also adds the slices up."""
        N, C, H, W = x.shape
        K = dy.shape[1]
        P = (m + 2) * (m + 2)
        T = N * (-(-H // m)) * (-(-W // m))
        S = self._winograd_slices(P, K, C, T)
        Tp = T // S
        dt, code = x.dtype, _code(x)
        V = self._tmp(P * C * T, dt).view(P * S, C, Tp)
        dM = self._tmp(P * K * T, dt).view(P * S, K, Tp)
        dU = self._tmp(P * S * K * C).view(P * S, K, C)     # float sums
        s = self._s()
        self.k.wino_prepare_grad(code, m, x.data_ptr(), V.data_ptr(),
                                 dy.data_ptr(), dM.data_ptr(), N, C, K, H, W,
                                 S, s)
        if dt == torch.float32:
            self.gemm(1.0, dM, V.transpose(1, 2), 0.0, dU)
        elif Tp % 8 or self.k.gemm_h16_partial(
                code, K, C, Tp, dM.data_ptr(), Tp, 1, K * Tp, V.data_ptr(), 1,
                Tp, C * Tp, dU.data_ptr(), P * S, s, 0) != 0:
            return False    # the tiled 16-bit kernel refuses the layout
        self.k.wino_filtergrad(code, m, dU.data_ptr(), True, dw.data_ptr(), K,
                               C, S, s)
        return True

    def _route32(self, mode, act, w, bias, addend, out, direct):
        """Run a float32 3x3 layer on the fastest of the direct kernels
        (``direct()``) and the two Winograd routes, measured once per
        geometry and kept with the launch plans.  Returns what ``direct``
        returned when it ran, else None."""
        key = ("w32", mode, tuple(act.shape), tuple(w.shape),
               bias is not None, addend is not None, _code(act))
        plan = self._conv_plans.get(key)
        # 16-bit operands: m = 2 only (the m = 4 transforms lose bits there)
        ms = (2, 4) if act.dtype == torch.float32 else (2,)
        if _CONV_WINOGRAD in ("2", "4"):
            plan = (int(_CONV_WINOGRAD) if int(_CONV_WINOGRAD) in ms else 2,
                    0)
        elif plan is None and _CONV_AUTOTUNE and not _DETERMINISTIC and \
                (_world_size() <= 1 or _rank() == 0):
            name = repr(key)
            stored = self._plan_store()[1].get(name)
            if stored is not None:
                plan = (int(stored[0]), int(stored[1]))
            elif not torch.cuda.is_current_stream_capturing():
                def run_direct():
                    direct()
                    return True
                def route(m):
                    return lambda: self._winograd_any(m, mode, act, w, bias,
                                                      addend, out)
                plan = self._measure_routes16(
                    [((0, 0), run_direct)] + [((m, 0), route(m)) for m in ms])
                if plan is not None:
                    self._persist_plan(name, plan)
            if plan is not None:
                self._conv_plans[key] = plan
        if plan is not None and plan[0] in (2, 4) and \
                self._winograd_any(plan[0], mode, act, w, bias, addend, out):
            return None
        return direct()

    def _winograd_any(self, m, mode, act, w, bias, addend, out):
        if mode == 2:       # act = dy, w = x, out = dw
            return self._winograd_wgrad(m, act, w, out)
        return self._winograd(m, mode, act, w, bias, addend, out)

    @property
    def STAT_COPIES(self):
        return self.k.stat_copies

    def conv_fwd(self, x, w, b, y, stride, pad, groups, stats=None,
                 shift=None):
        """``stats`` (float[STAT_COPIES * 2 * K + K], zero on entry) receives
        the output's per-channel sums of y - shift and (y - shift)^2 for a
        following batch normalisation: from the convolution kernel's epilogue
        where it has one, else from one pass over y."""
        w = self._contig(w)
        if stats is not None:
            assert stats.dtype == torch.float32 and \
                shift.dtype == torch.float32 and shift.is_contiguous()
        if stats is None and self._winograd_applies(x, w, y, stride, pad,
                                                    groups):
            self._route32(0, x, w, b, None, y,
                          lambda: self._conv_fwd_main(x, w, b, y, stride,
                                                      pad, groups, None,
                                                      None))
            return
        self._conv_fwd_main(x, w, b, y, stride, pad, groups, stats, shift)

    def _conv_fwd_main(self, x, w, b, y, stride, pad, groups, stats, shift):
        """The GEMM-shaped 16-bit kernels where one applies, else the
        implicit-GEMM kernels."""
        if self._conv_fwd_16bit(x, w, b, y, stride, pad, groups):
            if stats is not None:
                self.k.chan_sums(_code(y), y.data_ptr(), stats.data_ptr(),
                                 shift.data_ptr(), y.shape[0], y.shape[1],
                                 y.shape[2] * y.shape[3], self._s())
            return
        self._conv_fwd_direct(x, w, b, y, stride, pad, groups, stats, shift)

    def _conv_fwd_direct(self, x, w, b, y, stride, pad, groups, stats, shift):
        g = self._geom(x.shape, w.shape, y.shape, stride, pad, groups)
        scratch = self._split_scratch(  # noqa: F841 (keeps the buffer alive)
            g, y, w.shape[0] // groups, y.shape[0] * y.shape[2] * y.shape[3],
            groups)
        args = (x.data_ptr(), w.data_ptr(),
                b.data_ptr() if b is not None else 0, y.data_ptr())
        code = _code(x)
        key = (tuple(x.shape), tuple(w.shape), tuple(stride), tuple(pad),
               groups, code)
        self._tuned(0, g, key,
                    lambda gg: self.k.conv_fwd(code, gg, *args, self._s()))
        if stats is not None:
            # only now: the timing launches above must not add to the sums
            g.set_stats(stats.data_ptr(), shift.data_ptr())
        self.k.conv_fwd(code, g, *args, self._s())

    @staticmethod
    def _strided_pointwise16(x, w, stride, pad, groups):
        """16-bit 1x1 / stride 2 / unpadded convolution on even-sized images:
        a stride-1 problem on the gathered even positions."""
        return (_CONV_S2_GATHER and groups == 1 and
                x.dtype in (torch.float16, torch.bfloat16) and
                w.shape[2] == 1 and w.shape[3] == 1 and
                tuple(stride) == (2, 2) and tuple(pad) == (0, 0) and
                x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and
                x.is_contiguous())

    def _even_positions(self, x):
        N, C, H, W = x.shape
        xs = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype,
                         device=x.device)
        self.k.subsample2(_code(x), x.data_ptr(), xs.data_ptr(), N * C, H, W,
                          self._s())
        return xs

    def _conv_fwd_16bit(self, x, w, b, y, stride, pad, groups):
        """The 16-bit GEMM-shaped forward kernels; False when none applies."""
        if b is not None:
            return False
        if self._strided_pointwise16(x, w, stride, pad, groups):
            return self._conv_fwd_16bit(self._even_positions(x), w, None, y,
                                        (1, 1), (0, 0), 1)
        if x.dtype not in (torch.float16, torch.bfloat16) or groups != 1:
            return False
        return self._run_routes16(0, x, w, None, y, stride, pad, groups)

    def conv_bwd_data(self, dy, w, dx, stride, pad, groups, addend=None,
                      side_zero=None):
        """dx = backward-data(dy, w) [+ addend]: the sum with a gradient that
        reached the same input along another path happens in the kernel's
        epilogue instead of in a separate pass over dx.  ``side_zero`` is a
        float tensor (the layer's weight gradient) that the launch may clear
        on the side; returns True when it did."""
        w = self._contig(w)
        if addend is not None:
            assert addend.shape == dx.shape and addend.dtype == dx.dtype \
                and addend.data_ptr() != dx.data_ptr()
        if self._winograd_applies(dy, w, dx, stride, pad, groups):
            if addend is not None:
                addend = self._contig(addend)
            return self._route32(
                1, dy, w, None, addend, dx,
                lambda: self._conv_bwd_data_main(dy, w, dx, stride, pad,
                                                 groups, addend, side_zero))
        return self._conv_bwd_data_main(dy, w, dx, stride, pad, groups,
                                        addend, side_zero)

    def _conv_bwd_data_main(self, dy, w, dx, stride, pad, groups, addend,
                            side_zero):
        if self._strided_pointwise16(dx, w, stride, pad, groups) and \
                dy.is_contiguous():
            N, C, H, W = dx.shape
            dxs = torch.empty((N, C, H // 2, W // 2), dtype=dx.dtype,
                              device=dx.device)
            self.conv_bwd_data(dy, w, dxs, (1, 1), (0, 0), 1)
            ad = None if addend is None else self._contig(addend)
            self.k.upsample2_add(_code(dx), dxs.data_ptr(),
                                 ad.data_ptr() if ad is not None else 0,
                                 dx.data_ptr(), N * C, H, W, self._s())
            return
        if dx.dtype in (torch.float16, torch.bfloat16) and groups == 1 and \
                self._run_routes16(
                    1, dy, w, None if addend is None else self._contig(addend),
                    dx, stride, pad, groups):
            return
        return self._conv_bwd_data_direct(dy, w, dx, stride, pad, groups,
                                          addend, side_zero)

    def _conv_bwd_data_direct(self, dy, w, dx, stride, pad, groups, addend,
                              side_zero):
        g = self._geom(dx.shape, w.shape, dy.shape, stride, pad, groups)
        scratch = self._split_scratch(  # noqa: F841 (keeps the buffer alive)
            g, dx, dx.shape[1] // groups,
            dx.shape[0] * dx.shape[2] * dx.shape[3], groups)
        args = (dy.data_ptr(), w.data_ptr(), dx.data_ptr())
        code = _code(dx)
        key = (tuple(dx.shape), tuple(w.shape), tuple(stride), tuple(pad),
               groups, code)
        self._tuned(1, g, key,
                    lambda gg: self.k.conv_bwd_data(code, gg, *args,
                                                    self._s()))
        cleared = False
        if side_zero is not None and side_zero.dtype == torch.float32 and \
                side_zero.is_contiguous() and _CONV_SIDE_ZERO:
            g.set_side_zero(side_zero.data_ptr(), side_zero.numel())
            cleared = True
        if addend is None:
            self.k.conv_bwd_data(code, g, *args, self._s())
        else:
            assert addend.shape == dx.shape and addend.dtype == dx.dtype \
                and addend.data_ptr() != dx.data_ptr()
            addend = self._contig(addend)
            self.k.conv_bwd_data_acc(code, g, args[0], args[1],
                                     addend.data_ptr(), args[2], self._s())
        return cleared

    def _wgrad16(self, dy, x, dw, stride, pad, groups):
        """16-bit weight gradient of a 1x1 or 3x3 / pad 1 stride-1 convolution
        on the tiled 16-bit GEMM kernel (split reduction over positions,
        float partial products); False when the geometry does not fit."""
        if not (_CONV_WGRAD_GEMM and groups == 1 and
                x.dtype in (torch.float16, torch.bfloat16) and
                dw.dtype == x.dtype and tuple(stride) == (1, 1)):
            return False
        window = (dw.shape[2], dw.shape[3]) + tuple(pad)
        if window not in ((1, 1, 0, 0), (3, 3, 1, 1)) or not (
                dy.is_contiguous() and x.is_contiguous() and
                dw.is_contiguous()):
            return False
        N, C, H, W = x.shape
        K, taps = dw.shape[0], dw.shape[2] * dw.shape[3]
        assert tuple(dy.shape) == (N, K, H, W) and dw.shape[1] == C
        splits = self.k.conv_wgrad_h16_splits(N, C, K, H, W, taps)
        if not splits:
            return False
        part = torch.empty(splits * taps * K * C, dtype=torch.float32,
                           device=x.device)
        return self.k.conv_wgrad_h16(
            _code(x), N, C, K, H, W, taps, dy.data_ptr(), x.data_ptr(),
            part.data_ptr(), splits, dw.data_ptr(), self._s())

    def conv_bwd_weight(self, dy, x, dw, stride, pad, groups, zeroed=False):
        """``zeroed``: dw already holds zeros (cleared by the backward-data
        launch of the same layer)."""
        if self._strided_pointwise16(x, dw, stride, pad, groups) and \
                _CONV_WGRAD_GEMM and dw.dtype == x.dtype and \
                self.k.conv_wgrad_h16_splits(
                    x.shape[0], x.shape[1], dw.shape[0], x.shape[2] // 2,
                    x.shape[3] // 2, 1) and \
                self._wgrad16(dy, self._even_positions(x), dw, (1, 1), (0, 0),
                              1):
            return
        if dw.is_contiguous() and \
                self._winograd_applies(dy, dw, x, stride, pad, groups):
            # route timing runs the direct kernels more than once: only the
            # first of those launches may rely on dw holding zeros
            state = {"zeroed": zeroed}

            def direct():
                if self._wgrad16(dy, x, dw, stride, pad, groups):
                    return
                self._conv_bwd_weight_direct(dy, x, dw, stride, pad, groups,
                                             state["zeroed"])
                state["zeroed"] = False
            self._route32(2, dy, x, None, None, dw, direct)
            return
        if self._wgrad16(dy, x, dw, stride, pad, groups):
            return
        self._conv_bwd_weight_direct(dy, x, dw, stride, pad, groups, zeroed)

    def _conv_bwd_weight_direct(self, dy, x, dw, stride, pad, groups, zeroed):
        g = self._geom(x.shape, dw.shape, dy.shape, stride, pad, groups)
        key = (tuple(x.shape), tuple(dw.shape), tuple(stride), tuple(pad),
               groups, _code(x))
        code = _code(x)
        assert dy.dtype == x.dtype
        # the kernel reads dy / x in their own type and always writes floats
        acc = dw if dw.dtype == torch.float32 else torch.empty(
            dw.shape, dtype=torch.float32, device=dw.device)
        if dw.shape[2] * dw.shape[3] > 1:
            # zero-filled staging buffer shared by all layers of this size
            # (the launches hand it back holding zeros)
            n = acc.numel()
            buf = self._wgrad_scratch.get(n)
            if buf is None:
                buf = torch.zeros(n, dtype=torch.float32, device=acc.device)
                self._wgrad_scratch[n] = buf
            g.set_scratch(buf.data_ptr(), n)
        args = (dy.data_ptr(), x.data_ptr(), acc.data_ptr())
        wrote = self._tuned(2, g, key,
                            lambda gg: self.k.conv_bwd_weight(code, gg, *args,
                                                              self._s()))
        # timing launches leave a complete gradient in dw: not zero any more
        g.out_zeroed = bool(zeroed) and acc is dw and not wrote
        self.k.conv_bwd_weight(code, g, *args, self._s())
        if dw.dtype != torch.float32:
            self.copy(acc, dw)

    def conv_bwd_bias(self, dy, db):
        N, K = dy.shape[0], dy.shape[1]
        PQ = dy.shape[2] * dy.shape[3]
        self.k.conv_bwd_bias(_code(dy), dy.data_ptr(), db.data_ptr(), N, K, PQ,
                             self._s())

    # ---- batch norm ---------------------------------------------------------------------------------
    @staticmethod
    def _nchw(x):
        N, C = x.shape[0], x.shape[1]
        HW = 1
        for s in x.shape[2:]:
            HW *= s
        return N, C, HW

    def _f32(self, t):
        return t if t.dtype == torch.float32 else self._cast(t, torch.float32)

    def bn_fwd_train(self, x, scale, bias, rmean, rvar, factor, eps, y, mean,
                     invstd, act=0, residual=None, stats=None):
        """act: 0 none, 1 ReLU, 2 add ``residual`` then ReLU (fused into the
        normalisation pass).  With ``stats`` (filled by the producer of x, see
        conv_fwd) the statistics pass is skipped: one pass over x, which also
        clears the buffer for the next step."""
        N, C, HW = self._nchw(x)
        if stats is not None:
            assert rmean.dtype == torch.float32 and \
                rvar.dtype == torch.float32 and stats.dtype == torch.float32
            sc, bi = self._f32(scale), self._f32(bias)
            self.k.bn_fwd_train_sums(
                _code(x), x.data_ptr(), sc.data_ptr(), bi.data_ptr(),
                rmean.data_ptr(), rvar.data_ptr(), float(factor), float(eps),
                y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), N, C, HW,
                stats.data_ptr(), int(act),
                residual.data_ptr() if residual is not None else 0, self._s())
            return
        work = self._tmp(self.k.bn_workspace(N, C, HW))
        sc, bi = self._f32(scale), self._f32(bias)
        rm, rv = self._f32(rmean), self._f32(rvar)
        self.k.bn_fwd_train_act(_code(x), x.data_ptr(), sc.data_ptr(),
                                bi.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                float(factor), float(eps), y.data_ptr(),
                                mean.data_ptr(), invstd.data_ptr(), N, C, HW,
                                work.data_ptr(), int(act),
                                residual.data_ptr() if residual is not None
                                else 0, self._s())
        if rm is not rmean:
            self.copy(rm, rmean)
            self.copy(rv, rvar)

    def bn_fwd_infer(self, x, scale, bias, rmean, rvar, eps, y, act=0,
                     residual=None):
        N, C, HW = self._nchw(x)
        # converted copies must stay referenced until the launch is issued:
        # a freed temporary's storage is handed to the next one
        sc, bi = self._f32(scale), self._f32(bias)
        rm, rv = self._f32(rmean), self._f32(rvar)
        self.k.bn_fwd_infer_act(_code(x), x.data_ptr(), sc.data_ptr(),
                                bi.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                float(eps),
                                y.data_ptr(), N, C, HW, int(act),
                                residual.data_ptr() if residual is not None
                                else 0, self._s())

    def bn_bwd(self, dy, x, scale, mean, invstd, dx, dscale, dbias, act=0,
               bias=None, y=None, dres=None):
        """With act 1 the ReLU mask is recomputed from x (needs ``bias``);
        with act 2 it comes from the saved output ``y`` and the masked
        gradient is also written to ``dres``."""
        N, C, HW = self._nchw(x)
        work = self._tmp(self.k.bn_workspace(N, C, HW))
        ds = dscale if dscale.dtype == torch.float32 else self._tmp(C)
        db = dbias if dbias.dtype == torch.float32 else self._tmp(C)
        sc = self._f32(scale)
        bi = self._f32(bias) if bias is not None else None
        self.k.bn_bwd_act(_code(x), dy.data_ptr(), x.data_ptr(),
                          y.data_ptr() if y is not None else 0,
                          sc.data_ptr(),
                          bi.data_ptr() if bi is not None else 0,
                          mean.data_ptr(), invstd.data_ptr(),
                          dx.data_ptr(),
                          dres.data_ptr() if dres is not None else 0,
                          ds.data_ptr(), db.data_ptr(), N, C, HW,
                          work.data_ptr(), int(act), self._s())
        if ds is not dscale:
            self.copy(ds, dscale)
            self.copy(db, dbias)

    # ---- pooling ---------------------------------------------------------------------------------------
    def _pgeom(self, x, y, kernel, stride, pad, is_max):
        return self.k.PoolGeom(x.shape[0], x.shape[1], x.shape[2], x.shape[3],
                               y.shape[2], y.shape[3], kernel[0], kernel[1],
                               stride[0], stride[1], pad[0], pad[1],
                               1 if is_max else 0)

    def pool_fwd(self, x, y, kernel, stride, pad, is_max):
        g = self._pgeom(x, y, kernel, stride, pad, is_max)
        self.k.pool_fwd(_code(x), g, x.data_ptr(), y.data_ptr(), self._s())

    def pool_bwd(self, dy, x, y, dx, kernel, stride, pad, is_max):
        g = self._pgeom(x, y, kernel, stride, pad, is_max)
        y = self._contig(y)
        self.k.pool_bwd(_code(x), g, dy.data_ptr(), x.data_ptr(), y.data_ptr(),
                        dx.data_ptr(), self._s())

    # ---- index / gather -------------------------------------------------------------------------------------
    def _idx(self, idx):
        idx = self._contig(idx.reshape(-1))
        if idx.dtype != torch.int32:
            idx = self._cast(idx, torch.int32)
        return idx

    def index_select(self, x, axis, idx, out):
        idx = self._idx(idx)
        outer = 1
        for s in x.shape[:axis]:
            outer *= s
        inner = 1
        for s in x.shape[axis + 1:]:
            inner *= s
        self.k.index_select(_code(x), x.data_ptr(), idx.data_ptr(),
                            out.data_ptr(), outer, x.shape[axis], idx.numel(),
                            inner, self._s())

    def index_add(self, dy, axis, idx, out):
        idx = self._idx(idx)
        outer = 1
        for s in out.shape[:axis]:
            outer *= s
        inner = 1
        for s in out.shape[axis + 1:]:
            inner *= s
        d = self._f32c(dy)
        tgt = out if out.dtype == torch.float32 else \
            torch.empty(out.shape, dtype=torch.float32, device=out.device)
        self.k.index_add(0, d.data_ptr(), idx.data_ptr(), tgt.data_ptr(), outer,
                         out.shape[axis], idx.numel(), inner, self._s())
        if tgt is not out:
            self.copy(tgt, out)

    def gather_rows(self, table, idx, out):
        self.index_select(table, 0, idx, out.reshape(-1, table.shape[1]))

    def scatter_add_rows(self, grad, idx, dtable):
        self.index_add(grad.reshape(-1, dtable.shape[1]), 0, idx, dtable)

    # ---- collectives support (reference src/io/communicator.cc) ----------------------------
    def pack_plan(self, views):
        """(pointers, element counts, total) of the tensors of one bucket, in
        bucket order.  Host data only: the kernels receive the entries in
        their argument block."""
        return ([v.data_ptr() for v in views], [v.numel() for v in views],
                sum(v.numel() for v in views))

    def pack(self, plan, views, bucket):
        ptrs, counts, _ = plan
        assert all(v.dtype == views[0].dtype for v in views), \
            "a fusion bucket holds tensors of one dtype"
        self.k.pack(_code(views[0]), _code(bucket), ptrs, counts,
                    bucket.data_ptr(), self._s())

    def unpack(self, plan, bucket, views, scale):
        ptrs, counts, _ = plan
        assert all(v.dtype == views[0].dtype for v in views), \
            "a fusion bucket holds tensors of one dtype"
        self.k.unpack(_code(bucket), _code(views[0]), ptrs, counts,
                      bucket.data_ptr(), float(scale), self._s())

    def sparse_select(self, x, thr, topk, payload, cap, work):
        """Fill ``payload`` (int32, 1 + 2 * cap words: count, indices, float
        bits) with the entries of ``x`` that are exchanged: |x| >= thr, or the
        ceil(thr * n) entries of largest magnitude.  Device only: the count is
        never read by the host."""
        assert x.dtype == torch.float32 and x.is_contiguous()
        assert payload.dtype == torch.int32 and payload.numel() >= 1 + 2 * cap
        n = x.numel()
        k = min(cap, max(1, int(math.ceil(thr * n)))) if topk else 0
        self.k.sparse_select(x.data_ptr(), n, float(thr), 1 if topk else 0, k,
                             payload.data_ptr(), int(cap), work.data_ptr(),
                             self._s())

    def sparse_workspace(self):
        return self._tmp(self.k.topk_abs_workspace(), torch.int32)

    def sparse_accumulate(self, payloads, nseg, cap, dense, sign=1.0):
        """dense[idx] += sign * value over ``nseg`` payloads laid end to end."""
        assert dense.dtype == torch.float32 and dense.is_contiguous()
        self.k.scatter_add_segments(payloads.data_ptr(), int(nseg), int(cap),
                                    float(sign), dense.data_ptr(),
                                    dense.numel(), self._s())

    def sparsify_threshold(self, x, thr):
        """(values, indices, count) of the entries with |x| >= thr, compacted
        on the device; buffers hold up to x.numel() entries."""
        n = x.numel()
        vals = self._tmp(n)
        idx = self._tmp(n, torch.int32)
        cnt = self._tmp(1, torch.int32)
        self.k.sparsify_threshold(x.data_ptr(), float(thr), vals.data_ptr(),
                                  idx.data_ptr(), cnt.data_ptr(), n, n,
                                  self._s())
        return vals, idx, cnt

    def topk_abs(self, x, k):
        """The k entries of largest magnitude (values with their sign, and
        int32 positions), in no particular order: radix select on the device,
        no host synchronisation."""
        assert x.dtype == torch.float32 and x.is_contiguous()
        k = int(k)
        vals = self._tmp(k)
        idx = self._tmp(k, torch.int32)
        work = self._tmp(self.k.topk_abs_workspace(), torch.int32)
        self.k.topk_abs(x.data_ptr(), x.numel(), k, vals.data_ptr(),
                        idx.data_ptr(), work.data_ptr(), self._s())
        return vals, idx

    def scatter_add_sparse(self, vals, idx, cnt, dense):
        self.k.scatter_add_sparse(vals.data_ptr(), idx.data_ptr(),
                                  cnt.data_ptr() if cnt is not None else 0,
                                  vals.numel(), dense.data_ptr(), dense.numel(),
                                  self._s())

    # ---- fused optimiser step -------------------------------------------------------------------------------------
    def sgd_update(self, p, g, buf, lr, momentum, dampening, weight_decay,
                   nesterov, scale):
        g = self._contig(g)
        if g.dtype != p.dtype:
            g = self._cast(g, p.dtype)
        # keep the float copies of 16-bit hyper-parameters alive until launch
        hyper = [self._f32(t) for t in (lr, momentum, dampening, weight_decay)]
        self.k.sgd_update(_code(p), p.data_ptr(), g.data_ptr(),
                          buf.data_ptr() if buf is not None else 0,
                          hyper[0].data_ptr(), hyper[1].data_ptr(),
                          hyper[2].data_ptr(), hyper[3].data_ptr(),
                          1 if nesterov else 0, float(scale), p.numel(),
                          self._s())

    def sgd_update_multi(self, ps, gs, bufs, lr, momentum, dampening,
                         weight_decay, nesterov, scale, masters=None):
        """One launch per 48 tensors; all tensors share the dtype of ps[0].
        With ``masters`` (float copies of 16-bit parameters) the step is
        taken in float -- ``bufs`` are float then -- and rounded into ps."""
        if masters:
            assert all(m.dtype == torch.float32 for m in masters)
            assert not bufs or all(b.dtype == torch.float32 for b in bufs)
        gg = []
        for p, g in zip(ps, gs):
            g = self._contig(g)
            if g.dtype != p.dtype:
                g = self._cast(g, p.dtype)
            gg.append(g)
        # keep the float copies of 16-bit hyper-parameters alive until launch
        hyper = [self._f32(t) for t in (lr, momentum, dampening, weight_decay)]
        self.k.sgd_update_multi(
            _code(ps[0]), [p.data_ptr() for p in ps],
            [g.data_ptr() for g in gg],
            [b.data_ptr() for b in bufs] if bufs else [],
            [m.data_ptr() for m in masters] if masters else [],
            [p.numel() for p in ps], hyper[0].data_ptr(), hyper[1].data_ptr(),
            hyper[2].data_ptr(), hyper[3].data_ptr(), 1 if nesterov else 0,
            float(scale), self._s())

    def adaptive_update_multi(self, rule, ps, gs, s1, s2, lr, wd, eps, a, b,
                              step, scale):
        """RMSProp (rule 0) / AdaGrad (1) / Adam (2) on a list of parameters
        of one dtype in one launch per 48 tensors; states have the
        parameters' dtype, hyper-parameters are one-element tensors."""
        gg = []
        for p, g in zip(ps, gs):
            g = self._contig(g)
            if g.dtype != p.dtype:
                g = self._cast(g, p.dtype)
            gg.append(g)
        hyper = [self._f32(t) if t is not None else None
                 for t in (lr, wd, eps, a, b, step)]
        ptr = [t.data_ptr() if t is not None else 0 for t in hyper]
        self.k.adaptive_update_multi(
            _code(ps[0]), int(rule), [p.data_ptr() for p in ps],
            [g.data_ptr() for g in gg], [t.data_ptr() for t in s1],
            [t.data_ptr() for t in s2] if s2 else [],
            [p.numel() for p in ps], ptr[0], ptr[1], ptr[2], ptr[3], ptr[4],
            ptr[5], float(scale), self._s())

    # ---- layer norm / gelu ------------------------------------------------------
    def layernorm_fwd(self, x, gamma, beta, y, mean, rstd, eps):
        """Rows = all leading dimensions, normalised over the last one."""
        D = x.shape[-1]
        rows = x.numel() // D
        x = self._contig(x)
        g = self._contig(gamma) if gamma is not None else None
        b = self._contig(beta) if beta is not None else None
        self.k.layernorm_fwd(_code(x), x.data_ptr(),
                             g.data_ptr() if g is not None else 0,
                             b.data_ptr() if b is not None else 0,
                             y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                             rows, D, float(eps), self._s())

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dx, dgamma, dbeta):
        D = x.shape[-1]
        rows = x.numel() // D
        dy, x = self._contig(dy), self._contig(x)
        g = self._contig(gamma) if gamma is not None else None
        want = dgamma is not None
        if want and _LN_PARAM_ONE_LAUNCH and not _DETERMINISTIC and \
                dgamma.is_contiguous() and \
                dbeta.is_contiguous() and dgamma.dtype == dbeta.dtype and \
                dgamma.dtype in (torch.float32, x.dtype):
            # dx, then both parameter gradients in one launch that writes
            # them in their own type (no fill before, no cast behind)
            self.k.layernorm_bwd(_code(x), dy.data_ptr(), x.data_ptr(),
                                 g.data_ptr() if g is not None else 0,
                                 mean.data_ptr(), rstd.data_ptr(),
                                 dx.data_ptr(), 0, 0, rows, D, self._s())
            sums, tickets = self._zero_scratch()
            if self.k.layernorm_bwd_param(
                    _code(x), dy.data_ptr(), x.data_ptr(), mean.data_ptr(),
                    rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                    0 if dgamma.dtype == torch.float32 else 1,
                    sums.data_ptr(), sums.numel(), tickets.data_ptr(),
                    tickets.numel(), rows, D, self._s()):
                return
            # scratch too small: the float path below repeats dx as well
        # the parameter gradients are accumulated in float
        fg = fb = None
        if want:
            fg = dgamma if dgamma.dtype == torch.float32 else self._tmp(D)
            fb = dbeta if dbeta.dtype == torch.float32 else self._tmp(D)
        self.k.layernorm_bwd(_code(x), dy.data_ptr(), x.data_ptr(),
                             g.data_ptr() if g is not None else 0,
                             mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                             fg.data_ptr() if want else 0,
                             fb.data_ptr() if want else 0, rows, D, self._s())
        if want and fg is not dgamma:
            self.copy(fg, dgamma)
            self.copy(fb, dbeta)

    def gelu_fwd(self, x, y):
        x = self._contig(x)
        self.k.gelu_fwd(_code(x), x.data_ptr(), y.data_ptr(), x.numel(),
                        self._s())

    def gelu_bwd(self, dy, x, dx):
        dy, x = self._contig(dy), self._contig(x)
        self.k.gelu_bwd(_code(x), dy.data_ptr(), x.data_ptr(), dx.data_ptr(),
                        x.numel(), self._s())

    # ---- fused attention ----------------------------------------------------------------
    def _attn_desc(self, q, kt, v, mask, scale):
        """q (B, H, Sq, D), kt (B, H, D, Sk), v (B, H, Sk, D): strided views
        with the head dimension contiguous (copies are made otherwise)."""
        B, H, Sq, D = q.shape
        Sk = v.shape[2]
        k = kt.permute(0, 1, 3, 2)                 # (B, H, Sk, D) view
        keep = []

        def rows(t):
            if t.stride(3) != 1 or any(st % 4 for st in t.stride()[:3]):
                t = t.contiguous()
                keep.append(t)
            return t

        q, k, v = rows(q), rows(k), rows(v)
        ms = [0, 0, 0, 0]
        mptr = 0
        if mask is not None:
            if mask.dtype != q.dtype:
                mask = self._cast(mask, q.dtype)
                keep.append(mask)
            while mask.dim() < 4:
                mask = mask.unsqueeze(0)
            mask = mask.expand(B, H, Sq, Sk)
            ms = list(mask.stride())
            mptr = mask.data_ptr()
        desc = self.k.AttnDesc(q.data_ptr(), k.data_ptr(), v.data_ptr(), mptr,
                               list(q.stride()[:3]), list(k.stride()[:3]),
                               list(v.stride()[:3]), ms, B, H, Sq, Sk, D,
                               float(scale))
        return desc, keep

    def attention_fwd(self, q, kt, v, mask, scale, out, lse):
        """out (B, Sq, H, D) contiguous, lse (B, H, Sq) float32."""
        desc, keep = self._attn_desc(q, kt, v, mask, scale)
        self.k.attention_fwd(_code(q), desc, out.data_ptr(), lse.data_ptr(),
                             self._s())

    def attention_bwd(self, dout, q, kt, v, mask, scale, out, lse, dq, dk,
                      dv):
        """dout / out / dq / dk / dv: contiguous (B, S, H, D)."""
        desc, keep = self._attn_desc(q, kt, v, mask, scale)
        dout = self._contig(dout)
        delta = self._tmp(lse.numel())
        self.k.attention_bwd(_code(q), desc, out.data_ptr(), dout.data_ptr(),
                             lse.data_ptr(), delta.data_ptr(), dq.data_ptr(),
                             dk.data_ptr(), dv.data_ptr(), self._s())

    # ---- recurrent cells ---------------------------------------------------------------------------------------------
    def lstm_cell_fwd(self, gates, c_prev, c, h, acts):
        self.k.lstm_cell_fwd(_code(gates), gates.data_ptr(), c_prev.data_ptr(),
                             c.data_ptr(), h.data_ptr(), acts.data_ptr(),
                             c_prev.shape[0], c_prev.shape[1], self._s())

    def lstm_cell_bwd(self, dh, dc_next, acts, c_prev, c, dgates, dc_prev):
        self.k.lstm_cell_bwd(_code(acts), dh.data_ptr(), dc_next.data_ptr(),
                             acts.data_ptr(), c_prev.data_ptr(), c.data_ptr(),
                             dgates.data_ptr(), dc_prev.data_ptr(),
                             c_prev.shape[0], c_prev.shape[1], self._s())

    def gru_cell_fwd(self, gx, gh, brn, h_prev, h, acts, ghn):
        """One GRU step from the (batch, 3H) input and recurrent projections
        (gates r, z, n); ``brn`` (H,) or None is added to gh_n inside the
        reset-gated term."""
        self.k.gru_cell_fwd(_code(gx), gx.data_ptr(), gh.data_ptr(),
                            0 if brn is None else brn.data_ptr(),
                            h_prev.data_ptr(), h.data_ptr(), acts.data_ptr(),
                            ghn.data_ptr(), h_prev.shape[0], h_prev.shape[1],
                            self._s())

    def gru_cell_bwd(self, dh, acts, ghn, h_prev, dgx, dgh, dh_prev):
        self.k.gru_cell_bwd(_code(acts), dh.data_ptr(), acts.data_ptr(),
                            ghn.data_ptr(), h_prev.data_ptr(), dgx.data_ptr(),
                            dgh.data_ptr(), dh_prev.data_ptr(),
                            h_prev.shape[0], h_prev.shape[1], self._s())

    def rnn_act_bwd(self, mode, dh, h, d):
        """d = dh * act'(h); mode 0 relu, 1 tanh."""
        self.k.rnn_act_bwd(_code(h), int(mode), dh.data_ptr(), h.data_ptr(),
                           d.data_ptr(), h.numel(), self._s())

    def lstm_seq_supported(self, batch, hidden, dtype):
        return dtype == torch.float32 and \
            self.k.lstm_seq_supported(int(batch), int(hidden))

    def lstm_seq_fwd(self, gx, R, h0, c0, hseq, cseq, acts, bar, reverse):
        """Whole-sequence LSTM forward of one layer / direction in one launch;
        fills hseq, cseq (T, B, H) and acts (T, B, 4H)."""
        T, B, H = hseq.shape
        for t in (gx, R, h0, c0, hseq, cseq, acts):
            assert t.is_contiguous() and t.dtype == torch.float32
        assert tuple(gx.shape) == (T, B, 4 * H) and tuple(R.shape) == (4 * H, H)
        assert tuple(h0.shape) == (B, H) == tuple(c0.shape)
        assert tuple(acts.shape) == (T, B, 4 * H) and bar.numel() >= 2
        self.k.lstm_seq_fwd(gx.data_ptr(), R.data_ptr(), h0.data_ptr(),
                            c0.data_ptr(), hseq.data_ptr(), cseq.data_ptr(),
                            acts.data_ptr(), bar.data_ptr(), B, H, T,
                            1 if reverse else 0, self._s())

    def lstm_seq_bwd(self, R, c0, cseq, acts, dy, dhy, dcy, dgates, dhx, dcx,
                     bar, reverse):
        """Whole-sequence LSTM backward: dy is a (T, B, H) view whose rows may
        be strided (one direction of a bidirectional output); fills dgates
        (T, B, 4H), dhx and dcx (B, H)."""
        T, B, H = cseq.shape
        for t in (R, c0, cseq, acts, dhy, dcy, dgates, dhx, dcx):
            assert t.is_contiguous() and t.dtype == torch.float32
        assert tuple(dy.shape) == (T, B, H) and dy.stride(2) == 1 and \
            dy.stride(0) == B * dy.stride(1) and dy.dtype == torch.float32
        assert tuple(dgates.shape) == (T, B, 4 * H) == tuple(acts.shape)
        assert tuple(dhy.shape) == (B, H) == tuple(dcy.shape)
        assert tuple(dhx.shape) == (B, H) == tuple(dcx.shape)
        slabs = torch.empty(2 * (H // 8) * B * H, dtype=torch.float32,
                            device=cseq.device)
        self.k.lstm_seq_bwd(R.data_ptr(), c0.data_ptr(), cseq.data_ptr(),
                            acts.data_ptr(), dy.data_ptr(), dy.stride(1),
                            dhy.data_ptr(), dcy.data_ptr(), dgates.data_ptr(),
                            dhx.data_ptr(), dcx.data_ptr(), slabs.data_ptr(),
                            bar.data_ptr(), B, H, T, 1 if reverse else 0,
                            self._s())
