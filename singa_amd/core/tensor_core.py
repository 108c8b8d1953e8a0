"""The core tensor (what the Python API layer calls ``CTensor``) and the free
functions operating on it.

A tensor is (shape, stride, dtype, device, block).  Strides are in elements; a
stride of 0 marks a broadcast dimension and non-monotonic strides mark a
transposed view, so ``Transpose``/``Broadcast`` never move data.  Blocks are
allocated lazily by the owning device.  Every function that touches data is
submitted to the device as a named op with explicit read / write blocks, which
is what lets the device record a whole training step and replay it.

Behaviour parity: reference include/singa/core/tensor.h:55-664,
src/core/tensor/tensor.cc (semantics of every function are cited where they
are not obvious); the Python-visible names follow src/api/core_tensor.i:84-391.
"""
import numpy as np
import os

import torch

from . import dtypes
from .dtypes import kFloat32, kFloat16, kInt, kChar, kDouble, kBFloat16
from .device import get_default_device, Device


def Product(shape, start=0, end=None):
    shape = list(shape)
    if end is None:
        end = len(shape)
    if len(shape) == 0:
        return 0
    p = 1
    for s in shape[start:end]:
        p *= int(s)
    return p


def _contig_strides(shape):
    st, acc = [], 1
    for s in reversed(shape):
        st.append(acc)
        acc *= int(s)
    return list(reversed(st))


class Tensor(object):
    """Core tensor.  ``Tensor(other)`` is a shallow copy sharing the block."""

    # _stat_source: set on a convolution's output, see nn_ops.ChannelStatLink
    __slots__ = ("_shape", "_stride", "_dtype", "_dev", "_block",
                 "_stat_source", "__weakref__")

    def __init__(self, shape=None, device=None, dtype=kFloat32):
        if isinstance(shape, Tensor):  # copy constructor
            src = shape
            self._shape = list(src._shape)
            self._stride = list(src._stride)
            self._dtype = src._dtype
            self._dev = src._dev
            self._block = src._block
            if self._block is not None:
                self._block.IncRefCount()
            return
        if shape is None:
            shape = []
        self._shape = [int(s) for s in shape]
        self._stride = _contig_strides(self._shape)
        self._dtype = int(dtype)
        self._dev = device if device is not None else get_default_device()
        n = Product(self._shape)
        self._block = self._dev.NewBlock(n * dtypes.SizeOf(self._dtype))
        self._block.IncRefCount()

    def __del__(self):
        try:
            b = self._block
            if b is not None:
                b.DecRefCount()
        except Exception:
            pass

    # ---- metadata -------------------------------------------------------------
    def shape(self, idx=None):
        if idx is None:
            return list(self._shape)
        return self._shape[idx]

    def stride(self):
        return list(self._stride)

    def nDim(self):
        return len(self._shape)

    def Size(self):
        return Product(self._shape)

    def MemSize(self):
        return self._block.size() if self._block is not None else 0

    def data_type(self):
        return self._dtype

    def device(self):
        return self._dev

    def block(self):
        return self._block

    def initialized(self):
        return self._block is not None and self._block.initialized()

    def transpose(self):
        """True when the strides are not those of a plain row-major layout
        (reference tensor.h: strides not monotonically decreasing)."""
        if len(self._stride) > 1:
            prev = None
            for s, d in zip(self._stride, self._shape):
                if s == 0:
                    continue
                if prev is not None and s > prev:
                    return True
                prev = s
        return False

    def broadcasted(self):
        return any(s == 0 for s in self._stride)

    def is_contiguous(self):
        return self._stride == _contig_strides(self._shape)

    # ---- storage access (used by op bodies) -------------------------------------
    def _storage(self):
        b = self._block
        d = b.data
        if d is None:
            d = self._dev.alloc(b, self._dtype)
        return d

    def view(self):
        """torch view with this tensor's shape and strides."""
        st = self._storage()
        if not self._shape:
            return st[:0]
        # one call for every layout (a slice + view pair costs four times
        # as much on the dispatch path of each operator)
        return torch.as_strided(st, self._shape, self._stride)

    def flat(self):
        """Contiguous 1-d view of the whole block."""
        return self._storage()

    # ---- shape manipulation (in place on the handle) -------------------------------
    def _assign(self, other):
        """Make this handle refer to ``other``'s data (operator= semantics)."""
        if other._block is not None:
            other._block.IncRefCount()
        if self._block is not None:
            self._block.DecRefCount()
        self._shape = list(other._shape)
        self._stride = list(other._stride)
        self._dtype = other._dtype
        self._dev = other._dev
        self._block = other._block

    def Reshape(self, shape):
        r = Reshape(self, shape)
        self._assign(r)
        return self

    def Resize(self, shape):
        shape = [int(s) for s in shape]
        if Product(shape) != Product(self._shape):
            t = Tensor(shape, self._dev, self._dtype)
            self._assign(t)
        else:
            self._shape = shape
            self._stride = _contig_strides(shape)
        return self

    def ResetLike(self, other):
        if (self._block is None or self._dev is not other._dev or
                self.MemSize() != other.MemSize() or
                self._dtype != other._dtype):
            t = Tensor(other._shape, other._dev, other._dtype)
            self._assign(t)
        self._shape = list(other._shape)
        self._stride = list(other._stride)
        return self

    def T(self):
        assert len(self._shape) == 2, "T() works on 2-d tensors only"
        return DefaultTranspose(self)

    # ---- type / placement -------------------------------------------------------------
    def AsType(self, dtype):
        """New tensor with the requested type (same tensor when equal)."""
        dtype = int(dtype)
        if dtype == self._dtype:
            return self
        out = Tensor(self._shape, self._dev, dtype)
        _submit(self._dev, "CastType",
                lambda s, o: s._dev.math.copy(s.view(), o.view()),
                [self], [out])
        return out

    def ToType(self, dtype):
        """Change the type in place (the handle is re-pointed to new data)."""
        t = self.AsType(dtype)
        if t is not self:
            self._assign(t)
        return self

    def ToDevice(self, dev):
        if dev is self._dev:
            return self
        out = Tensor(self._shape, dev, self._dtype)
        out._stride = list(self._stride)
        if self.initialized():
            src_flat = self.flat()
            # cross-device moves are executed immediately on both devices
            out._dev.alloc(out._block, out._dtype)
            out.flat().copy_(src_flat, non_blocking=False)
        self._assign(out)
        return self

    def ToHost(self):
        return self.ToDevice(get_default_device())

    def Clone(self, device=None):
        dev = device if device is not None else self._dev
        out = Tensor(self._shape, dev, self._dtype)
        out._stride = list(self._stride)
        out.CopyData(self)
        return out

    def Contiguous(self):
        r = Contiguous(self)
        if r is not self:
            self._assign(r)
        return self

    # ---- data movement -------------------------------------------------------------------
    def CopyData(self, src):
        """Raw block copy (layout is carried over unchanged)."""
        assert src.Size() == self.Size() or src.MemSize() == self.MemSize(), \
            "size mismatch in CopyData"
        if src._dev is self._dev:
            def body(s, d):
                n = min(s.flat().numel(), d.flat().numel())
                d._dev.math.copy(s.flat()[:n], d.flat()[:n])
            _submit(self._dev, "CopyData", body, [src], [self])
        else:
            if not src.initialized():
                return
            n = min(src.flat().numel(), self._storage().numel())
            self.flat()[:n].copy_(src.flat()[:n])

    def _copy_from_numpy(self, arr, tdtype):
        arr = np.ascontiguousarray(arr).reshape(-1)
        n = self.Size()
        assert arr.size >= n, "host buffer smaller than the tensor"
        t = torch.from_numpy(arr[:n].copy())
        if t.dtype != tdtype:
            t = t.to(tdtype)
        # host to device copies are never recorded: a recorded training step
        # reads whatever the input blocks hold at replay time
        self._storage()[:n].copy_(t.to(dtypes.to_torch(self._dtype)))

    def CopyFloatDataFromHostPtr(self, arr):
        self._copy_from_numpy(np.asarray(arr, dtype=np.float32), torch.float32)

    def CopyIntDataFromHostPtr(self, arr):
        self._copy_from_numpy(np.asarray(arr, dtype=np.int32), torch.int32)

    def CopyHalfFloatDataFromHostPtr(self, arr):
        self._copy_from_numpy(np.asarray(arr, dtype=np.float16), torch.float16)

    def CopyDoubleDataFromHostPtr(self, arr):
        self._copy_from_numpy(np.asarray(arr, dtype=np.float64), torch.float64)

    def _to_numpy(self, np_dtype, n=None):
        if not self.initialized():
            self._storage()
        v = self.view() if self._shape else self.flat()[:0]
        v = v.detach()
        if v.dtype == torch.bfloat16:
            v = v.float()
        arr = v.to("cpu").contiguous().numpy().reshape(-1)
        check = getattr(self._dev, "check_flags", None)
        if check is not None:
            check()
        if n is not None:
            arr = arr[:n]
        return arr.astype(np_dtype, copy=True)

    def GetFloatValue(self, n=None):
        return self._to_numpy(np.float32, n)

    def GetIntValue(self, n=None):
        return self._to_numpy(np.int32, n)

    def GetHalfFloatValue(self, n=None):
        return self._to_numpy(np.float16, n)

    def GetDoubleValue(self, n=None):
        return self._to_numpy(np.float64, n)

    def get_value(self, n=None):
        return self._to_numpy(dtypes.to_numpy_dtype(self._dtype), n)

    def SetFloatValue(self, x):
        x = float(x)
        _submit(self._dev, "SetValue",
                lambda o: o._dev.math.fill(o.flat(), x), [], [self])

    SetValue = SetFloatValue
    SetIntValue = SetFloatValue

    def Repeat(self, repeats, axis, device=None):
        return _repeat(self, list(repeats), axis)

    def RepeatData(self, repeats, axis, total_repeats, src):
        r = _repeat(src, list(repeats), axis)
        self.CopyData(r)

    # ---- norms (host results, unsafe inside a recorded graph) ----------------------------------
    def L1(self):
        _no_host_result_while_recording(self._dev, "L1")
        n = max(self.Size(), 1)
        return self._dev.math.asum(self.view()) / n

    def l1(self):
        return self.L1()

    def L2(self):
        _no_host_result_while_recording(self._dev, "L2")
        n = max(self.Size(), 1)
        return self._dev.math.nrm2(self.view()) / n

    def l2(self):
        return self.L2()

    # ---- arithmetic assignment -----------------------------------------------------------
    def __iadd__(self, x):
        if isinstance(x, Tensor):
            Add(self, x, self)
        else:
            AddFloatWithRet(self, float(x), self)
        return self

    def __isub__(self, x):
        if isinstance(x, Tensor):
            Sub(self, x, self)
        else:
            SubFloatWithRet(self, float(x), self)
        return self

    def __imul__(self, x):
        if isinstance(x, Tensor):
            EltwiseMult(self, x, self)
        else:
            EltwiseMultFloatWithRet(self, float(x), self)
        return self

    def __itruediv__(self, x):
        if isinstance(x, Tensor):
            Div(self, x, self)
        else:
            DivFloatWithRet(self, float(x), self)
        return self

    __idiv__ = __itruediv__

    def __repr__(self):
        return "CTensor(shape=%s, stride=%s, dtype=%s, dev=%d)" % (
            self._shape, self._stride, dtypes.name(self._dtype), self._dev.id())


# ---------------------------------------------------------------------------
# op submission
# ---------------------------------------------------------------------------
def _submit(dev, name, body, reads, writes):
    """Run ``body(*reads, *writes)`` now, or record it on the device graph.

    When recording, the closure keeps shallow copies of its operands: they pin
    the blocks and raise ``Block.ref_count`` by exactly the number the
    scheduler's ``graph_ref`` accounting expects, so that blocks referenced
    only by the graph can be recycled after their last use."""
    if dev.graph_enabled():
        r = [Tensor(t) for t in reads]
        w = [Tensor(t) for t in writes]
        dev.Exec(lambda: body(*r, *w), [t._block for t in r],
                 [t._block for t in w], name)
    else:
        body(*reads, *writes)


def SizeOf(dtype):
    return dtypes.SizeOf(dtype)


# ---------------------------------------------------------------------------
# views
# ---------------------------------------------------------------------------
def Transpose(t, axes=None):
    out = Tensor(t)
    if axes is None:
        out._shape.reverse()
        out._stride.reverse()
    else:
        axes = [int(a) for a in axes]
        assert len(axes) == len(t._shape), \
            "Transpose axes's length should be equal to shape"
        out._shape = [t._shape[a] for a in axes]
        out._stride = [t._stride[a] for a in axes]
    return out


def DefaultTranspose(t):
    return Transpose(t, None)


def Broadcast(t, shape, ignore_last_dim=0):
    """View of ``t`` expanded against ``shape`` (numpy rules, trailing
    ``ignore_last_dim`` dimensions untouched).  reference tensor.cc:481-509."""
    out = Tensor(t)
    shape = list(shape)
    m, n = len(out._shape) - 1, len(shape) - 1
    if ignore_last_dim < min(m, n) + 1:
        for i in range(ignore_last_dim, min(m, n) + 1):
            if shape[n - i] != out._shape[m - i] and shape[n - i] != 1:
                assert out._shape[m - i] == 1, (
                    "cannot broadcast %s against %s" % (t._shape, shape))
                out._shape[m - i] = shape[n - i]
                out._stride[m - i] = 0
    if m < n:
        for i in range(m + 1, n + 1):
            out._shape.insert(0, shape[n - i])
            out._stride.insert(0, 0)
    return out


def Contiguous(t):
    if t.is_contiguous():
        return t
    out = Tensor(t._shape, t._dev, t._dtype)
    _submit(t._dev, "Transform",
            lambda s, o: s._dev.math.copy(s.view(), o.view()), [t], [out])
    return out


def Reshape(t, shape):
    shape = [int(s) for s in shape]
    assert Product(shape) == Product(t._shape) or (
        len(shape) == 0 and Product(t._shape) <= 1), (
        "Reshape volume mismatch %s -> %s" % (t._shape, shape))
    src = t if t.is_contiguous() else Contiguous(t)
    out = Tensor(src)
    out._shape = shape
    out._stride = _contig_strides(shape)
    return out


# ---------------------------------------------------------------------------
# element-wise
# ---------------------------------------------------------------------------
def _unary(name, op):
    def fn(x, out=None):
        ret = out if out is not None else Tensor(x._shape, x._dev, x._dtype)
        _submit(x._dev, name,
                lambda a, o: a._dev.math.unary(op, a.view(), o.view()),
                [x], [ret])
        return ret
    fn.__name__ = name
    return fn


Abs = _unary("Abs", "abs")
Exp = _unary("Exp", "exp")
Log = _unary("Log", "log")
Sqrt = _unary("Sqrt", "sqrt")
Square = _unary("Square", "square")
Sign = _unary("Sign", "sign")
Ceil = _unary("Ceil", "ceil")
Floor = _unary("Floor", "floor")
Round = _unary("Round", "round")
RoundE = _unary("RoundE", "rounde")
Sigmoid = _unary("Sigmoid", "sigmoid")
Tanh = _unary("Tanh", "tanh")
ReLU = _unary("ReLU", "relu")
Erf = _unary("Erf", "erf")
SoftPlus = _unary("SoftPlus", "softplus")
SoftSign = _unary("SoftSign", "softsign")
Cos = _unary("Cos", "cos")
Cosh = _unary("Cosh", "cosh")
Acos = _unary("Acos", "acos")
Acosh = _unary("Acosh", "acosh")
Sin = _unary("Sin", "sin")
Sinh = _unary("Sinh", "sinh")
Asin = _unary("Asin", "asin")
Asinh = _unary("Asinh", "asinh")
Tan = _unary("Tan", "tan")
Atan = _unary("Atan", "atan")
Atanh = _unary("Atanh", "atanh")


def Clamp(x, low, high, out=None):
    ret = out if out is not None else Tensor(x._shape, x._dev, x._dtype)
    low, high = float(low), float(high)
    _submit(x._dev, "Clamp",
            lambda a, o: a._dev.math.clamp(a.view(), low, high, o.view()),
            [x], [ret])
    return ret


def Transform(x, out=None):
    """Materialise ``x`` (possibly transposed / broadcast) contiguously."""
    ret = out if out is not None else Tensor(x._shape, x._dev, x._dtype)
    _submit(x._dev, "Transform",
            lambda s, o: s._dev.math.copy(s.view(), o.view()), [x], [ret])
    return ret


def _binary_into(name, op, lhs, rhs, ret):
    if lhs._shape != rhs._shape:
        l2 = Broadcast(lhs, rhs._shape)
        r2 = Broadcast(rhs, lhs._shape)
        assert l2._shape == ret._shape, (
            "%s: broadcast shape %s does not match output %s" %
            (name, l2._shape, ret._shape))
        lhs, rhs = l2, r2
    else:
        assert lhs._shape == ret._shape
    _submit(ret._dev, name,
            lambda a, b, o: o._dev.math.binary(op, a.view(), b.view(), o.view()),
            [lhs, rhs], [ret])


def _binary(name, op):
    """Tensor (x) tensor with numpy broadcasting.  Mixed int / float operands
    are computed in float32; int (x) int returns int (reference
    tensor.cc:1040-1092)."""
    def fn_new(lhs, rhs):
        fl = (dtypes.is_float(lhs._dtype), dtypes.is_float(rhs._dtype))
        both_int = not fl[0] and not fl[1]
        if lhs._dtype != rhs._dtype or both_int:
            if fl[0] and fl[1]:
                tgt = lhs._dtype
            else:
                tgt = kFloat32 if not (fl[0] or fl[1]) else (
                    lhs._dtype if fl[0] else rhs._dtype)
            a, b = lhs.AsType(tgt), rhs.AsType(tgt)
        else:
            a, b = lhs, rhs
        if a._shape != b._shape:
            shape = Broadcast(a, b._shape)._shape
        else:
            shape = a._shape
        ret = Tensor(shape, a._dev, a._dtype)
        _binary_into(name, op, a, b, ret)
        if both_int:
            return ret.AsType(kInt)
        return ret

    def fn_ret(lhs, rhs, ret):
        _binary_into(name, op, lhs, rhs, ret)
        return ret
    fn_new.__name__ = name
    return fn_new, fn_ret


__add__, Add = _binary("Add", "add")
__sub__, Sub = _binary("Sub", "sub")
__mul__, EltwiseMult = _binary("EltwiseMult", "mul")
__div__, Div = _binary("Div", "div")
Pow, PowWithRet = _binary("Pow", "pow")
__lt__, LT = _binary("LT", "lt")
__le__, LE = _binary("LE", "le")
__gt__, GT = _binary("GT", "gt")
__ge__, GE = _binary("GE", "ge")
__eq__, EQ = _binary("EQ", "eq")
ReLUBackward, ReLUBackwardWithRet = _binary("ReLUBackward", "relu_bwd")


def _scalar(name, op):
    def fn_ret(x, s, ret):
        s = float(s)
        _submit(ret._dev, name,
                lambda a, o: o._dev.math.binary_scalar(op, a.view(), s, o.view()),
                [x], [ret])
        return ret

    def fn_new(x, s):
        # an int tensor combined with a float scalar gives a float tensor
        # (reference tensor.cc, GenBinaryScalarFunction)
        if not dtypes.is_float(x._dtype):
            x = x.AsType(kFloat32)
        return fn_ret(x, s, Tensor(x._shape, x._dev, x._dtype))
    fn_new.__name__ = name
    return fn_new, fn_ret


AddFloat, AddFloatWithRet = _scalar("AddFloat", "add")
SubFloat, SubFloatWithRet = _scalar("SubFloat", "sub")
MultFloat, EltwiseMultFloatWithRet = _scalar("MultFloat", "mul")
MultFloatWithRet = EltwiseMultFloatWithRet
DivFloat, DivFloatWithRet = _scalar("DivFloat", "div")
PowFloat, PowFloatWithRet = _scalar("PowFloat", "pow")
LTFloat, _ = _scalar("LTFloat", "lt")
LEFloat, _ = _scalar("LEFloat", "le")
GTFloat, _ = _scalar("GTFloat", "gt")
GEFloat, _ = _scalar("GEFloat", "ge")
EQFloat, _ = _scalar("EQFloat", "eq")
_rdiv_new, _rdiv_ret = _scalar("DivScalarTensor", "rdiv")


def DivFloatTensor(s, x, ret=None):
    """s / x element-wise."""
    if ret is None:
        return _rdiv_new(x, s)
    return _rdiv_ret(x, s, ret)


def Axpy(alpha, x, y):
    """y += alpha * x.  ``alpha`` may be a one-element tensor living on the
    device, which keeps schedule-dependent scalars (learning rate, momentum)
    out of the recorded closures."""
    if isinstance(alpha, Tensor):
        _submit(y._dev, "Axpy",
                lambda al, a, yin, o: o._dev.math.axpy(al.flat()[:1], a.view(),
                                                       o.view()),
                [alpha, x, y], [y])
    else:
        alpha = float(alpha)
        _submit(y._dev, "Axpy",
                lambda a, yin, o: o._dev.math.axpy(alpha, a.view(), o.view()),
                [x, y], [y])
    return y


# ---------------------------------------------------------------------------
# random
# ---------------------------------------------------------------------------
def Bernoulli(p, t):
    p = float(p)
    _submit(t._dev, "Bernoulli", lambda o: o._dev.math.bernoulli(o.view(), p),
            [], [t])
    return t


def Gaussian(mean, std, t):
    mean, std = float(mean), float(std)
    _submit(t._dev, "Gaussian",
            lambda o: o._dev.math.gaussian(o.view(), mean, std), [], [t])
    return t


def Uniform(low, high, t):
    low, high = float(low), float(high)
    _submit(t._dev, "Uniform",
            lambda o: o._dev.math.uniform(o.view(), low, high), [], [t])
    return t


# ---------------------------------------------------------------------------
# reductions
# ---------------------------------------------------------------------------
def SumAll(x):
    out = Tensor([1], x._dev, x._dtype)
    _submit(x._dev, "SumAll",
            lambda a, o: a._dev.math.sum_all(a.view(), o.view()), [x], [out])
    return out


def _no_host_result_while_recording(dev, what):
    """Reductions that hand a Python float back read the operand at call
    time.  While a step is being recorded the operand's producers have not
    run (and never run in program order under replay), so the value would be
    garbage -- the reference documents the same hazard for ``Sum<float>`` and
    the norms (src/core/tensor/tensor.cc:815-843, 1211-1227) and returns it
    silently.  Here the call fails."""
    if dev.graph_enabled():
        raise RuntimeError(
            "%s returns a host value and cannot be part of a recorded graph: "
            "call it outside train_one_batch / with graph mode off, or use "
            "the tensor-valued form (SumAll, autograd.reduce_sum)" % what)


def SumAsFloat(x):
    _no_host_result_while_recording(x._dev, "SumAsFloat")
    return x._dev.math.sum_float(x.view())


def SumRows(m, out):
    """out[j] = sum_i m[i, j]"""
    assert m.nDim() == 2
    _submit(m._dev, "SumRows",
            lambda a, o: a._dev.math.sum_axis(a.view(), 0, o.view()), [m], [out])
    return out


def SumColumns(m, out):
    """out[i] = sum_j m[i, j]"""
    assert m.nDim() == 2
    _submit(m._dev, "SumColumns",
            lambda a, o: a._dev.math.sum_axis(a.view(), 1, o.view()), [m], [out])
    return out


def Sum(m, axis):
    if axis == 0:
        out = Tensor([m._shape[1]], m._dev, m._dtype)
        return SumRows(m, out)
    assert axis == 1, "Sum supports axis 0 or 1 of a matrix"
    out = Tensor([m._shape[0]], m._dev, m._dtype)
    return SumColumns(m, out)


def Average(m, axis):
    s = Sum(m, axis)
    return MultFloat(s, 1.0 / m._shape[axis])


def RowMax(m):
    assert m.nDim() == 2
    out = Tensor([m._shape[0]], m._dev, m._dtype)
    _submit(m._dev, "RowMax",
            lambda a, o: a._dev.math.row_max(a.view(), o.view()), [m], [out])
    return out


def _as_matrix(t, axis):
    """(rows, cols) when the tensor is flattened at ``axis``."""
    nd = t.nDim()
    if axis < 0:
        axis += nd
    if nd == 1:
        return 1, t._shape[0]
    rows = Product(t._shape, 0, axis) if axis > 0 else 1
    cols = Product(t._shape, axis, nd)
    return rows, cols


def SoftMax(x, axis_or_out=None, axis=None):
    """SoftMax(in) / SoftMax(in, axis) / SoftMax(in, out): soft-max over the
    trailing dimensions starting at ``axis`` (default: the rows of a matrix)."""
    out = None
    if isinstance(axis_or_out, Tensor):
        out = axis_or_out
        ax = 0 if axis is None else axis
        if axis is None:
            ax = 1 if x.nDim() > 1 else 0
    else:
        ax = axis_or_out if axis_or_out is not None else (
            1 if x.nDim() > 1 else 0)
    rows, cols = _as_matrix(x, ax)
    src = x if x.is_contiguous() else Contiguous(x)
    ret = out if out is not None else Tensor(x._shape, x._dev, x._dtype)

    def body(a, o):
        a._dev.math.softmax(a.flat()[:rows * cols].view(rows, cols),
                            o.flat()[:rows * cols].view(rows, cols))
    _submit(x._dev, "SoftMax", body, [src], [ret])
    return ret


def SoftMaxBackward(dy, axis, y):
    rows, cols = _as_matrix(y, axis)
    d = dy if dy.is_contiguous() else Contiguous(dy)
    yy = y if y.is_contiguous() else Contiguous(y)
    ret = Tensor(y._shape, y._dev, y._dtype)

    def body(a, b, o):
        n = rows * cols
        a._dev.math.softmax_bwd(a.flat()[:n].view(rows, cols),
                                b.flat()[:n].view(rows, cols),
                                o.flat()[:n].view(rows, cols))
    _submit(y._dev, "SoftMaxBackward", body, [d, yy], [ret])
    return ret


def CrossEntropyFwd(p, t):
    batch = p._shape[0] if p.nDim() == 2 else 1
    dim = p.Size() // batch
    loss = Tensor([batch], p._dev, p._dtype)

    def body(pp, tt, ll):
        pp._dev.math.cross_entropy_fwd(pp.view().reshape(batch, dim), tt.view(),
                                       ll.view())
    _submit(p._dev, "ComputeCrossEntropy", body, [p, t], [loss])
    return loss


def SoftmaxCrossEntropyBwd(p, t):
    batch = p._shape[0] if p.nDim() == 2 else 1
    dim = p.Size() // batch
    g = p.Clone()

    def body(gg_in, tt, gg):
        gg._dev.math.softmax_ce_bwd(gg.view().reshape(batch, dim), tt.view())
    _submit(p._dev, "SoftmaxCrossEntropyBackward", body, [g, t], [g])
    return g


def SoftmaxCrossEntropyFused(x, t):
    """One-kernel soft-max + mean cross-entropy.  Returns (loss[1], p)."""
    batch = x._shape[0]
    dim = x.Size() // batch
    src = x if x.is_contiguous() else Contiguous(x)
    p = Tensor(x._shape, x._dev, x._dtype)
    loss = Tensor([1], x._dev, x._dtype)

    def body(a, tt, pp, ll):
        a._dev.math.softmax_ce_fused_fwd(a.view().reshape(batch, dim), tt.view(),
                                         pp.view().reshape(batch, dim), ll.view())
    _submit(x._dev, "SoftmaxCrossEntropyForward", body, [src, t], [p, loss])
    return loss, p


def SoftmaxCrossEntropyFusedBwd(p, t):
    """dx = (p - onehot(t)) / batch in one kernel."""
    batch = p._shape[0]
    dim = p.Size() // batch
    dx = Tensor(p._shape, p._dev, p._dtype)

    def body(pp, tt, dd):
        pp._dev.math.softmax_ce_fused_bwd(pp.view().reshape(batch, dim),
                                          tt.view(), dd.view().reshape(batch, dim))
    _submit(p._dev, "SoftmaxCrossEntropyBackward", body, [p, t], [dx])
    return dx


# ---------------------------------------------------------------------------
# linear algebra
# ---------------------------------------------------------------------------
def MultWithScale(alpha, A, B, beta, C):
    alpha, beta = float(alpha), float(beta)
    reads = [A, B] + ([C] if beta else [])
    # the kernels trust these extents: check them here, on the host
    inner = B._shape[0] if B.nDim() == 1 else B._shape[-2]
    if A._shape[-1] != inner:
        raise ValueError("matmul: inner extents differ, %s x %s" %
                         (A._shape, B._shape))
    want = list(A._shape[:-1]) + ([] if B.nDim() == 1 else [B._shape[-1]])
    if Product(C._shape) != Product(want) or (
            B.nDim() > 1 and C._shape[-1] != B._shape[-1]):
        raise ValueError("matmul: result %s does not fit %s x %s" %
                         (C._shape, A._shape, B._shape))
    if B.nDim() > 2 and list(A._shape[:-2]) != list(B._shape[:-2]):
        raise ValueError("matmul: batch extents differ, %s x %s" %
                         (A._shape, B._shape))
    if B.nDim() == 1:
        assert A.nDim() == 2

        def body(*ts):
            a, b, c = ts[0], ts[1], ts[-1]
            c._dev.math.gemv(alpha, a.view(), b.view(), beta, c.view())
        _submit(C._dev, "GEMV", body, reads, [C])
    elif B.nDim() == 2:
        assert A.nDim() == 2 and not C.transpose()

        def body(*ts):
            a, b, c = ts[0], ts[1], ts[-1]
            c._dev.math.gemm(alpha, a.view(), b.view(), beta, c.view())
        _submit(C._dev, "GEMM", body, reads, [C])
    elif B.nDim() in (3, 4):
        assert A.nDim() == B.nDim() and not C.transpose()

        def body(*ts):
            a, b, c = ts[0], ts[1], ts[-1]
            av, bv, cv = a.view(), b.view(), c.view()
            if av.dim() == 4:
                av = av.reshape(-1, av.shape[-2], av.shape[-1])
                bv = bv.reshape(-1, bv.shape[-2], bv.shape[-1])
                cv = cv.view(-1, cv.shape[-2], cv.shape[-1])
            c._dev.math.gemm(alpha, av, bv, beta, cv)
        _submit(C._dev, "GEMMBatched", body, reads, [C])
    else:
        raise ValueError("Un-supported tensor dimentions %dd matmul %dd" %
                         (A.nDim(), B.nDim()))
    return C


def LinearBias(x, w, b):
    """x (..., K) @ w (K, N) + b (N,) with the bias added in the epilogue of
    the product: one kernel, no broadcast addition."""
    if x._shape[-1] != w._shape[0] or w.nDim() != 2 or \
            Product(b._shape) != w._shape[1]:
        raise ValueError("linear: %s x %s + %s" % (x._shape, w._shape,
                                                     b._shape))
    out = Tensor(list(x._shape[:-1]) + [w._shape[1]], x._dev, x._dtype)
    src = x if x.is_contiguous() else Contiguous(x)
    K, N = w._shape

    def body(a, wt, bias, c):
        c._dev.math.gemm(1.0, a.view().reshape(-1, K), wt.view(), 0.0,
                         c.view().reshape(-1, N), bias=bias.view().reshape(-1))
    _submit(x._dev, "LinearBias", body, [src, w, b], [out])
    return out


# The three gradients of y = x w + b read dy and nothing of each other.
# SINGA_LINEAR_BWD_STREAMS=1 runs them on three streams, joined before the
# operation ends (parallel branches of a captured step).  Off by default: at
# transformer sizes the two products fill the chip on their own and the
# concurrent launches cost more than the hidden column sum saves (BERT-base
# fine-tuning, batch 32: fp16 13.83 -> 13.97 ms, float32 49.0 -> 50.7 ms).
_LINEAR_BWD_STREAMS = os.environ.get("SINGA_LINEAR_BWD_STREAMS", "0") == "1"


def LinearBackward(dy2, x2, w, need_dx=True, need_dw=True):
    """(dx, dw, db) of y = x2 (M, K) @ w (K, N) + b for dy2 (M, N), as ONE
    recorded operation: dx = dy2 w^T, dw = x2^T dy2, db = column sums of dy2
    (entries not asked for are None)."""
    M, K = x2._shape
    N = w._shape[1]
    assert dy2._shape == [M, N] and w._shape[0] == K
    dev = dy2._dev
    g = dy2 if dy2.is_contiguous() else Contiguous(dy2)
    xs = x2 if x2.is_contiguous() else Contiguous(x2)
    dx = Tensor([M, K], dev, dy2._dtype) if need_dx else None
    dw = Tensor([K, N], dev, w._dtype) if need_dw else None
    db = Tensor([N], dev, dy2._dtype)
    outs = [t for t in (dx, dw, db) if t is not None]

    def body(gg, xx, ww, *os_):
        os_ = list(os_)
        odx = os_.pop(0) if need_dx else None
        odw = os_.pop(0) if need_dw else None
        odb = os_.pop(0)
        m = gg._dev.math
        gv, xv, wv = gg.view(), xx.view(), ww.view()
        dxv = odx.view() if odx is not None else None
        dwv = odw.view() if odw is not None else None
        dbv = odb.view()
        jobs = [lambda: m.sum_axis(gv, 0, dbv)]
        if dwv is not None:
            jobs.append(lambda: m.gemm(1.0, xv.t(), gv, 0.0, dwv))
        if dxv is not None:
            jobs.append(lambda: m.gemm(1.0, gv, wv.t(), 0.0, dxv))
        if not (_LINEAR_BWD_STREAMS and gg._dev.lang() != 0) or len(jobs) < 2:
            for job in jobs:
                job()
            return
        tdev = gg._dev.torch_device
        sides = getattr(gg._dev, "_streams_linear_bwd", None)
        if sides is None:
            sides = gg._dev._streams_linear_bwd = [
                torch.cuda.Stream(device=tdev) for _ in range(2)]
        cur = torch.cuda.current_stream(tdev)
        used = sides[:len(jobs) - 1]
        for side, job in zip(used, jobs[:-1]):
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                job()
        jobs[-1]()                      # the last job stays on this stream
        for side in used:
            cur.wait_stream(side)

    _submit(dev, "LinearBackward", body, [g, xs, w], outs)
    return dx, dw, db


def MultWithRet(A, B, C):
    return MultWithScale(1.0, A, B, 0.0, C)


def Mult(A, B):
    if B.nDim() == 1:
        out = Tensor([A._shape[0]], A._dev, A._dtype)
        MultWithScale(1.0, A, B, 0.0, out)
        return out
    A_ = Broadcast(A, B._shape, 2)
    B_ = Broadcast(B, A._shape, 2)
    s = list(A_._shape)
    s.pop()
    if B.nDim() > 1:
        s.append(B._shape[-1])
    out = Tensor(s, A._dev, A._dtype)
    MultWithScale(1.0, A_, B_, 0.0, out)
    return out


def _row_col(name, fn):
    def op(v, M):
        assert M.nDim() == 2
        _submit(M._dev, name, lambda vv, mm_in, mm: fn(mm._dev.math, vv, mm),
                [v, M], [M])
        return M
    op.__name__ = name
    return op


def _addrow_body(alpha, beta):
    def f(math, v, m):
        mv = m.view()
        if beta != 1.0:
            math.binary_scalar("mul", mv, beta, mv)
        vv = v.view().reshape(1, -1)
        if alpha != 1.0:
            tmp = torch.empty_like(vv)
            math.binary_scalar("mul", vv, alpha, tmp)
            vv = tmp
        math.binary("add", mv, vv.expand_as(mv), mv)
    return f


def _addcol_body(alpha, beta):
    def f(math, v, m):
        mv = m.view()
        if beta != 1.0:
            math.binary_scalar("mul", mv, beta, mv)
        vv = v.view().reshape(-1, 1)
        if alpha != 1.0:
            tmp = torch.empty_like(vv)
            math.binary_scalar("mul", vv, alpha, tmp)
            vv = tmp
        math.binary("add", mv, vv.expand_as(mv), mv)
    return f


def AddRow(v, M):
    """M[i, :] += v"""
    return _row_col("AddRow", _addrow_body(1.0, 1.0))(v, M)


def AddRowWithScale(alpha, beta, v, M):
    """M[i, :] = alpha * v + beta * M[i, :]"""
    return _row_col("AddRow", _addrow_body(float(alpha), float(beta)))(v, M)


def AddColumn(v, M):
    """M[:, j] += v"""
    return _row_col("AddColumn", _addcol_body(1.0, 1.0))(v, M)


def AddColumnWithScale(alpha, beta, v, M):
    return _row_col("AddColumn", _addcol_body(float(alpha), float(beta)))(v, M)


def _rc_binary(op, row):
    def f(math, v, m):
        mv = m.view()
        vv = v.view().reshape(1, -1) if row else v.view().reshape(-1, 1)
        math.binary(op, mv, vv.expand_as(mv), mv)
    return f


def SubRow(v, M):
    return _row_col("SubRow", _rc_binary("sub", True))(v, M)


def SubColumn(v, M):
    return _row_col("SubColumn", _rc_binary("sub", False))(v, M)


def MultRow(v, M):
    return _row_col("MultRow", _rc_binary("mul", True))(v, M)


def MultColumn(v, M):
    return _row_col("MultColumn", _rc_binary("mul", False))(v, M)


def DivRow(v, M):
    return _row_col("DivRow", _rc_binary("div", True))(v, M)


def DivColumn(v, M):
    return _row_col("DivColumn", _rc_binary("div", False))(v, M)


# ---------------------------------------------------------------------------
# copy / slice / concat / repeat
# ---------------------------------------------------------------------------
def CopyDataToFrom(dst, src, num, dst_offset=0, src_offset=0):
    num, dst_offset, src_offset = int(num), int(dst_offset), int(src_offset)
    if dst._dev is src._dev:
        def body(s, d):
            d._dev.math.copy(s.flat()[src_offset:src_offset + num],
                             d.flat()[dst_offset:dst_offset + num])
        _submit(dst._dev, "CopyDataToFrom", body, [src], [dst])
    else:
        dst._storage()[dst_offset:dst_offset + num].copy_(
            src.flat()[src_offset:src_offset + num])


def RepeatDataToFrom(broadcast_flag, repeats, axis, dst, src, num=None):
    r = _repeat(src, list(repeats), axis)
    dst.CopyData(r)


def _repeat(t, repeats, axis):
    if axis is None or axis == 9999:
        total = repeats[0]
        shape = [Product(t._shape) * total]
        src = Reshape(t, [Product(t._shape)])
        ax = 0
    else:
        ax = axis if axis >= 0 else axis + t.nDim()
        shape = list(t._shape)
        shape[ax] = t._shape[ax] * repeats[0] if len(repeats) == 1 \
            else int(sum(repeats))
        src = t
    out = Tensor(shape, t._dev, t._dtype)
    if len(repeats) == 1:
        reps = int(repeats[0])

        def body(s, o):
            # out[..., i * reps + j, ...] = src[..., i, ...]: one strided copy
            v = s.view()
            shp = list(v.shape)
            ov = o.view().view(shp[:ax] + [shp[ax], reps] + shp[ax + 1:])
            o._dev.math.copy(v.unsqueeze(ax + 1).expand(ov.shape), ov)
        _submit(t._dev, "Repeat", body, [src], [out])
        return out
    # per-element repeat counts: gather through a host-built index map
    idx_np = np.repeat(np.arange(src._shape[ax], dtype=np.int32),
                       np.asarray(repeats, dtype=np.int64))
    idx = Tensor([idx_np.size], t._dev, kInt)
    idx.CopyIntDataFromHostPtr(idx_np)
    _submit(t._dev, "Repeat",
            lambda s, ix, o: s._dev.math.index_select(
                s.view(), ax, ix.view(), o.view()),
            [src if src.is_contiguous() else Contiguous(src), idx], [out])
    return out


def SliceOn(t, start, end, axis):
    start, end, axis = int(start), int(end), int(axis)
    shape = list(t._shape)
    assert 0 <= start < end <= shape[axis], "invalid slice"
    shape[axis] = end - start
    out = Tensor(shape, t._dev, t._dtype)
    _submit(t._dev, "SliceOn",
            lambda s, o: s._dev.math.copy(
                s.view().narrow(axis, start, end - start), o.view()),
            [t], [out])
    return out


# element types the multi-tensor pack / unpack kernels move
_PACKABLE = (dtypes.kFloat32, dtypes.kFloat16, dtypes.kBFloat16)


def _blockwise(shape, axis):
    """True when pieces cut along ``axis`` are contiguous blocks of a
    contiguous tensor of ``shape`` (every earlier axis has extent 1)."""
    return Product(shape[:axis]) == 1 if axis > 0 else True


def ConcatOn(tensors, axis):
    """Concatenate along ``axis``: one multi-tensor pack kernel when the
    pieces are contiguous blocks of the result, one strided copy per piece
    otherwise."""
    tensors = list(tensors)
    axis = int(axis)
    shape = list(tensors[0]._shape)
    for t in tensors[1:]:
        other = list(t._shape)
        if len(other) != len(shape) or \
                other[:axis] + other[axis + 1:] != shape[:axis] + shape[axis + 1:]:
            raise ValueError("concatenation along axis %d of shapes %s and %s"
                             % (axis, tensors[0]._shape, t._shape))
    shape[axis] = sum(t._shape[axis] for t in tensors)
    out = Tensor(shape, tensors[0]._dev, tensors[0]._dtype)
    sizes = [t._shape[axis] for t in tensors]
    packed = _blockwise(shape, axis) and len(tensors) > 1 and \
        out._dtype in _PACKABLE and \
        all(t._dtype == out._dtype for t in tensors)
    if packed:
        tensors = [t if t.is_contiguous() else Contiguous(t) for t in tensors]

    def body(*ts):
        o = ts[-1]
        m = o._dev.math
        if packed:
            views = [s.view() for s in ts[:-1]]
            m.pack(m.pack_plan(views), views, o.view().view(-1))
            return
        ov = o.view()
        pos = 0
        for s, n in zip(ts[:-1], sizes):
            m.copy(s.view(), ov.narrow(axis, pos, n))
            pos += n
    _submit(out._dev, "ConcatOn", body, tensors, [out])
    return out


def SplitOn(t, sizes, axis):
    """The pieces of ``t`` along ``axis`` with the given extents, as new
    tensors produced by ONE graph node (one multi-tensor unpack kernel when
    the pieces are contiguous blocks)."""
    axis = int(axis)
    sizes = [int(n) for n in sizes]
    assert sum(sizes) == t._shape[axis], "split sizes do not cover the axis"
    outs = []
    for n in sizes:
        shape = list(t._shape)
        shape[axis] = n
        outs.append(Tensor(shape, t._dev, t._dtype))
    packed = _blockwise(t._shape, axis) and len(sizes) > 1 and \
        t._dtype in _PACKABLE
    src = t if (not packed or t.is_contiguous()) else Contiguous(t)

    def body(s, *os):
        m = s._dev.math
        if packed:
            views = [o.view() for o in os]
            m.unpack(m.pack_plan(views), s.view().reshape(-1), views, 1.0)
            return
        sv = s.view()
        pos = 0
        for o, n in zip(os, sizes):
            m.copy(sv.narrow(axis, pos, n), o.view())
            pos += n
    _submit(t._dev, "SplitOn", body, [src], outs)
    return outs


def ConcatRows(tensors):
    return ConcatOn(tensors, 0)


ConcatenateRows = ConcatRows


def ConcatColumns(tensors):
    return ConcatOn(tensors, 1)


ConcatenateColumns = ConcatColumns


def CopyRows(t, start, end):
    return SliceOn(t, start, end, 0)


SliceRows = CopyRows


def CopyColumns(t, start, end):
    return SliceOn(t, start, end, 1)


SliceColumns = CopyColumns


# ---------------------------------------------------------------------------
# index ops that the reference builds from per-row copies
# ---------------------------------------------------------------------------
def GatherRows(table, indices, out_shape):
    """out[..., :] = table[indices[...], :] in one kernel."""
    out = Tensor(out_shape, table._dev, table._dtype)
    src = table if table.is_contiguous() else Contiguous(table)
    _submit(table._dev, "GatherRows",
            lambda tb, ix, o: tb._dev.math.gather_rows(
                tb.view(), ix.view(), o.view()),
            [src, indices], [out])
    return out


def ScatterAddRows(grad, indices, table_shape):
    """dtable[i, :] = sum over positions p with indices[p] == i of grad[p, :]"""
    out = Tensor(table_shape, grad._dev, grad._dtype)
    g = grad if grad.is_contiguous() else Contiguous(grad)
    _submit(grad._dev, "ScatterAddRows",
            lambda gg, ix, o: gg._dev.math.scatter_add_rows(
                gg.view(), ix.view(), o.view()),
            [g, indices], [out])
    return out


# ---------------------------------------------------------------------------
# strided region copies, tiling and n-d reductions (single-kernel forms of ops
# the reference composes from many per-slice copies)
# ---------------------------------------------------------------------------
def _slices(starts, ends, steps):
    return tuple(slice(int(a), int(b), int(c))
                 for a, b, c in zip(starts, ends, steps))


def SliceND(t, starts, ends, steps=None):
    """out = t[starts:ends:steps] (one entry per dimension, positive steps)."""
    nd = t.nDim()
    steps = [1] * nd if steps is None else list(steps)
    sl = _slices(starts, ends, steps)
    shape = [len(range(int(a), int(b), int(c)))
             for a, b, c in zip(starts, ends, steps)]
    out = Tensor(shape, t._dev, t._dtype)
    _submit(t._dev, "SliceND",
            lambda s, o: s._dev.math.copy(s.view()[sl], o.view()), [t], [out])
    return out


def EmbedND(src, shape, starts, ends, steps=None, fill=0.0):
    """out = full(shape, fill); out[starts:ends:steps] = src."""
    nd = len(shape)
    steps = [1] * nd if steps is None else list(steps)
    sl = _slices(starts, ends, steps)
    out = Tensor(shape, src._dev, src._dtype)
    fill = float(fill)

    def body(s, o):
        m = o._dev.math
        m.fill(o.flat(), fill)
        m.copy(s.view(), o.view()[sl])
    _submit(src._dev, "EmbedND", body, [src], [out])
    return out


def TileND(t, repeats):
    """numpy.tile with len(repeats) == t.nDim()."""
    repeats = [int(r) for r in repeats]
    shape = [s * r for s, r in zip(t._shape, repeats)]
    out = Tensor(shape, t._dev, t._dtype)
    inter, exp_shape = [], []
    for s, r in zip(t._shape, repeats):
        inter += [r, s]

    def body(s, o):
        v = s.view()
        idx = []
        for _ in v.shape:
            idx += [None, slice(None)]
        src = v[tuple(idx)].expand(inter)
        o._dev.math.copy(src, o.view().view(inter))
    _submit(t._dev, "Tile", body, [t], [out])
    return out


def UntileND(dy, x_shape, repeats):
    """Gradient of TileND: sums the tiles back onto the original extent."""
    repeats = [int(r) for r in repeats]
    inter = []
    for s, r in zip(x_shape, repeats):
        inter += [r, s]
    axes = list(range(0, 2 * len(x_shape), 2))
    d = dy if dy.is_contiguous() else Contiguous(dy)
    return ReduceND(Reshape(d, inter), axes, "sum", keepdims=False)


def ReduceND(t, axes, op="sum", keepdims=False):
    """Reduce over ``axes`` with sum / mean / max / min."""
    nd = t.nDim()
    axes = sorted(set(a + nd if a < 0 else a for a in axes))
    kept = [i for i in range(nd) if i not in axes]
    out_shape = [t._shape[i] for i in kept]
    full_shape = [1 if i in axes else t._shape[i] for i in range(nd)]
    out = Tensor(full_shape if keepdims else (out_shape or [1]),
                 t._dev, t._dtype)
    _submit(t._dev, "Reduce_" + op,
            lambda s, o: s._dev.math.reduce(op, s.view(), axes, o.flat()),
            [t], [out])
    return out


def BroadcastTo(t, shape):
    """Materialised numpy-style broadcast of ``t`` to ``shape``."""
    b = Broadcast(t, shape)
    assert b._shape == list(shape), "cannot broadcast %s to %s" % (
        t._shape, list(shape))
    return Contiguous(b)


def IndexSelect(t, axis, index):
    """out = take(t, index, axis); ``index`` is a 1-d int tensor."""
    shape = list(t._shape)
    shape[axis] = index.Size()
    out = Tensor(shape, t._dev, t._dtype)
    src = t if t.is_contiguous() else Contiguous(t)
    _submit(t._dev, "IndexSelect",
            lambda s, ix, o: s._dev.math.index_select(
                s.view(), axis, ix.view().reshape(-1), o.view()),
            [src, index], [out])
    return out


def IndexAdd(dy, axis, index, x_shape):
    """Gradient of IndexSelect: out = zeros(x_shape); out.index_add(dy)."""
    out = Tensor(x_shape, dy._dev, dy._dtype)
    d = dy if dy.is_contiguous() else Contiguous(dy)
    _submit(dy._dev, "IndexAdd",
            lambda g, ix, o: g._dev.math.index_add(
                g.view(), axis, ix.view().reshape(-1), o.view()),
            [d, index], [out])
    return out


def SGDUpdate(p, g, buf, lr, momentum, dampening, weight_decay, nesterov,
              scale=1.0):
    """Fused SGD step on one parameter: optional gradient scaling, weight
    decay, momentum (with dampening / Nesterov) and the update itself in a
    single kernel.  Hyper-parameters are one-element device tensors so the
    recorded step follows schedule changes."""
    nesterov = bool(nesterov)
    scale = float(scale)
    if buf is not None:
        def body(pp_in, gg, bb_in, l, m, d, w, pp, bb):
            pp._dev.math.sgd_update(pp.view(), gg.view(), bb.view(), l.flat(),
                                    m.flat(), d.flat(), w.flat(), nesterov,
                                    scale)
        _submit(p._dev, "SGDUpdate", body,
                [p, g, buf, lr, momentum, dampening, weight_decay], [p, buf])
    else:
        def body(pp_in, gg, l, m, d, w, pp):
            pp._dev.math.sgd_update(pp.view(), gg.view(), None, l.flat(),
                                    m.flat(), d.flat(), w.flat(), nesterov,
                                    scale)
        _submit(p._dev, "SGDUpdate", body,
                [p, g, lr, momentum, dampening, weight_decay], [p])
    return p


def _one_device(what, dev, tensors):
    """Fused multi-tensor kernels take raw addresses: a tensor of another
    device (optimizer state read from a checkpoint and left on the host) must
    stop here, not inside the kernel."""
    for i, t in enumerate(tensors):
        if t._dev is not dev:
            raise ValueError(
                "%s: operand %d lives on another device than the parameters"
                % (what, i))


def SGDUpdateMulti(params, grads, bufs, lr, momentum, dampening, weight_decay,
                   nesterov, scale=1.0, masters=None):
    """The fused SGD step for a whole list of parameters as ONE graph node
    (and one kernel launch per 48 tensors).  ``bufs`` is a list of momentum
    buffers or None; ``masters`` a list of float32 copies of 16-bit parameters
    (the step is then taken on the copies, with float32 ``bufs``) or None.
    All tensors must live on one device and share a dtype; callers group by
    dtype."""
    nesterov = bool(nesterov)
    scale = float(scale)
    n = len(params)
    assert n == len(grads) and n > 0
    has_buf = bufs is not None
    bufs = list(bufs) if has_buf else []
    has_master = masters is not None
    masters = list(masters) if has_master else []

    def body(*args):
        gs = args[n:2 * n]
        off = 2 * n + len(bufs) + len(masters)
        l, m, d, w = args[off:off + 4]
        outs = args[off + 4:]
        ps = outs[0:n]
        bs = outs[n:2 * n] if has_buf else None
        ms = outs[n + len(bufs):n + len(bufs) + n] if has_master else None
        ps[0]._dev.math.sgd_update_multi(
            [t.view() for t in ps], [t.view() for t in gs],
            [t.view() for t in bs] if has_buf else None, l.flat(), m.flat(),
            d.flat(), w.flat(), nesterov, scale,
            [t.view() for t in ms] if has_master else None)

    reads = list(params) + list(grads) + bufs + masters + [
        lr, momentum, dampening, weight_decay]
    writes = list(params) + bufs + masters
    _one_device("SGDUpdateMulti", params[0]._dev, reads)
    _submit(params[0]._dev, "SGDUpdateMulti", body, reads, writes)
    return params


ADAPT_RMSPROP, ADAPT_ADAGRAD, ADAPT_ADAM = 0, 1, 2


def AdaptiveUpdateMulti(rule, params, grads, state1, state2, hypers,
                        scale=1.0):
    """One fused step of RMSProp / AdaGrad / Adam for a list of parameters (one
    graph node, one launch per 48 tensors).  ``hypers`` = (lr, weight_decay,
    epsilon, a, b, step): one-element device tensors, ``a`` / ``b`` / ``step``
    may be None for rules that do not use them.  ``state2`` is None except
    for Adam."""
    n = len(params)
    assert n > 0 and n == len(grads) == len(state1)
    state2 = list(state2) if state2 is not None else []
    hyp = [h for h in hypers if h is not None]
    slots = [h is not None for h in hypers]
    scale = float(scale)

    def body(*args):
        gs = args[n:2 * n]
        off = 2 * n + n + len(state2)
        hs = iter(args[off:off + len(hyp)])
        full = [next(hs).flat() if present else None for present in slots]
        outs = args[off + len(hyp):]
        ps, s1 = outs[0:n], outs[n:2 * n]
        s2 = outs[2 * n:3 * n] if state2 else None
        ps[0]._dev.math.adaptive_update_multi(
            rule, [t.view() for t in ps], [t.view() for t in gs],
            [t.view() for t in s1], [t.view() for t in s2] if s2 else None,
            *full, scale)

    reads = list(params) + list(grads) + list(state1) + state2 + hyp
    writes = list(params) + list(state1) + state2
    _one_device("AdaptiveUpdateMulti", params[0]._dev, reads)
    _submit(params[0]._dev, "AdaptiveUpdateMulti", body, reads, writes)
    return params


_bin_max, MaxWithRet = _binary("Max", "max")
_bin_min, MinWithRet = _binary("Min", "min")
EltwiseMax = _bin_max
EltwiseMin = _bin_min


class VecTensor(list):
    """Stand-in for the std::vector<Tensor> binding."""

    def push_back(self, t):
        self.append(t)

    def size(self):
        return len(self)


def Shape(*a):
    return list(*a)
