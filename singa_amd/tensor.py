"""User-facing Tensor: a core tensor plus autograd bookkeeping
(``requires_grad``, ``stores_grad``, ``creator``, ``name``), numpy interop and
the module-level math functions.

API parity: reference python/singa/tensor.py (class Tensor :73-779, module
functions :789-1841).  Behavioural notes kept from the reference: ``from_numpy``
narrows float64 -> float32 and int64 -> int32; ``sum`` with an axis returns a
tensor with that axis removed; ``PlaceHolder`` is an alias of ``Tensor``.
"""
import re
from functools import reduce

import itertools

import numpy as np

from . import singa_wrap as singa
from .device import get_default_device

int32 = singa.kInt
float16 = singa.kFloat16
float32 = singa.kFloat32
bfloat16 = singa.kBFloat16
CTensor = singa.Tensor
_NP2DTYPE = {np.float32: float32, np.float64: float32, np.float16: float16,
             np.int32: int32, np.int64: int32}


class Tensor(object):
    tensor_count = 0
    _serial = itertools.count(1)

    @property
    def uid(self):
        """Process-wide serial number of this tensor object.  The autograd
        graph identifies tensors by it: unlike ``id()`` it is never handed to
        a second tensor after the first one has been garbage-collected, which
        would merge the consumer counts of two unrelated tensors."""
        u = self.__dict__.get("_uid")
        if u is None:
            u = self.__dict__["_uid"] = next(Tensor._serial)
        return u

    def __init__(self, shape=(), device=None, dtype=None, data=None,
                 requires_grad=True, stores_grad=False, creator=None,
                 name=None):
        if device is None:
            device = get_default_device()
        if dtype is None:
            # a numpy array brings its own element type (float64 / int64 are
            # narrowed to the 32-bit types the kernels support)
            dtype = float32
            if isinstance(data, np.ndarray):
                dtype = _NP2DTYPE.get(data.dtype.type, float32)
        if isinstance(data, np.ndarray):
            self.data = CTensor(list(data.shape), device, dtype)
            copy_from_numpy(self.data, data)
        elif isinstance(data, CTensor):
            self.data = data
            assert data.device().id() == device.id(), 'not the same device'
        else:
            self.data = CTensor(list(shape), device, dtype)

        self.shape = tuple(self.data.shape())
        self.device = device
        self.dtype = self.data.data_type()
        self.requires_grad = requires_grad
        self.stores_grad = stores_grad
        if name is None:
            self.name = 'Dummy#{}'.format(Tensor.tensor_count)
            Tensor.tensor_count += 1
        else:
            self.name = name
        if creator is None:
            from . import autograd
            self.creator = autograd.Dummy(self, name)
        else:
            self.creator = creator

    def __getitem__(self, keys):
        if type(keys) != tuple:
            keys = (keys,)
        ret = self.clone()
        axis_index = 0
        for key in keys:
            if type(key) == int:
                key += self.shape[axis_index] if key < 0 else 0
                if not (0 <= key < self.shape[axis_index]):
                    raise ValueError("Invalid Index")
                ret.data = singa.SliceOn(ret.data, key, key + 1, axis_index)
            elif type(key) == slice:
                start = key.start if key.start else 0
                end = key.stop if key.stop else self.shape[axis_index]
                start += self.shape[axis_index] if start < 0 else 0
                end += self.shape[axis_index] if end < 0 else 0
                if not (0 <= start < end <= self.shape[axis_index]):
                    raise ValueError("Invalid Index")
                ret.data = singa.SliceOn(ret.data, start, end, axis_index)
            else:
                raise ValueError("Invalid Index")
            axis_index += 1
        ret.shape = tuple(ret.data.shape())
        return ret

    def __setitem__(self, keys, value):
        """``t[:] = other`` (whole-tensor assignment) as used by ``sum(out=)``."""
        if keys == slice(None, None, None):
            self.copy_from(value)
        else:
            raise NotImplementedError("only t[:] = value is supported")

    def is_dummy(self):
        return bool(re.match(r'Dummy#\d+', self.name))

    def ndim(self):
        return self.data.nDim()

    def is_empty(self):
        return self.ndim() == 0

    def is_transpose(self):
        return self.data.transpose()

    def transpose(self, axes=None):
        if axes is None:
            return _wrap(singa.DefaultTranspose(self.data))
        if len(axes) != len(self.shape):
            raise ValueError('dimensions do not match')
        return _wrap(singa.Transpose(self.data, list(axes)))

    def size(self):
        return self.data.Size()

    def memsize(self):
        return self.data.MemSize()

    def contiguous(self):
        return _wrap(singa.Contiguous(self.data))

    def reshape(self, shape):
        assert product(self.shape) == product(shape), \
            'product of shape should be equal'
        return _wrap(singa.Reshape(self.data, list(shape)))

    def reset_like(self, t):
        self.data.ResetLike(t.data)
        self.shape = t.shape
        self.device = t.device
        self.dtype = t.dtype

    @staticmethod
    def _dtype_code(dtype):
        if dtype in (singa.kInt, singa.kFloat16, singa.kFloat32,
                     singa.kBFloat16, singa.kDouble):
            return dtype
        if dtype == 'int':
            return singa.kInt
        if dtype == 'float':
            return singa.kFloat32
        raise TypeError("invalid data type %s" % dtype)

    def as_type(self, dtype):
        dtype = self._dtype_code(dtype)
        return _wrap(singa.Tensor(self.data.AsType(dtype)))

    def to_type(self, dtype):
        assert self.data.initialized()
        dtype = self._dtype_code(dtype)
        self.data.ToType(dtype)
        self.dtype = dtype
        return self

    def to_device(self, device):
        self.data.ToDevice(device)
        self.device = device

    def to_host(self):
        self.data.ToHost()
        self.device = get_default_device()

    def l2(self):
        return self.data.L2()

    def l1(self):
        return self.data.L1()

    def set_value(self, x, inplace=True):
        if not inplace:
            raise NotImplementedError
        self.data.SetFloatValue(float(x))
        return self

    def copy_from_numpy(self, np_array, offset=0):
        assert np_array.size == self.size(), 'tensor shape should be the same'
        copy_from_numpy(self.data, np_array)

    def copy_data(self, t):
        assert isinstance(t, Tensor), 't must be a singa Tensor instance'
        assert t.size() == self.size(), "tensor shape should be the same"
        self.data.CopyData(t.data)

    def copy_from(self, t, offset=0):
        if isinstance(t, Tensor):
            self.copy_data(t)
        elif isinstance(t, np.ndarray):
            self.copy_from_numpy(t)
        else:
            raise ValueError("t should be Tensor or numpy array.")

    def clone(self):
        return _wrap(self.data.Clone())

    def repeat(self, repeats, axis):
        t_ndim = self.ndim()
        if isinstance(repeats, int):
            if repeats < 0:
                raise ValueError(
                    "'repeats' should not be negative: {}".format(repeats))
            if axis is not None and axis < 0:
                axis += t_ndim
            return _wrap(self.data.Repeat([repeats],
                                          9999 if axis is None else axis))
        elif isinstance(repeats, (tuple, list)):
            for rep in repeats:
                if rep < 0:
                    raise ValueError(
                        "'repeats' should be int or sequence: {}".format(
                            repeats))
            if axis is None:
                raise ValueError(
                    "when axis us None, 'repeats' should be int: {}".format(
                        repeats))
            if axis < 0:
                axis += t_ndim
            return _wrap(self.data.Repeat(list(repeats), axis))
        raise ValueError('repeats should be int or sequence')

    def T(self):
        return _wrap(singa.DefaultTranspose(self.data))

    def copy(self):
        """Shallow copy sharing the underlying memory."""
        return _wrap(CTensor(self.data))

    def deepcopy(self):
        return self.clone()

    def bernoulli(self, p, inplace=True):
        if not inplace:
            raise NotImplementedError
        singa.Bernoulli(float(p), self.data)
        return self

    def gaussian(self, mean, std, inplace=True):
        if not inplace:
            raise NotImplementedError
        singa.Gaussian(float(mean), float(std), self.data)
        return self

    def uniform(self, low, high, inplace=True):
        if not inplace:
            raise NotImplementedError
        singa.Uniform(float(low), float(high), self.data)
        return self

    # deprecated row / column helpers (kept for old scripts)
    def add_column(self, v):
        singa.AddColumn(v.data, self.data)

    def add_row(self, v):
        singa.AddRow(v.data, self.data)

    def div_column(self, v):
        singa.DivColumn(v.data, self.data)

    def div_row(self, v):
        singa.DivRow(v.data, self.data)

    def mult_column(self, v):
        singa.MultColumn(v.data, self.data)

    def mult_row(self, v):
        singa.MultRow(v.data, self.data)

    # in-place arithmetic
    def __iadd__(self, x):
        self.data += x.data if isinstance(x, Tensor) else float(x)
        return self

    def __isub__(self, x):
        self.data -= x.data if isinstance(x, Tensor) else float(x)
        return self

    def __imul__(self, x):
        self.data *= x.data if isinstance(x, Tensor) else float(x)
        return self

    def __itruediv__(self, x):
        self.data /= x.data if isinstance(x, Tensor) else float(x)
        return self

    # binary arithmetic
    def __add__(self, rhs):
        if isinstance(rhs, Tensor):
            return from_raw_tensor(singa.__add__(self.data, rhs.data))
        return _call_singa_func(singa.AddFloat, self.data, float(rhs))

    def __sub__(self, rhs):
        if isinstance(rhs, Tensor):
            return from_raw_tensor(singa.__sub__(self.data, rhs.data))
        return _call_singa_func(singa.SubFloat, self.data, float(rhs))

    def __mul__(self, rhs):
        if isinstance(rhs, Tensor):
            return from_raw_tensor(singa.__mul__(self.data, rhs.data))
        return _call_singa_func(singa.MultFloat, self.data, float(rhs))

    def __div__(self, rhs):
        if isinstance(rhs, Tensor):
            return from_raw_tensor(singa.__div__(self.data, rhs.data))
        return _call_singa_func(singa.DivFloat, self.data, float(rhs))

    def __truediv__(self, rhs):
        return self.__div__(rhs)

    def __floordiv__(self, rhs):
        tmp = self.__div__(rhs)
        return _call_singa_func(singa.Floor, tmp.data)

    def __lt__(self, rhs):
        if isinstance(rhs, Tensor):
            return from_raw_tensor(singa.__lt__(self.data, rhs.data))
        return _call_singa_func(singa.LTFloat, self.data, float(rhs))

    def __le__(self, rhs):
        if isinstance(rhs, Tensor):
            return from_raw_tensor(singa.__le__(self.data, rhs.data))
        return _call_singa_func(singa.LEFloat, self.data, float(rhs))

    def __gt__(self, rhs):
        if isinstance(rhs, Tensor):
            return from_raw_tensor(singa.__gt__(self.data, rhs.data))
        return _call_singa_func(singa.GTFloat, self.data, float(rhs))

    def __ge__(self, rhs):
        if isinstance(rhs, Tensor):
            return from_raw_tensor(singa.__ge__(self.data, rhs.data))
        return _call_singa_func(singa.GEFloat, self.data, float(rhs))

    def __eq__(self, rhs):
        if isinstance(rhs, Tensor):
            return from_raw_tensor(singa.__eq__(self.data, rhs.data))
        elif rhs is None:
            return False
        return _call_singa_func(singa.EQFloat, self.data, float(rhs))

    __hash__ = object.__hash__

    def __bool__(self):
        # ``if self.b:`` in layer code asks "is there a tensor", not its value
        return True

    def _rop(self, lhs, fn):
        one = Tensor(self.shape, self.device, self.dtype)
        one.set_value(float(lhs))
        return fn(one)

    def __radd__(self, lhs):
        return self._rop(lhs, lambda one: one.__iadd__(self))

    def __rsub__(self, lhs):
        return self._rop(lhs, lambda one: one.__isub__(self))

    def __rmul__(self, lhs):
        return self._rop(lhs, lambda one: one.__imul__(self))

    def __rdiv__(self, lhs):
        return self._rop(lhs, lambda one: one.__itruediv__(self))

    __rtruediv__ = __rdiv__

    def __neg__(self):
        return self * -1.0

    dtype_name = {float16: "float16", float32: "float32", int32: "int32",
                  bfloat16: "bfloat16"}

    def __repr__(self):
        return "%s, %s" % (np.array2string(to_numpy(self)),
                           self.dtype_name.get(self.dtype, str(self.dtype)))


PlaceHolder = Tensor


def _wrap(ct):
    t = Tensor.__new__(Tensor)
    t.data = ct
    t.shape = tuple(ct.shape())
    t.device = ct.device()
    t.dtype = ct.data_type()
    t.requires_grad = True
    t.stores_grad = False
    t.name = 'Dummy#{}'.format(Tensor.tensor_count)
    Tensor.tensor_count += 1
    from . import autograd
    t.creator = autograd.Dummy(t, None)
    return t


def from_raw_tensor(t):
    return _wrap(t)


def from_raw_tensors(tt):
    return [from_raw_tensor(t) for t in list(tt)]


def _call_singa_func(_singa_func, *args):
    return _wrap(_singa_func(*args))


def zeros_like(t):
    ret = Tensor(t.shape, t.device, t.dtype)
    ret.set_value(0.0)
    return ret


def ones_like(t):
    ret = Tensor(t.shape, t.device, t.dtype)
    ret.set_value(1.0)
    return ret


def product(shape):
    return reduce(lambda x, y: x * y, shape)


def sizeof(dtype):
    return singa.SizeOf(dtype)


def contiguous(tensor):
    return _call_singa_func(singa.Contiguous, tensor.data)


def reshape(tensor, shape):
    return _call_singa_func(singa.Reshape, tensor.data, list(shape))


def transpose(t, axes=None):
    return t.transpose(axes)


def copy_data_to_from(dst, src, size, dst_offset=0, src_offset=0):
    singa.CopyDataToFrom(dst.data, src.data, size, dst_offset, src_offset)


def from_numpy(np_array, dev=None):
    assert type(np_array) is np.ndarray, 'Must input numpy array'
    if np_array.dtype == np.float64:
        np_array = np_array.astype(np.float32)
    if np_array.dtype == np.int64:
        np_array = np_array.astype(np.int32)
    if np_array.dtype == np.float32:
        dtype = float32
    elif np_array.dtype == np.float16:
        dtype = float16
    else:
        assert np_array.dtype == np.int32, \
            'Only float and int tensors are supported'
        dtype = int32
    ret = Tensor(np_array.shape, dtype=dtype)
    ret.copy_from_numpy(np_array)
    if dev:
        ret.to_device(dev)
    return ret


def to_host(t):
    ret = t.clone()
    ret.to_host()
    return ret


def to_numpy(t):
    ct = t.data
    if ct.data_type() == float32:
        arr = ct.GetFloatValue()
    elif ct.data_type() == float16:
        arr = ct.GetHalfFloatValue()
    elif ct.data_type() == int32:
        arr = ct.GetIntValue()
    elif ct.data_type() == bfloat16:
        arr = ct.GetFloatValue()
    else:
        arr = ct.get_value()
    return arr.reshape(t.shape)


def abs(t):
    return _call_singa_func(singa.Abs, t.data)


def exp(t):
    return _call_singa_func(singa.Exp, t.data)


def ceil(t):
    return _call_singa_func(singa.Ceil, t.data)


def log(t):
    return _call_singa_func(singa.Log, t.data)


def sigmoid(t):
    return _call_singa_func(singa.Sigmoid, t.data)


def sign(t):
    return _call_singa_func(singa.Sign, t.data)


def sqrt(t):
    return _call_singa_func(singa.Sqrt, t.data)


def square(t):
    return _call_singa_func(singa.Square, t.data)


def tanh(t):
    return _call_singa_func(singa.Tanh, t.data)


def relu(t):
    return _call_singa_func(singa.ReLU, t.data)


def sum(t, axis=None, out=None):
    """Sum over ``axis`` (None, int or tuple); the reduced axes are removed."""
    t_ndim = t.ndim()
    if axis is None:
        one = Tensor(t.shape, t.device)
        one.set_value(1.0)
        ret = tensordot(t, one, t_ndim)
    elif isinstance(axis, int):
        if axis < 0:
            axis += t_ndim
        one = Tensor(shape=(int(t.shape[axis]),), device=t.device)
        one.set_value(1.0)
        ret = tensordot(t, one, axes=([axis], [0]))
    elif isinstance(axis, (tuple, list)):
        l_axis = [a + t_ndim if a < 0 else a for a in axis]
        one = Tensor(tuple(t.shape[x] for x in l_axis), t.device)
        one.set_value(1.0)
        ret = tensordot(t, one, (l_axis, list(range(one.ndim()))))
    else:
        raise ValueError("axis must be None, int or tuple")
    if out is not None:
        if out.shape != ret.shape:
            raise ValueError('dimensions do not match')
        out[:] = ret
        return out
    return ret


def pow(t, x, out=None):
    if out is None:
        if isinstance(x, Tensor):
            return _call_singa_func(singa.Pow, t.data, x.data)
        return _call_singa_func(singa.PowFloat, t.data, x)
    if isinstance(x, Tensor):
        singa.PowWithRet(t.data, x.data, out.data)
    else:
        singa.PowFloatWithRet(t.data, x, out.data)
    return out


def average(t, axis=None):
    if t.ndim() > 1 and axis is not None:
        return _call_singa_func(singa.Average, t.data, axis)
    return singa.SumAsFloat(t.data) / t.size()


def softmax(t, out=None):
    if out is None:
        return _call_singa_func(singa.SoftMax, t.data)
    singa.SoftMax(t.data, out.data)
    return out


def lt(t, x):
    return t < x


def le(t, x):
    return t <= x


def gt(t, x):
    return t > x


def ge(t, x):
    return t >= x


def eq(t, x):
    return t == x


def add(lhs, rhs, ret=None):
    if ret is None:
        return lhs + rhs
    if isinstance(rhs, Tensor):
        singa.Add(lhs.data, rhs.data, ret.data)
    else:
        singa.AddFloatWithRet(lhs.data, rhs, ret.data)
    return ret


def sub(lhs, rhs, ret=None):
    if ret is None:
        return lhs - rhs
    if isinstance(rhs, Tensor):
        singa.Sub(lhs.data, rhs.data, ret.data)
    else:
        singa.SubFloatWithRet(lhs.data, rhs, ret.data)
    return ret


def eltwise_mult(lhs, rhs, ret=None):
    if ret is None:
        return lhs * rhs
    if isinstance(rhs, Tensor):
        singa.EltwiseMult(lhs.data, rhs.data, ret.data)
    else:
        singa.EltwiseMultFloatWithRet(lhs.data, rhs, ret.data)
    return ret


def div(lhs, rhs, ret=None):
    if ret is None:
        return lhs / rhs
    if isinstance(rhs, Tensor):
        singa.Div(lhs.data, rhs.data, ret.data)
    else:
        singa.DivFloatWithRet(lhs.data, rhs, ret.data)
    return ret


def mult(A, B, C=None, alpha=1.0, beta=0.0):
    """C = alpha * A @ B + beta * C (matrix-vector, matrix-matrix or batched)."""
    if C is None:
        return _call_singa_func(singa.Mult, A.data, B.data)
    singa.MultWithScale(alpha, A.data, B.data, beta, C.data)
    return C


def einsum(ops, *args):
    """Two-operand einsum built from repeat / transpose / multiply / sum."""
    if len(ops) == 0:
        raise ValueError("No input operands")
    if len(args) != 2:
        raise ValueError("Currently only two operands are supported")
    inputops, outputops = ops.split('->')
    inputops = inputops.split(',')
    A, B = args
    if A.ndim() != len(inputops[0]) or B.ndim() != len(inputops[1]):
        raise ValueError("input dim doesn't match operands")
    sums = sorted(list((set(inputops[0]) | set(inputops[1])) - set(outputops)))
    broadcast_A = sorted(list(set(inputops[1]) - set(inputops[0])))
    broadcast_B = sorted(list(set(inputops[0]) - set(inputops[1])))
    outputall = sorted(list(set(inputops[0]) | set(inputops[1])))
    sums = [outputall.index(x) for x in sums]
    broadcast_idA = [inputops[1].find(x) for x in broadcast_A]
    broadcast_idB = [inputops[0].find(x) for x in broadcast_B]
    broadcast_a = [B.shape[x] for x in broadcast_idA]
    broadcast_b = [A.shape[x] for x in broadcast_idB]
    transpose_A = [(list(inputops[0]) + broadcast_A).index(x) for x in outputall]
    transpose_B = [(list(inputops[1]) + broadcast_B).index(x) for x in outputall]
    reshape_A = list(A.shape) + broadcast_a
    reshape_B = list(B.shape) + broadcast_b
    if len(broadcast_a) == 0:
        broadcast_a = [1]
    if len(broadcast_b) == 0:
        broadcast_b = [1]
    mult_A = repeat(A, product(broadcast_a))
    mult_A = mult_A.reshape(reshape_A)
    mult_A = transpose(mult_A, transpose_A)
    mult_B = repeat(B, product(broadcast_b))
    mult_B = mult_B.reshape(reshape_B)
    mult_B = transpose(mult_B, transpose_B)
    if mult_A.shape != mult_B.shape:
        raise ValueError("Error: matrix dimension mismatch")
    res = eltwise_mult(mult_A, mult_B)
    for i in sorted(sums, reverse=True):
        res = sum(res, axis=i)
    transpose_res = [sorted(list(outputops)).index(x) for x in list(outputops)]
    res = transpose(res, transpose_res)
    return res


def repeat(t, repeats, axis=None):
    return t.repeat(repeats, axis)


def tensordot(A, B, axes=2):
    """numpy.tensordot semantics, evaluated as one matrix product."""
    if type(axes) == int:
        axes_A = list(range(-axes, 0))
        axes_B = list(range(0, axes))
    else:
        axes_A, axes_B = axes
    axes_A = list(axes_A) if isinstance(axes_A, (list, tuple)) else [axes_A]
    axes_B = list(axes_B) if isinstance(axes_B, (list, tuple)) else [axes_B]
    na, nb = len(axes_A), len(axes_B)
    a_shape, nda = A.shape, A.ndim()
    b_shape, ndb = B.shape, B.ndim()
    equal = na == nb
    if equal:
        for k in range(na):
            if a_shape[axes_A[k]] != b_shape[axes_B[k]]:
                equal = False
                break
            if axes_A[k] < 0:
                axes_A[k] += nda
            if axes_B[k] < 0:
                axes_B[k] += ndb
    if not equal:
        raise ValueError("shape-mismatch for sum")

    notin = [k for k in range(nda) if k not in axes_A]
    newaxes_a = notin + axes_A
    N2 = 1
    for axis in axes_A:
        N2 *= a_shape[axis]
    N1 = 1
    for ax in notin:
        N1 *= a_shape[ax]
    newshape_a = (N1, N2)
    olda = [a_shape[axis] for axis in notin]
    notin = [k for k in range(ndb) if k not in axes_B]
    newaxes_b = axes_B + notin
    N2 = 1
    for axis in axes_B:
        N2 *= b_shape[axis]
    N1 = 1
    for bx in notin:
        N1 *= b_shape[bx]
    newshape_b = (N2, N1)
    oldb = [b_shape[axis] for axis in notin]

    A = transpose(A, newaxes_a)
    B = transpose(B, newaxes_b)
    at = reshape(A, newshape_a)
    bt = reshape(B, newshape_b)
    res = mult(at, bt)
    if len(olda + oldb) == 0:
        res = res.reshape((1, 1))
    else:
        res = res.reshape(tuple(olda + oldb))
    return res


def axpy(alpha, x, y):
    singa.Axpy(float(alpha), x.data, y.data)
    return y


def bernoulli(p, t):
    singa.Bernoulli(float(p), t.data)
    return t


def gaussian(mean, std, t):
    singa.Gaussian(float(mean), float(std), t.data)
    return t


def uniform(low, high, t):
    singa.Uniform(float(low), float(high), t.data)
    return t


def add_column(alpha, v, beta, M):
    singa.AddColumnWithScale(float(alpha), float(beta), v.data, M.data)
    return M


def add_row(alpha, v, beta, M):
    singa.AddRowWithScale(alpha, beta, v.data, M.data)
    return M


def sum_columns(M):
    assert M.ndim() == 2, 'M.nDim() is supposed to be 2'
    ret = Tensor((M.shape[0], 1), M.data.device())
    singa.SumColumns(M.data, ret.data)
    return ret


def sum_rows(M):
    assert M.ndim() == 2, 'M.nDim() is supposed to be 2'
    ret = Tensor((1, M.shape[1]), M.data.device())
    singa.SumRows(M.data, ret.data)
    return ret


def copy_from_numpy(data, np_array):
    """Copy a numpy array into a core tensor of the same size."""
    assert np_array.size == data.Size(), 'tensor shape should be the same'
    if not np_array.ndim == 1:
        np_array = np_array.flatten()
    dt = np_array.dtype
    if dt == np.float32:
        data.CopyFloatDataFromHostPtr(np_array)
    elif dt == np.float16:
        data.CopyHalfFloatDataFromHostPtr(np_array)
    elif dt == np.int32 or dt == np.int64:
        data.CopyIntDataFromHostPtr(np_array.astype(np.int32))
    elif dt == np.float64:
        data.CopyFloatDataFromHostPtr(np_array.astype(np.float32))
    else:
        raise NotImplementedError('Not implemented yet for ', dt)


def concatenate(tensors, axis):
    ctensors = singa.VecTensor()
    for t in tensors:
        ctensors.append(t.data)
    return _call_singa_func(singa.ConcatOn, ctensors, axis)


def random(shape, device=None):
    ret = Tensor(shape, device=device or get_default_device())
    ret.uniform(0, 1)
    return ret


def zeros(shape, device=None):
    ret = Tensor(shape, device=device or get_default_device())
    ret.set_value(0.0)
    return ret


def ones(shape, device=None):
    ret = Tensor(shape, device=device or get_default_device())
    ret.set_value(1.0)
    return ret
