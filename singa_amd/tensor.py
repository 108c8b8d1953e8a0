"""User-facing Tensor: a core tensor plus autograd bookkeeping
(``requires_grad``, ``stores_grad``, ``creator``, ``name``), numpy interop and
the module-level math functions.

The arithmetic / comparison operators, the in-place operators and the unary
math functions are generated from tables that pair each name with its core
kernel entry points.  ``tensordot`` and ``einsum`` both lower to ONE (batched)
matrix product: einsum classifies every label as batch / contracted / free,
pre-reduces labels that appear in a single operand only, and lets the GEMM do
the contraction, instead of materialising the broadcast product and summing
it axis by axis.

API parity: reference python/singa/tensor.py (class Tensor :73-779, module
functions :789-1841).  Behavioural notes kept from the reference:
``from_numpy`` narrows float64 -> float32 and int64 -> int32; ``sum`` with an
axis returns a tensor with that axis removed; ``PlaceHolder`` is an alias of
``Tensor``.
"""
import itertools
import math
import operator
import re

import numpy as np

from . import singa_wrap as singa
from .device import get_default_device

int32 = singa.kInt
float16 = singa.kFloat16
float32 = singa.kFloat32
bfloat16 = singa.kBFloat16
CTensor = singa.Tensor

# numpy element type -> tensor dtype (64-bit types are narrowed)
_NP2DTYPE = {np.float32: float32, np.float64: float32, np.float16: float16,
             np.int32: int32, np.int64: int32}
_DTYPE_ALIASES = {'int': singa.kInt, 'float': singa.kFloat32}
_DTYPE_CODES = (singa.kInt, singa.kFloat16, singa.kFloat32, singa.kBFloat16,
                singa.kDouble)


def _axis_range(key, extent):
    """[start, stop) selected on an axis of ``extent`` elements by an integer
    or a unit-step slice."""
    if isinstance(key, (int, np.integer)) and not isinstance(key, bool):
        lo = int(key) + (extent if key < 0 else 0)
        hi = lo + 1
    elif isinstance(key, slice):
        lo = key.start or 0
        hi = key.stop or extent
        lo += extent if lo < 0 else 0
        hi += extent if hi < 0 else 0
    else:
        raise ValueError("an index is an integer or a unit-step slice, "
                         "got %r" % (key,))
    if not 0 <= lo < hi <= extent:
        raise ValueError("index %r outside an axis of %d" % (key, extent))
    return lo, hi


class Tensor(object):
    tensor_count = 0
    _serial = itertools.count(1)
    dtype_name = {float16: "float16", float32: "float32", int32: "int32",
                  bfloat16: "bfloat16"}

    def __init__(self, shape=(), device=None, dtype=None, data=None,
                 requires_grad=True, stores_grad=False, creator=None,
                 name=None):
        device = device if device is not None else get_default_device()
        from_array = isinstance(data, np.ndarray)
        if dtype is None:
            # a numpy array brings its own element type
            dtype = _NP2DTYPE.get(data.dtype.type, float32) if from_array \
                else float32
        if from_array:
            core = CTensor(list(data.shape), device, dtype)
            copy_from_numpy(core, data)
        elif isinstance(data, CTensor):
            assert data.device().id() == device.id(), 'not the same device'
            core = data
        else:
            core = CTensor(list(shape), device, dtype)
        self._adopt(core, device, requires_grad, stores_grad, creator, name)

    def _adopt(self, core, device=None, requires_grad=True, stores_grad=False,
               creator=None, name=None):
        """Bind this object to a core tensor and start its autograd record."""
        self.data = core
        self.shape = tuple(core.shape())
        self.device = device if device is not None else core.device()
        self.dtype = core.data_type()
        self.requires_grad, self.stores_grad = requires_grad, stores_grad
        self.name = name
        if name is None:
            self.name = 'Dummy#%d' % Tensor.tensor_count
            Tensor.tensor_count += 1
        if creator is None:
            from . import autograd
            creator = autograd.Dummy(self, name)
        self.creator = creator
        return self

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

    # ---- indexing ------------------------------------------------------------------
    def __getitem__(self, keys):
        keys = keys if isinstance(keys, tuple) else (keys,)
        core = None
        for axis, key in enumerate(keys):
            lo, hi = _axis_range(key, self.shape[axis])
            if (lo, hi) != (0, self.shape[axis]) or core is None:
                core = singa.SliceOn(self.data if core is None else core,
                                     lo, hi, axis)
        return _wrap(core if core is not None else self.data.Clone())

    def __setitem__(self, keys, value):
        """``t[:] = other`` (whole-tensor assignment) as used by ``sum(out=)``."""
        if keys != slice(None, None, None):
            raise NotImplementedError("only t[:] = value is supported")
        self.copy_from(value)

    # ---- shape and layout ----------------------------------------------------------
    def is_dummy(self):
        return re.match(r'Dummy#\d+', str(self.name)) is not None

    def is_empty(self):
        return self.ndim() == 0

    def transpose(self, axes=None):
        if axes is None:
            return _wrap(singa.DefaultTranspose(self.data))
        if len(axes) != len(self.shape):
            raise ValueError('%d axes given for %d dimensions' % (
                len(axes), len(self.shape)))
        return _wrap(singa.Transpose(self.data, list(axes)))

    def T(self):
        return self.transpose()

    def contiguous(self):
        return _wrap(singa.Contiguous(self.data))

    def reshape(self, shape):
        assert product(self.shape) == product(shape), \
            'reshape keeps the number of elements: %s -> %s' % (
                self.shape, tuple(shape))
        return _wrap(singa.Reshape(self.data, list(shape)))

    def reset_like(self, t):
        self.data.ResetLike(t.data)
        self.shape, self.device, self.dtype = t.shape, t.device, t.dtype

    def repeat(self, repeats, axis):
        nd = self.ndim()
        if isinstance(repeats, int):
            counts = [repeats]
            # the core takes 9999 for "flatten first"
            axis = 9999 if axis is None else axis + (nd if axis < 0 else 0)
        elif isinstance(repeats, (tuple, list)):
            if axis is None:
                raise ValueError("per-element repeat counts need an axis")
            counts = list(repeats)
            axis += nd if axis < 0 else 0
        else:
            raise ValueError("repeats is an int or a sequence of ints, got "
                             "%r" % (repeats,))
        if min(counts, default=0) < 0:
            raise ValueError("negative repeat count in %r" % (repeats,))
        return _wrap(self.data.Repeat(counts, axis))

    # ---- dtype and placement -----------------------------------------------------------
    @staticmethod
    def _dtype_code(dtype):
        if dtype in _DTYPE_CODES:
            return dtype
        try:
            return _DTYPE_ALIASES[dtype]
        except (KeyError, TypeError):
            raise TypeError("invalid data type %s" % dtype)

    def as_type(self, dtype):
        return _wrap(singa.Tensor(self.data.AsType(self._dtype_code(dtype))))

    def to_type(self, dtype):
        assert self.data.initialized()
        self.dtype = self._dtype_code(dtype)
        self.data.ToType(self.dtype)
        return self

    def to_device(self, device):
        self.data.ToDevice(device)
        self.device = device

    def to_host(self):
        self.data.ToHost()
        self.device = get_default_device()

    # ---- values ------------------------------------------------------------------------
    def set_value(self, x, inplace=True):
        _inplace_only(inplace)
        self.data.SetFloatValue(float(x))
        return self

    def copy_from_numpy(self, np_array, offset=0):
        copy_from_numpy(self.data, np_array)

    def copy_data(self, t):
        assert isinstance(t, Tensor), 'the source must be a Tensor'
        assert t.size() == self.size(), \
            "copy between tensors of %d and %d elements" % (t.size(),
                                                            self.size())
        self.data.CopyData(t.data)

    def copy_from(self, t, offset=0):
        for kind, load in ((Tensor, self.copy_data),
                           (np.ndarray, self.copy_from_numpy)):
            if isinstance(t, kind):
                return load(t)
        raise ValueError("cannot copy from a %s" % type(t).__name__)

    def clone(self):
        return _wrap(self.data.Clone())

    deepcopy = clone

    def copy(self):
        """Shallow copy sharing the underlying memory."""
        return _wrap(CTensor(self.data))

    # ---- operators that are not table-generated ------------------------------------------
    def __floordiv__(self, rhs):
        return _wrap(singa.Floor((self / rhs).data))

    def __eq__(self, rhs):
        if rhs is None:
            return False
        return _binary_eq(self, rhs)

    __hash__ = object.__hash__

    def __bool__(self):
        # ``if self.b:`` in layer code asks "is there a tensor", not its value
        return True

    def __neg__(self):
        return self * -1.0

    def __repr__(self):
        return "%s, %s" % (np.array2string(to_numpy(self)),
                           self.dtype_name.get(self.dtype, str(self.dtype)))


PlaceHolder = Tensor


def _inplace_only(inplace):
    if not inplace:
        raise NotImplementedError("only the in-place form exists")


def _wrap(ct):
    """A fresh user tensor around a core tensor."""
    return Tensor.__new__(Tensor)._adopt(ct)


def from_raw_tensor(t):
    return _wrap(t)


def from_raw_tensors(tt):
    return [_wrap(t) for t in tt]


def _call_singa_func(_singa_func, *args):
    return _wrap(_singa_func(*args))


# ---------------------------------------------------------------------------
# generated operators
# ---------------------------------------------------------------------------
def _binary(with_tensor, with_scalar):
    def op(self, rhs):
        if isinstance(rhs, Tensor):
            return _wrap(with_tensor(self.data, rhs.data))
        return _wrap(with_scalar(self.data, float(rhs)))
    return op


def _inplace(name):
    def op(self, x):
        # the core tensor implements the in-place protocol itself
        self.data = getattr(self.data, name)(
            x.data if isinstance(x, Tensor) else float(x))
        return self
    return op


def _reflected(name):
    def op(self, lhs):
        # scalar (op) tensor: fill a tensor with the scalar, combine in place
        filled = Tensor(self.shape, self.device, self.dtype)
        filled.set_value(float(lhs))
        return getattr(filled, name)(self)
    return op


for _name, _pair in {
        '__add__': (singa.__add__, singa.AddFloat),
        '__sub__': (singa.__sub__, singa.SubFloat),
        '__mul__': (singa.__mul__, singa.MultFloat),
        '__div__': (singa.__div__, singa.DivFloat),
        '__truediv__': (singa.__div__, singa.DivFloat),
        '__lt__': (singa.__lt__, singa.LTFloat),
        '__le__': (singa.__le__, singa.LEFloat),
        '__gt__': (singa.__gt__, singa.GTFloat),
        '__ge__': (singa.__ge__, singa.GEFloat)}.items():
    setattr(Tensor, _name, _binary(*_pair))
_binary_eq = _binary(singa.__eq__, singa.EQFloat)

def _core_query(method):
    def fn(self):
        return getattr(self.data, method)()
    return fn


def _random_fill(kernel, names):
    """``t.<fill>(p...)``: the distribution's parameters by position or by
    the reference's keyword names (python/singa/tensor.py: bernoulli(p),
    gaussian(mean, std), uniform(low, high))."""
    def fn(self, *params, inplace=True, **named):
        _inplace_only(inplace)
        if len(params) > len(names):
            raise TypeError("at most %d parameters (%s)" % (
                len(names), ", ".join(names)))
        values = list(params)
        for n in names[len(params):]:
            if n not in named:
                raise TypeError("missing parameter %r" % n)
            values.append(named.pop(n))
        if named:
            raise TypeError("unexpected parameters %s" % sorted(named))
        kernel(*(float(v) for v in values), self.data)
        return self
    return fn


def _with_vector(kernel):
    # deprecated row / column helpers (kept for old scripts)
    def fn(self, v):
        kernel(v.data, self.data)
    return fn


for _name, _method in {'ndim': 'nDim', 'size': 'Size', 'memsize': 'MemSize',
                       'is_transpose': 'transpose', 'l1': 'L1',
                       'l2': 'L2'}.items():
    setattr(Tensor, _name, _core_query(_method))

for _name, (_kernel, _params) in {
        'bernoulli': (singa.Bernoulli, ('p',)),
        'gaussian': (singa.Gaussian, ('mean', 'std')),
        'uniform': (singa.Uniform, ('low', 'high'))}.items():
    setattr(Tensor, _name, _random_fill(_kernel, _params))

for _name, _kernel in {'add_column': singa.AddColumn, 'add_row': singa.AddRow,
                       'div_column': singa.DivColumn, 'div_row': singa.DivRow,
                       'mult_column': singa.MultColumn,
                       'mult_row': singa.MultRow}.items():
    setattr(Tensor, _name, _with_vector(_kernel))

for _name in ('__iadd__', '__isub__', '__imul__', '__itruediv__'):
    setattr(Tensor, _name, _inplace(_name))

for _name, _target in {'__radd__': '__iadd__', '__rsub__': '__isub__',
                       '__rmul__': '__imul__', '__rdiv__': '__itruediv__',
                       '__rtruediv__': '__itruediv__'}.items():
    setattr(Tensor, _name, _reflected(_target))


def _unary(kernel):
    def fn(t):
        return _wrap(kernel(t.data))
    return fn


abs = _unary(singa.Abs)
exp = _unary(singa.Exp)
ceil = _unary(singa.Ceil)
log = _unary(singa.Log)
sigmoid = _unary(singa.Sigmoid)
sign = _unary(singa.Sign)
sqrt = _unary(singa.Sqrt)
square = _unary(singa.Square)
tanh = _unary(singa.Tanh)
relu = _unary(singa.ReLU)


def contiguous(tensor):
    return _wrap(singa.Contiguous(tensor.data))


def _with_ret(operator, tensor_kernel, scalar_kernel):
    """``f(lhs, rhs, ret=None)``: operator form, or write into ``ret``."""
    def fn(lhs, rhs, ret=None):
        if ret is None:
            return operator(lhs, rhs)
        if isinstance(rhs, Tensor):
            tensor_kernel(lhs.data, rhs.data, ret.data)
        else:
            scalar_kernel(lhs.data, rhs, ret.data)
        return ret
    return fn


add = _with_ret(lambda a, b: a + b, singa.Add, singa.AddFloatWithRet)
sub = _with_ret(lambda a, b: a - b, singa.Sub, singa.SubFloatWithRet)
eltwise_mult = _with_ret(lambda a, b: a * b, singa.EltwiseMult,
                         singa.EltwiseMultFloatWithRet)
div = _with_ret(lambda a, b: a / b, singa.Div, singa.DivFloatWithRet)


# comparison functions: the operator forms under their reference names
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


# ---------------------------------------------------------------------------
# construction and conversion
# ---------------------------------------------------------------------------
def _filled(shape, device, dtype, value):
    t = Tensor(shape, device if device is not None else get_default_device(),
               dtype)
    return t.set_value(value)


def zeros_like(t):
    return _filled(t.shape, t.device, t.dtype, 0.0)


def ones_like(t):
    return _filled(t.shape, t.device, t.dtype, 1.0)


def zeros(shape, device=None):
    return _filled(shape, device, None, 0.0)


def ones(shape, device=None):
    return _filled(shape, device, None, 1.0)


def random(shape, device=None):
    return Tensor(shape, device=device or get_default_device()).uniform(0, 1)


def product(shape):
    return math.prod(shape)


def sizeof(dtype):
    return singa.SizeOf(dtype)


def reshape(tensor, shape):
    return _wrap(singa.Reshape(tensor.data, list(shape)))


def transpose(t, axes=None):
    return t.transpose(axes)


def repeat(t, repeats, axis=None):
    return t.repeat(repeats, axis)


def concatenate(tensors, axis):
    return _wrap(singa.ConcatOn(singa.VecTensor([t.data for t in tensors]),
                                axis))


def copy_data_to_from(dst, src, size, dst_offset=0, src_offset=0):
    singa.CopyDataToFrom(dst.data, src.data, size, dst_offset, src_offset)


_HOST_UPLOAD = {np.dtype(np.float32): ('CopyFloatDataFromHostPtr', None),
                np.dtype(np.float64): ('CopyFloatDataFromHostPtr', np.float32),
                np.dtype(np.float16): ('CopyHalfFloatDataFromHostPtr', None),
                np.dtype(np.int32): ('CopyIntDataFromHostPtr', None),
                np.dtype(np.int64): ('CopyIntDataFromHostPtr', np.int32)}


def copy_from_numpy(data, np_array):
    """Copy a numpy array into a core tensor of the same size."""
    assert np_array.size == data.Size(), (
        'array of %d elements into a tensor of %d' % (np_array.size,
                                                      data.Size()))
    try:
        method, narrow = _HOST_UPLOAD[np_array.dtype]
    except KeyError:
        raise NotImplementedError('Not implemented yet for ', np_array.dtype)
    flat = np_array.reshape(-1)
    getattr(data, method)(flat if narrow is None else flat.astype(narrow))


def from_numpy(np_array, dev=None):
    """A host tensor holding the array's values, moved to ``dev`` if given."""
    assert type(np_array) is np.ndarray, \
        'from_numpy takes a numpy array, got %s' % type(np_array).__name__
    kind = np_array.dtype.type
    assert kind in _NP2DTYPE, 'no tensor dtype for numpy %s' % np_array.dtype
    t = Tensor(np_array.shape, dtype=_NP2DTYPE[kind])
    copy_from_numpy(t.data, np_array)
    if dev:
        t.to_device(dev)
    return t


def to_host(t):
    host = t.clone()
    host.to_host()
    return host


_HOST_DOWNLOAD = {float32: 'GetFloatValue', float16: 'GetHalfFloatValue',
                  int32: 'GetIntValue', bfloat16: 'GetFloatValue'}


def to_numpy(t):
    getter = _HOST_DOWNLOAD.get(t.data.data_type(), 'get_value')
    return getattr(t.data, getter)().reshape(t.shape)


# ---------------------------------------------------------------------------
# reductions and products
# ---------------------------------------------------------------------------
def sum(t, axis=None, out=None):
    """Sum over ``axis`` (None, int or tuple); the reduced axes are removed.
    One matrix-vector product against a vector of ones."""
    nd = t.ndim()
    if axis is None:
        over = list(range(nd))
    elif isinstance(axis, (int, np.integer)):
        over = [int(axis)]
    elif isinstance(axis, (tuple, list)):
        over = list(axis)
    else:
        raise ValueError("axis must be None, int or tuple")
    over = [a + nd if a < 0 else a for a in over]
    ones_ = _filled(tuple(t.shape[a] for a in over), t.device, t.dtype, 1.0)
    total = tensordot(t, ones_, (over, list(range(len(over)))))
    if out is not None:
        if out.shape != total.shape:
            raise ValueError('out has shape %s, the sum %s' % (out.shape,
                                                               total.shape))
        out.copy_from(total)
    return total if out is None else out


def pow(t, x, out=None):
    tensor_exponent = isinstance(x, Tensor)
    if out is None:
        return _wrap(singa.Pow(t.data, x.data) if tensor_exponent
                     else singa.PowFloat(t.data, x))
    if tensor_exponent:
        singa.PowWithRet(t.data, x.data, out.data)
    else:
        singa.PowFloatWithRet(t.data, x, out.data)
    return out


def average(t, axis=None):
    if axis is None or t.ndim() <= 1:
        return singa.SumAsFloat(t.data) / t.size()
    return _wrap(singa.Average(t.data, axis))


def softmax(t, out=None):
    if out is None:
        return _wrap(singa.SoftMax(t.data))
    singa.SoftMax(t.data, out.data)
    return out


def mult(A, B, C=None, alpha=1.0, beta=0.0):
    """C = alpha * A @ B + beta * C (matrix-vector, matrix-matrix or batched)."""
    if C is None:
        return _wrap(singa.Mult(A.data, B.data))
    singa.MultWithScale(alpha, A.data, B.data, beta, C.data)
    return C


def _axes_list(axes, nd):
    axes = list(axes) if isinstance(axes, (list, tuple)) else [axes]
    return [int(a) + nd if a < 0 else int(a) for a in axes]


def tensordot(A, B, axes=2):
    """numpy.tensordot semantics, evaluated as one matrix product: the
    contracted axes are moved to the end of ``A`` and the front of ``B``."""
    nda, ndb = A.ndim(), B.ndim()
    if isinstance(axes, (int, np.integer)):
        con_a = list(range(nda - axes, nda))
        con_b = list(range(axes))
    else:
        con_a, con_b = _axes_list(axes[0], nda), _axes_list(axes[1], ndb)
    if len(con_a) != len(con_b) or any(
            A.shape[i] != B.shape[j] for i, j in zip(con_a, con_b)):
        raise ValueError("contracted axes %s / %s differ in extent" % (
            con_a, con_b))
    keep_a = [i for i in range(nda) if i not in con_a]
    keep_b = [j for j in range(ndb) if j not in con_b]
    free_a = [A.shape[i] for i in keep_a]
    free_b = [B.shape[j] for j in keep_b]
    depth = product(A.shape[i] for i in con_a)
    left = reshape(A.transpose(keep_a + con_a), (product(free_a), depth))
    right = reshape(B.transpose(con_b + keep_b), (depth, product(free_b)))
    return mult(left, right).reshape(tuple(free_a + free_b) or (1, 1))


def einsum(ops, *args):
    """Two-operand einsum, e.g. ``einsum('bij,bjk->bik', x, y)``, as one
    batched matrix product.

    Labels present in both operands and the output index the batch; labels
    present in both operands only are contracted by the GEMM; the remaining
    output labels are the free rows (first operand) and columns (second);
    a label that occurs in a single operand and not in the output is summed
    away before the product."""
    if not ops or len(args) != 2:
        raise ValueError("einsum takes a subscript string and two operands")
    spec_in, _, spec_out = ops.replace(' ', '').partition('->')
    la, lb = spec_in.split(',')
    A, B = args
    if A.ndim() != len(la) or B.ndim() != len(lb):
        raise ValueError("subscripts %r do not match operand ranks %d, %d" % (
            spec_in, A.ndim(), B.ndim()))
    if len(set(la)) != len(la) or len(set(lb)) != len(lb):
        raise ValueError("repeated labels within one operand are not supported")
    if not set(spec_out) <= set(la) | set(lb):
        raise ValueError("output label missing from the operands")
    extent = {}
    for labels, t in ((la, A), (lb, B)):
        for c, n in zip(labels, t.shape):
            if extent.setdefault(c, n) != n:
                raise ValueError("label %r has extents %d and %d" % (
                    c, extent[c], n))

    def presum(t, labels, other):
        gone = [i for i, c in enumerate(labels)
                if c not in other and c not in spec_out]
        if gone:
            t = sum(t, axis=tuple(gone))
            labels = ''.join(c for i, c in enumerate(labels) if i not in gone)
        return t, labels

    A, la = presum(A, la, lb)
    B, lb = presum(B, lb, la)
    batch = [c for c in la if c in lb and c in spec_out]
    inner = [c for c in la if c in lb and c not in spec_out]
    rows = [c for c in la if c not in lb]
    cols = [c for c in lb if c not in la]

    def count(labels):
        return product(extent[c] for c in labels)

    left = A.transpose([la.index(c) for c in batch + rows + inner])
    right = B.transpose([lb.index(c) for c in batch + inner + cols])
    if not inner and not rows and not cols:
        res = left.contiguous() * right.contiguous()       # pure Hadamard
    else:
        left = reshape(left, (count(batch), count(rows), count(inner)))
        right = reshape(right, (count(batch), count(inner), count(cols)))
        res = mult(left, right)
    natural = batch + rows + cols
    res = reshape(res, tuple(extent[c] for c in natural) or (1,))
    if list(spec_out) != natural:
        res = res.transpose([natural.index(c) for c in spec_out])
    return res


# ---------------------------------------------------------------------------
# BLAS-1 style helpers and random fills
# ---------------------------------------------------------------------------
def axpy(alpha, x, y):
    singa.Axpy(float(alpha), x.data, y.data)
    return y


def bernoulli(p, t):
    return t.bernoulli(p)


def gaussian(mean, std, t):
    return t.gaussian(mean, std)


def uniform(low, high, t):
    return t.uniform(low, high)


def add_column(alpha, v, beta, M):
    singa.AddColumnWithScale(float(alpha), float(beta), v.data, M.data)
    return M


def add_row(alpha, v, beta, M):
    singa.AddRowWithScale(alpha, beta, v.data, M.data)
    return M


def _matrix_reduce(kernel, shape_of):
    def fn(M):
        assert M.ndim() == 2, 'expects a matrix, got %d dimensions' % M.ndim()
        ret = Tensor(shape_of(M), M.data.device())
        kernel(M.data, ret.data)
        return ret
    return fn


sum_columns = _matrix_reduce(singa.SumColumns, lambda M: (M.shape[0], 1))
sum_rows = _matrix_reduce(singa.SumRows, lambda M: (1, M.shape[1]))
