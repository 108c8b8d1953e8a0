"""Define-by-run automatic differentiation.

``Operator._do_forward`` records, for every input, (creator op, python id of
the input, the input itself when it stores a gradient, stores_grad); outputs
are wrapped into tensors whose ``creator`` is the op.  ``backward`` walks this
tape from the loss in dependency order and is a *generator*: it yields
``(parameter, gradient)`` as soon as a parameter's gradient is complete, which
is what lets the data-parallel optimiser start reducing the gradients of the
last layers while earlier layers are still back-propagating.

API parity: reference python/singa/autograd.py (Operator :227-341, Dummy
:344-364, infer_dependency :71-102, backward :128-224, ~100 operators).  Ops
the reference assembles from per-index copies (Gather, Slice, Tile, Expand,
Pad, Embedding, OneHot, reductions ...) are single device kernels here.
"""
from collections import Counter

import numpy as np

from . import singa_wrap as singa
from . import tensor
from . import utils
from .tensor import Tensor

CTensor = singa.Tensor
training = False


# ---------------------------------------------------------------------------
# broadcasting helpers
# ---------------------------------------------------------------------------
def axis_helper(y_shape, x_shape):
    """Axes of ``y_shape`` along which ``x_shape`` was broadcast (missing
    leading axes and axes whose extents differ)."""
    lead = len(y_shape) - len(x_shape)
    return tuple(i for i, n in enumerate(y_shape)
                 if i < lead or x_shape[i - lead] != n)


def back_broadcast(y_shape, x_shape, x):
    """Reduce gradient ``x`` (core tensor of shape y_shape) to ``x_shape``."""
    if list(y_shape) == list(x_shape):
        return x
    summed = singa.ReduceND(x, list(axis_helper(y_shape, x_shape)), "sum",
                            keepdims=False)
    return singa.Reshape(summed, list(x_shape))


def make_slice(arr, axis, i):
    """Index list selecting position(s) ``i`` along ``axis`` of ``arr``
    (reference python/singa/autograd.py:1631)."""
    return [i if d == axis else slice(None) for d in range(arr.ndim)]


def _filled_core(shape, dev, dtype, value):
    t = CTensor(list(shape), dev, dtype)
    t.SetFloatValue(float(value))
    return t


# ---------------------------------------------------------------------------
# the tape
# ---------------------------------------------------------------------------
def infer_dependency(op):
    """Count, for the sub-graph ending at ``op``, how many consumers every
    producing op and every tensor (by ``Tensor.uid``) has."""
    op_uses, tensor_uses = Counter(), Counter()
    frontier = [op]
    while frontier:
        for producer, uid, _, _ in frontier.pop().src:
            tensor_uses[uid] += 1
            op_uses[producer] += 1
            if op_uses[producer] == 1:      # first sighting: expand it
                frontier.append(producer)
    return op_uses, tensor_uses


def gradients(y, dy=None):
    """Dictionary id(param) -> gradient for every tensor that stores grad."""
    return {id(p): g for p, g in backward(y, dy)}


def _seed_gradient(y, dy):
    """The upstream gradient in the form ``y``'s creator expects: the loss
    operators take a python float, every other op a core tensor."""
    if isinstance(dy, Tensor):
        return dy.data
    value = 1.0 if dy is None else float(dy)
    if getattr(y.creator, "takes_scalar_dy", False):
        return value
    return _filled_core(y.shape, y.device, y.dtype, value)


def _offer_partial_sum(op, pending):
    """An op whose kernel can add onto an existing gradient is handed the
    partial sum already collected for its first input; what it returns then
    replaces that partial sum instead of being added in a pass of its own."""
    if not (getattr(op, "fuses_dx_addend", False) and op.src):
        return False
    producer, uid = op.src[0][0], op.src[0][1]
    collected = pending.get(producer)
    if collected is None or collected[producer.y_id2idx[uid]] is None:
        return False
    op.dx_addend = collected[producer.y_id2idx[uid]]
    return True


def backward(y, dy=None):
    """Generator over (param, grad) pairs, in the order gradients complete.

    ``pending[op]`` holds the output gradients of ``op`` gathered so far; an
    op runs its backward once every consumer has reported (its count in
    ``ops_left`` reaches zero), and a gradient-storing tensor is yielded the
    moment its last consumer has reported, so the caller can start reducing
    it while the sweep continues."""
    assert isinstance(y, Tensor), "backward starts from a Tensor"
    assert y.size() == 1, (
        "backward starts from a single value, y has %d elements" % y.size())
    ops_left, tensors_left = infer_dependency(y.creator)
    seed = _seed_gradient(y, dy)
    if y.stores_grad:
        g = seed if not isinstance(seed, float) \
            else _filled_core([1], y.device, tensor.float32, seed)
        yield y, Tensor(device=g.device(), data=g)

    pending = {}
    runnable = [(y.creator, (seed,))]
    while runnable:
        op, dys = runnable.pop()
        if isinstance(op, Dummy) or not op.requires_grad:
            continue
        folded = _offer_partial_sum(op, pending)
        dxs = op._do_backward(*dys)
        assert len(dxs) == len(op.src), (
            "%s returned %d gradients for %d inputs" %
            (op.name, len(dxs), len(op.src)))
        for position, (source, dx) in enumerate(zip(op.src, dxs)):
            producer, uid, kept, keeps_grad = source
            if isinstance(producer, Dummy) and not producer.stores_grad:
                continue
            slot = producer.y_id2idx[uid]
            collected = pending.setdefault(producer,
                                           [None] * len(producer.y_id2idx))
            if collected[slot] is None or (folded and position == 0):
                collected[slot] = dx
            elif dx is not None:
                # a tensor consumed by several ops: gradients add up
                collected[slot] = singa.__add__(collected[slot], dx)
            ops_left[producer] -= 1
            tensors_left[uid] -= 1
            if keeps_grad and tensors_left[uid] == 0:
                g = collected[slot]
                yield kept, Tensor(device=g.device(), data=g,
                                   name=producer.grad_name(slot))
            if ops_left[producer] == 0:
                complete = pending.pop(producer)
                if producer.requires_grad is True:
                    runnable.append((producer, complete))
        del op          # release everything the op cached


class _AllTrue(object):
    """Default of ``Operator.x_needs_grad`` for ops driven by hand through
    ``forward`` / ``backward`` rather than ``__call__``."""

    def __getitem__(self, idx):
        return True


class Operator(object):
    """Forward + backward of one tensor computation on core tensors.

    Calling an operator on Tensors records it on the tape: ``src`` lists, per
    input, (creator op, input uid, the input itself when it stores a gradient,
    stores_grad); the outputs are new Tensors whose ``creator`` is the op and
    ``y_id2idx`` maps their uids to output positions."""

    op_count = 0
    x_needs_grad = _AllTrue()

    def __init__(self, name=None):
        if name is None:
            name = "%s#%d" % (type(self).__name__, Operator.op_count)
            Operator.op_count += 1
        self.name = name

    def __call__(self, *xs):
        return self._do_forward(*xs)

    def output_name(self, idx):
        return "%s:%d" % (self.name, idx)

    def grad_name(self, idx):
        return self.output_name(idx) + "_g"

    def _do_forward(self, *xs):
        for x in xs:
            assert isinstance(x, Tensor), (
                "%s takes Tensors, got %s" % (self.name, type(x).__name__))
        # inputs whose gradient nobody consumes (data fed to the network):
        # ops may return None for them instead of computing it
        self.x_needs_grad = tuple(
            x.stores_grad or
            (x.requires_grad and not isinstance(x.creator, Dummy))
            for x in xs)
        # an op fed only by such inputs has nothing to back-propagate, and
        # neither has anything computed from its outputs alone (e.g. an
        # attention mask derived from the input ids)
        self.requires_grad = any(self.x_needs_grad)
        self.src = [(x.creator, x.uid, x if x.stores_grad else None,
                     x.stores_grad) for x in xs]
        produced = self.forward(*(x.data for x in xs))
        if not isinstance(produced, tuple):
            produced = (produced,)
        ys = tuple(
            Tensor(device=core.device(), data=core, creator=self,
                   requires_grad=self.requires_grad,
                   name=self.output_name(i))
            for i, core in enumerate(produced))
        self.y_id2idx = {y.uid: i for i, y in enumerate(ys)}
        return ys

    def _do_backward(self, *dys):
        dxs = self.backward(*dys)
        return tuple(dxs) if isinstance(dxs, (list, tuple)) else (dxs,)

    def forward(self, *xs):
        raise NotImplementedError

    def backward(self, *dys):
        raise NotImplementedError

    def get_params(self):
        return []


class Dummy(Operator):
    """Leaf placeholder: the creator of tensors not produced by an op.
    Attribute look-ups it cannot answer go to its tensor."""

    requires_grad = False

    def __init__(self, tensor, name=None):
        Operator.__init__(self, name)
        self.tensor = tensor
        self.src = []
        self.y_id2idx = {tensor.uid: 0}

    def output_name(self, idx):
        return self.name

    def grad_name(self, idx):
        return "%s_g" % self.name

    def __getattr__(self, name):
        if name == "tensor":            # not bound yet (copy / unpickle)
            raise AttributeError(name)
        return getattr(self.tensor, name)


def _scalar_like(x, value):
    t = CTensor(list(x.shape()), x.device(), x.data_type())
    t.SetFloatValue(value)
    return t


def _apply_dy(dx, dy):
    """Scale ``dx`` by an upstream gradient that is a float or a 1-element
    tensor."""
    if isinstance(dy, float):
        if dy != 1.0:
            dx = singa.MultFloat(dx, dy)
        return dx
    return singa.__mul__(dx, dy)


# ---------------------------------------------------------------------------
# n-ary element-wise
# ---------------------------------------------------------------------------
class Mean(Operator):
    """Element-wise mean of the inputs."""

    def forward(self, *l):
        assert len(l) > 0
        self.l = len(l)
        x = l[0]
        for t in l[1:]:
            x = singa.__add__(x, t)
        return singa.MultFloat(x, 1.0 / len(l))

    def backward(self, dy):
        return [singa.MultFloat(dy, 1.0 / self.l)] * self.l


def mean(*l):
    return Mean()(*l)[0]


class Sum(Operator):
    """Element-wise sum of the inputs."""

    def forward(self, *l):
        assert len(l) > 0
        self.l = len(l)
        x = l[0]
        if len(l) == 1:
            return singa.AddFloat(x, 0.0)
        for t in l[1:]:
            x = singa.__add__(x, t)
        return x

    def backward(self, dy):
        return [dy] * self.l


def sum(*l):
    return Sum()(*l)[0]


def add_all(*xs):
    assert len(xs) > 2
    y = add(xs[0], xs[1])
    for x in xs[2:]:
        y = add(y, x)
    return


class _MinMax(Operator):
    """Element-wise min / max over n broadcastable inputs."""
    mode = "max"

    def forward(self, *x):
        assert len(x) > 0
        self.l = len(x)
        fn = singa.EltwiseMax if self.mode == "max" else singa.EltwiseMin
        if len(x) == 1:
            self.masks = None
            return singa.AddFloat(x[0], 0.0)
        out = x[0]
        for t in x[1:]:
            out = fn(out, t)
        if training:
            # gradient goes to the first input that attains the extremum
            self.shapes = [list(t.shape()) for t in x]
            self.out_shape = list(out.shape())
            taken = None
            self.masks = []
            for t in x:
                m = singa.__eq__(out, t)
                if taken is None:
                    taken = m
                else:
                    free = singa.AddFloat(singa.MultFloat(taken, -1.0), 1.0)
                    m = singa.__mul__(m, free)
                    taken = singa.__add__(taken, m)
                self.masks.append(m)
        return out

    def backward(self, dy):
        if self.l == 1:
            return dy
        return tuple(
            back_broadcast(self.out_shape, s, singa.__mul__(m, dy))
            for m, s in zip(self.masks, self.shapes))


class Min(_MinMax):
    mode = "min"


class Max(_MinMax):
    mode = "max"


def min(*l):
    return Min()(*l)[0]


def max(*l):
    return Max()(*l)[0]


# ---------------------------------------------------------------------------
# unary element-wise
# ---------------------------------------------------------------------------
class ReLU(Operator):

    def forward(self, x):
        if training:
            self.input = x
        return singa.ReLU(x)

    def backward(self, dy):
        return singa.ReLUBackward(dy, self.input)



class Clip(Operator):
    """Clamp to [min, max]; either bound may be None."""

    def __init__(self, min, max):
        super(Clip, self).__init__()
        self.max = max
        self.min = min

    def forward(self, x):
        lo = -3.4e38 if self.min is None else float(self.min)
        hi = 3.4e38 if self.max is None else float(self.max)
        y = singa.Clamp(x, lo, hi)
        if training:
            inside = singa.__mul__(singa.GEFloat(x, lo), singa.LEFloat(x, hi))
            self.mask = inside
        return y

    def backward(self, dy):
        return singa.__mul__(dy, self.mask)


def clip(x, min=None, max=None):
    return Clip(min, max)(x)[0]


class Identity(Operator):

    def forward(self, x):
        return x

    def backward(self, dy):
        return dy



class PRelu(Operator):
    """y = x if x >= 0 else slope * x (slope broadcastable to x)."""

    def forward(self, x, slope):
        mask0 = singa.LTFloat(x, 0.0)
        if training:
            self.input = x
            self.slope = slope
            self.mask0 = mask0
            self.shape0 = list(x.shape())
            self.shape1 = list(slope.shape())
        neg = singa.__mul__(singa.__mul__(x, mask0), slope)
        pos = singa.__mul__(x, singa.GEFloat(x, 0.0))
        self.shape3 = list(neg.shape())
        return singa.__add__(pos, neg)

    def backward(self, dy):
        dx1mask = singa.GEFloat(self.input, 0.0)
        dx2 = singa.__mul__(self.mask0, self.slope)
        dx = singa.__mul__(dy, singa.__add__(dx1mask, dx2))
        ds = singa.__mul__(dy, singa.__mul__(self.mask0, self.input))
        dx = back_broadcast(self.shape3, self.shape0, dx)
        ds = back_broadcast(self.shape3, self.shape1, ds)
        return dx, ds



class Elu(Operator):
    """f(x) = alpha * (exp(x) - 1) for x < 0, x otherwise."""

    def __init__(self, alpha=1.):
        super(Elu, self).__init__()
        self.alpha = alpha

    def forward(self, x):
        if training:
            self.input = x
        neg = singa.MultFloat(singa.AddFloat(singa.Exp(x), -1.0), self.alpha)
        neg = singa.__mul__(neg, singa.LTFloat(x, 0.0))
        pos = singa.ReLU(x)
        return singa.__add__(pos, neg)

    def backward(self, dy):
        dx1 = singa.GTFloat(self.input, 0.0)
        dx2 = singa.__mul__(singa.MultFloat(singa.Exp(self.input), self.alpha),
                            singa.LTFloat(self.input, 0.0))
        return singa.__mul__(dy, singa.__add__(dx1, dx2))


def elu(x, alpha=1):
    return Elu(alpha)(x)[0]


class SeLU(Operator):
    """y = gamma * (alpha * e^x - alpha) for x <= 0, gamma * x for x > 0."""

    def __init__(self, alpha=1.67326, gamma=1.0507):
        super(SeLU, self).__init__()
        self.alpha = alpha
        self.gamma = gamma

    def forward(self, x):
        if training:
            self.input = x
        pos = singa.MultFloat(singa.ReLU(x), self.gamma)
        neg = singa.MultFloat(singa.AddFloat(singa.Exp(x), -1.0),
                              self.alpha * self.gamma)
        neg = singa.__mul__(neg, singa.LEFloat(x, 0.0))
        return singa.__add__(pos, neg)

    def backward(self, dy):
        dx1 = singa.MultFloat(singa.GTFloat(self.input, 0.0), self.gamma)
        dx2 = singa.__mul__(
            singa.MultFloat(singa.Exp(self.input), self.gamma * self.alpha),
            singa.LEFloat(self.input, 0.0))
        return singa.__mul__(dy, singa.__add__(dx1, dx2))


def selu(x, alpha=1.67326, gamma=1.0507):
    return SeLU(alpha, gamma)(x)[0]


class LeakyRelu(Operator):
    """f(x) = a * x for x < 0, x otherwise."""

    def __init__(self, a):
        super(LeakyRelu, self).__init__()
        self.a = a

    def forward(self, x):
        if training:
            self.input = x
        neg = singa.MultFloat(singa.__mul__(x, singa.LEFloat(x, 0.0)), self.a)
        return singa.__add__(singa.ReLU(x), neg)

    def backward(self, dy):
        dx1 = singa.GTFloat(self.input, 0.0)
        dx2 = singa.MultFloat(singa.LEFloat(self.input, 0.0), self.a)
        return singa.__mul__(dy, singa.__add__(dx1, dx2))


def leakyrelu(x, a=0.01):
    return LeakyRelu(a)(x)[0]


class HardSigmoid(Operator):
    """y = max(0, min(1, alpha * x + gamma))."""

    def __init__(self, alpha=0.2, gamma=0.5):
        super(HardSigmoid, self).__init__()
        self.alpha = alpha
        self.gamma = gamma

    def forward(self, x):
        lin = singa.AddFloat(singa.MultFloat(x, self.alpha), self.gamma)
        if training:
            self.cache = lin
        return singa.Clamp(lin, 0.0, 1.0)

    def backward(self, dy):
        mask = singa.__mul__(singa.GTFloat(self.cache, 0.0),
                             singa.LTFloat(self.cache, 1.0))
        return singa.__mul__(singa.MultFloat(mask, self.alpha), dy)


def hardsigmoid(x, alpha=0.2, gamma=0.5):
    return HardSigmoid(alpha, gamma)(x)[0]


def _simple_unary(name, fwd, bwd, doc):
    """Factory for unary ops whose backward is dy * g(x, y)."""

    def forward(self, x):
        y = fwd(x)
        if training:
            self.input = x
            self.output = y
        return y

    def backward(self, dy):
        return singa.__mul__(dy, bwd(self.input, self.output))

    cls = type(name, (Operator,), {"forward": forward, "backward": backward,
                                   "__doc__": doc})
    return cls


def _one_minus_sq(t):
    return singa.AddFloat(singa.MultFloat(singa.Square(t), -1.0), 1.0)


Tanh = _simple_unary("Tanh", singa.Tanh, lambda x, y: _one_minus_sq(y),
                     "y = tanh(x)")
Sigmoid = _simple_unary(
    "Sigmoid", singa.Sigmoid,
    lambda x, y: singa.__mul__(y, singa.AddFloat(singa.MultFloat(y, -1.0), 1.0)),
    "y = 1 / (1 + exp(-x))")
Cos = _simple_unary("Cos", singa.Cos,
                    lambda x, y: singa.MultFloat(singa.Sin(x), -1.0), "cos")
Sin = _simple_unary("Sin", singa.Sin, lambda x, y: singa.Cos(x), "sin")
Tan = _simple_unary("Tan", singa.Tan,
                    lambda x, y: singa.AddFloat(singa.Square(y), 1.0), "tan")
Cosh = _simple_unary("Cosh", singa.Cosh, lambda x, y: singa.Sinh(x), "cosh")
Sinh = _simple_unary("Sinh", singa.Sinh, lambda x, y: singa.Cosh(x), "sinh")
Acos = _simple_unary(
    "Acos", singa.Acos,
    lambda x, y: singa.MultFloat(singa.PowFloat(_one_minus_sq(x), -0.5), -1.0),
    "acos")
Asin = _simple_unary(
    "Asin", singa.Asin, lambda x, y: singa.PowFloat(_one_minus_sq(x), -0.5),
    "asin")
Atan = _simple_unary(
    "Atan", singa.Atan,
    lambda x, y: singa.PowFloat(singa.AddFloat(singa.Square(x), 1.0), -1.0),
    "atan")
Acosh = _simple_unary(
    "Acosh", singa.Acosh,
    lambda x, y: singa.PowFloat(singa.AddFloat(singa.Square(x), -1.0), -0.5),
    "acosh")
Asinh = _simple_unary(
    "Asinh", singa.Asinh,
    lambda x, y: singa.PowFloat(singa.AddFloat(singa.Square(x), 1.0), -0.5),
    "asinh")
Atanh = _simple_unary(
    "Atanh", singa.Atanh, lambda x, y: singa.PowFloat(_one_minus_sq(x), -1.0),
    "atanh")
Abs = _simple_unary("Abs", singa.Abs, lambda x, y: singa.Sign(x), "abs")
Exp = _simple_unary("Exp", singa.Exp, lambda x, y: y, "exp")
Log = _simple_unary("Log", singa.Log, lambda x, y: singa.PowFloat(x, -1.0),
                    "natural log")
Sqrt = _simple_unary(
    "Sqrt", singa.Sqrt,
    lambda x, y: singa.MultFloat(singa.PowFloat(x, -0.5), 0.5), "sqrt")
SoftPlus = _simple_unary("SoftPlus", singa.SoftPlus,
                         lambda x, y: singa.Sigmoid(x), "log(1 + exp(x))")
SoftSign = _simple_unary(
    "SoftSign", singa.SoftSign,
    lambda x, y: singa.PowFloat(singa.AddFloat(singa.Abs(x), 1.0), -2.0),
    "x / (1 + |x|)")
Reciprocal = _simple_unary(
    "Reciprocal", lambda x: singa.PowFloat(x, -1.0),
    lambda x, y: singa.MultFloat(singa.Square(y), -1.0), "1 / x")
Erf = _simple_unary(
    "Erf", singa.Erf,
    lambda x, y: singa.MultFloat(
        singa.Exp(singa.MultFloat(singa.Square(x), -1.0)),
        2.0 / float(np.sqrt(np.pi))), "error function")























class _ZeroGradUnary(Operator):
    """Piece-wise constant functions: the gradient is zero everywhere."""
    fn = None

    def forward(self, x):
        return type(self).fn(x)

    def backward(self, dy):
        return singa.MultFloat(dy, 0.0)


class Sign(_ZeroGradUnary):
    fn = staticmethod(singa.Sign)


class Ceil(_ZeroGradUnary):
    fn = staticmethod(singa.Ceil)


class Floor(_ZeroGradUnary):
    fn = staticmethod(singa.Floor)


class Round(_ZeroGradUnary):
    """Round half away from zero."""
    fn = staticmethod(singa.Round)


class Rounde(_ZeroGradUnary):
    """Round half to even."""
    fn = staticmethod(singa.RoundE)







class Negative(Operator):

    def forward(self, x):
        return singa.MultFloat(x, -1.0)

    def backward(self, dy):
        return singa.MultFloat(dy, -1.0)



# ---------------------------------------------------------------------------
# binary arithmetic with broadcasting
# ---------------------------------------------------------------------------
class _BroadcastBinary(Operator):

    def _remember(self, a, b, out):
        self.shape0 = list(a.shape())
        self.shape1 = list(b.shape())
        self.shape3 = list(out.shape())


class Add(_BroadcastBinary):

    def forward(self, a, b):
        res = singa.__add__(a, b)
        self._remember(a, b, res)
        return res

    def backward(self, dy):
        dx0 = back_broadcast(self.shape3, self.shape0, dy)
        dx1 = back_broadcast(self.shape3, self.shape1, dy)
        return dx0, dx1



class Sub(_BroadcastBinary):

    def forward(self, a, b):
        res = singa.__sub__(a, b)
        self._remember(a, b, res)
        return res

    def backward(self, dy):
        dx0 = back_broadcast(self.shape3, self.shape0, dy)
        dx1 = back_broadcast(self.shape3, self.shape1,
                             singa.MultFloat(dy, -1.0))
        return dx0, dx1



class Mul(_BroadcastBinary):

    def forward(self, a, b):
        if training:
            self.input = (a, b)
        res = singa.__mul__(a, b)
        self._remember(a, b, res)
        return res

    def backward(self, dy):
        dx0 = back_broadcast(self.shape3, self.shape0,
                             singa.__mul__(dy, self.input[1]))
        dx1 = back_broadcast(self.shape3, self.shape1,
                             singa.__mul__(dy, self.input[0]))
        return dx0, dx1



class Div(_BroadcastBinary):

    def forward(self, a, b):
        res = singa.__mul__(a, singa.PowFloat(b, -1.0))
        if training:
            self.input = (singa.MultFloat(a, -1.0), singa.PowFloat(b, -1.0))
        self._remember(a, b, res)
        return res

    def backward(self, dy):
        # d/da = 1 / b ; d/db = -a / b^2
        dx0 = singa.__mul__(dy, self.input[1])
        dx1 = singa.__mul__(
            self.input[0], singa.PowFloat(self.input[1], 2.0))
        dx1 = singa.__mul__(dy, dx1)
        dx0 = back_broadcast(self.shape3, self.shape0, dx0)
        dx1 = back_broadcast(self.shape3, self.shape1, dx1)
        return dx0, dx1



class Pow(_BroadcastBinary):

    def forward(self, a, b):
        res = singa.Pow(a, b)
        if training:
            self.input = (a, b)
            self.res = res
        self._remember(a, b, res)
        return res

    def backward(self, dy):
        a, b = self.input
        da1 = singa.__mul__(b, singa.Pow(a, singa.AddFloat(b, -1.0)))
        dx0 = singa.__mul__(da1, dy)
        db1 = singa.__mul__(self.res, singa.Log(a))
        dx1 = singa.__mul__(db1, dy)
        dx0 = back_broadcast(self.shape3, self.shape0, dx0)
        dx1 = back_broadcast(self.shape3, self.shape1, dx1)
        return dx0, dx1



# ---------------------------------------------------------------------------
# comparisons and logic (no gradient)
# ---------------------------------------------------------------------------
class _NoGrad(Operator):
    """Operators without a derivative: their outputs stop the gradient, so
    chains that only transform data (an attention mask built from the input
    mask, index arithmetic) are never visited by ``backward``."""

    def _do_forward(self, *xs):
        ys = super(_NoGrad, self)._do_forward(*xs)
        self.requires_grad = False
        for y in ys:
            y.requires_grad = False
        return ys

    def backward(self, *dys):
        assert False, ('no backward function for %s' %
                       type(self).__name__.lower())


class Less(_NoGrad):

    def forward(self, x, y):
        return singa.__lt__(x, y)


class Greater(_NoGrad):

    def forward(self, x, y):
        return singa.__gt__(x, y)


class Equal(_NoGrad):

    def forward(self, x, y):
        return singa.__eq__(x, y)


class And(_NoGrad):

    def forward(self, a, b):
        m = singa.__mul__(a, b)
        return singa.Sign(singa.Abs(m)) if m.data_type() != singa.kInt else m


class Or(_NoGrad):

    def forward(self, a, b):
        m = singa.__add__(singa.Abs(singa.Sign(a)), singa.Abs(singa.Sign(b)))
        return singa.Sign(m)


class Not(_NoGrad):

    def forward(self, x):
        return singa.AddFloat(singa.MultFloat(singa.Abs(singa.Sign(x)), -1.0),
                              1.0)


class Xor(_NoGrad):

    def forward(self, a, b):
        m = singa.__sub__(singa.Abs(singa.Sign(a)), singa.Abs(singa.Sign(b)))
        return singa.Abs(m)









class Where(Operator):
    """out = condition ? a : b (all broadcastable)."""

    def __init__(self, condition):
        super(Where, self).__init__()
        self.condition = condition

    def forward(self, a, b):
        if isinstance(self.condition, list):
            self.condition = np.array(self.condition)
        if isinstance(self.condition, np.ndarray):
            self.condition = tensor.from_numpy(
                self.condition.astype(np.float32))
            self.condition.to_device(a.device())
        if isinstance(self.condition, Tensor):
            self.condition = self.condition.data
        cond = self.condition.AsType(a.data_type())
        self.cond = cond
        self.neg = singa.AddFloat(singa.MultFloat(cond, -1.0), 1.0)
        res = singa.__add__(singa.__mul__(cond, a), singa.__mul__(self.neg, b))
        self.shape0 = list(a.shape())
        self.shape1 = list(b.shape())
        self.shape3 = list(res.shape())
        return res

    def backward(self, dy):
        da = back_broadcast(self.shape3, self.shape0,
                            singa.__mul__(self.cond, dy))
        db = back_broadcast(self.shape3, self.shape1,
                            singa.__mul__(self.neg, dy))
        return da, db


def where(x, y, condition):
    return Where(condition)(x, y)[0]


# ---------------------------------------------------------------------------
# linear algebra
# ---------------------------------------------------------------------------
def _t_last2(t):
    nd = t.nDim()
    if nd <= 2:
        return singa.DefaultTranspose(t)
    axes = list(range(nd))
    axes[-1], axes[-2] = axes[-2], axes[-1]
    return singa.Transpose(t, axes)


class Matmul(Operator):
    """np.matmul for 2-d operands and for batched 3-d / 4-d operands."""

    def forward(self, x, w):
        if training:
            self.input = (x, w)
        return singa.Mult(x, w)

    def backward(self, dy):
        x, w = self.input
        dx = dw = None
        if self.x_needs_grad[0]:
            dx = singa.Mult(dy, _t_last2(w))
            if list(dx.shape()) != list(x.shape()):
                dx = back_broadcast(dx.shape(), x.shape(), dx)
        if self.x_needs_grad[1]:
            dw = singa.Mult(_t_last2(x), dy)
            if list(dw.shape()) != list(w.shape()):
                dw = back_broadcast(dw.shape(), w.shape(), dw)
        return dx, dw



class LinearBias(Operator):
    """y = x @ w + b for x of any rank (..., K), w (K, N), b (N,): the bias is
    added by the product kernel, its gradient is one column sum that reads
    ``dy`` in its own precision.  The importer maps MatMul(x, W) -> Add(b)
    with constant W and b to this operator (the reference keeps the two nodes,
    python/singa/sonnx.py)."""

    def forward(self, x, w, b):
        if training:
            self.input = (x, w)
        return singa.LinearBias(x, w, b)

    def backward(self, dy):
        x, w = self.input
        K, N = w.shape()
        M = int(np.prod(x.shape())) // K
        dy2 = singa.Reshape(dy, [M, N])
        dx, dw, db = singa.LinearBackward(
            dy2, singa.Reshape(x, [M, K]), w, bool(self.x_needs_grad[0]),
            bool(self.x_needs_grad[1]))
        if dx is not None:
            dx = singa.Reshape(dx, list(x.shape()))
        return dx, dw, db


def linear_bias(x, w, b):
    return LinearBias()(x, w, b)[0]


class AddBias(Operator):
    """Add a vector to every row (axis 0) or column (axis 1) of a matrix, in
    place."""

    def __init__(self, axis=0):
        super(AddBias, self).__init__()
        self.axis = axis

    def forward(self, x, b):
        if self.axis == 0:
            singa.AddRow(b, x)
        elif self.axis == 1:
            singa.AddColumn(b, x)
        return x

    def backward(self, dy):
        # the reductions accumulate in float32 whatever the type of dy
        return dy, singa.Sum(dy, 0 if self.axis == 0 else 1)


def add_bias(x, b, axis=0):
    assert x.ndim() == 2, "1st arg required 2d tensor. got shape: %s" % (
        x.shape,)
    assert b.ndim() == 1, "2nd arg required 1d tensor. got shape: %s" % (
        b.shape,)
    assert axis in [0, 1], "allowed axis: 0 or 1"
    return AddBias(axis)(x, b)[0]


def _scaled_product(scale, X, Y, into=None, keep=0.0):
    """``scale * X @ Y`` (+ ``keep * into``) as ONE product launch: the scale
    and the accumulation are part of the GEMM kernel's epilogue."""
    if into is None:
        into = singa.Tensor([X.shape()[0], Y.shape()[1]], X.device(),
                            X.data_type())
    singa.MultWithScale(scale, X, Y, keep, into)
    return into


class Gemm(Operator):
    """Y = alpha * op(A) @ op(B) + beta * C, ``op`` an optional transposition
    (ONNX Gemm).  alpha, beta and the addition of C ride on the product
    kernel; the transpositions are stride views."""

    def __init__(self, alpha=1.0, beta=1.0, transA=0, transB=0):
        super(Gemm, self).__init__()
        self.alpha, self.beta = float(alpha), float(beta)
        self.transA, self.transB = int(transA), int(transB)

    def _operands(self, A, B):
        return (singa.DefaultTranspose(A) if self.transA else A,
                singa.DefaultTranspose(B) if self.transB else B)

    def forward(self, A, B, C=None):
        left, right = self._operands(A, B)
        if training:
            self.inputs = (left, right, C)
        if C is None:
            return _scaled_product(self.alpha, left, right)
        rows, cols = left.shape()[0], right.shape()[1]
        # start from C broadcast to the result's shape, accumulate onto it
        y = singa.Contiguous(singa.Broadcast(C, [rows, cols]))
        if y is C or y.block() is C.block():
            y = y.Clone()
        return _scaled_product(self.alpha, left, right, into=y, keep=self.beta)

    def backward(self, dy):
        left, right, C = self.inputs
        # d op(A) = alpha * dy @ op(B)^T ; d op(B) = alpha * op(A)^T @ dy
        grads = [_scaled_product(self.alpha, dy, singa.DefaultTranspose(right)),
                 _scaled_product(self.alpha, singa.DefaultTranspose(left), dy)]
        for i, flipped in enumerate((self.transA, self.transB)):
            if flipped:
                grads[i] = singa.DefaultTranspose(grads[i])
        if C is not None:
            dc = dy if self.beta == 1.0 else singa.MultFloat(dy, self.beta)
            grads.append(back_broadcast(dy.shape(), C.shape(), dc))
        return tuple(grads)


def gemm(A, B, C=None, alpha=1.0, beta=1.0, transA=0, transB=0):
    if C is not None:
        return Gemm(alpha, beta, transA, transB)(A, B, C)[0]
    return Gemm(alpha, beta, transA, transB)(A, B)[0]


# ---------------------------------------------------------------------------
# soft-max and losses
# ---------------------------------------------------------------------------
class SoftMax(Operator):
    """Soft-max over the dimensions from ``axis`` on (flattened to 2-d)."""

    def __init__(self, axis=1):
        super(SoftMax, self).__init__()
        self.axis = axis

    def forward(self, x):
        self.output = singa.SoftMax(x, self.axis)
        return self.output

    def backward(self, dy):
        return singa.SoftMaxBackward(dy, self.axis, self.output)


def softmax(x, axis=1):
    return SoftMax(axis)(x)[0]


class LayerNorm(Operator):
    """y = (x - mean) / sqrt(var + eps) * scale + bias over the last
    dimension, one kernel per direction.  The reference has no such operator:
    models imported from ONNX run the exported chain of nine elementwise /
    reduction nodes (python/singa/sonnx.py); the importer here maps that chain
    (and the ONNX LayerNormalization node) onto this operator."""

    def __init__(self, eps=1e-5):
        super(LayerNorm, self).__init__()
        self.eps = float(eps)

    def forward(self, x, scale=None, bias=None):
        y, mean, rstd = singa.LayerNormForward(x, scale, bias, self.eps)
        if training:
            self.cache = (x, scale, mean, rstd, bias is not None)
        return y

    def backward(self, dy):
        x, scale, mean, rstd, has_bias = self.cache
        dx, ds, db = singa.LayerNormBackward(dy, x, scale, mean, rstd,
                                             scale is not None)
        if scale is None:
            return dx
        return (dx, ds, db) if has_bias else (dx, ds)


def layer_norm(x, scale=None, bias=None, eps=1e-5):
    args = [x] + [t for t in (scale, bias) if t is not None]
    assert bias is None or scale is not None, "a bias needs a scale"
    return LayerNorm(eps)(*args)[0]


class Gelu(Operator):
    """x * Phi(x) with the exact (erf) normal CDF, one kernel per direction
    (exported by transformer models as Div, Erf, Add, Mul, Mul)."""

    def forward(self, x):
        if training:
            self.input = x
        return singa.GeluForward(x)

    def backward(self, dy):
        return singa.GeluBackward(dy, self.input)



class Attention(Operator):
    """softmax(scale * q @ kt + mask) @ v with the heads merged:

        q (B, H, Sq, D), kt (B, H, D, Sk), v (B, H, Sk, D), mask broadcastable
        to (B, H, Sq, Sk)  ->  (B, Sq, H * D)

    One fused kernel per direction: scores and probabilities stay on chip,
    the transposed views of (B, S, H, D) activations that exporters write in
    front of the two MatMuls are read in place, and the output already has
    the layout of the merge-heads Transpose + Reshape that follows them.  The
    reference runs MatMul, Div, Add, Softmax, MatMul, Transpose, Reshape as
    separate nodes (examples/onnx/bert/bert-squad.py:102-165)."""

    def __init__(self, scale=1.0):
        super(Attention, self).__init__()
        self.scale = float(scale)

    def forward(self, q, kt, v, mask=None):
        assert q.nDim() == 4 and kt.nDim() == 4 and v.nDim() == 4, \
            "attention operands are (batch, heads, rows, columns)"
        out, lse = singa.AttentionForward(q, kt, v, mask, self.scale)
        B, Sq, H, D = out.shape()
        if training:
            self.cache = (q, kt, v, mask, out, lse)
        return singa.Reshape(out, [B, Sq, H * D])

    def backward(self, dy):
        q, kt, v, mask, out, lse = self.cache
        dq, dkt, dv = singa.AttentionBackward(
            singa.Reshape(dy, list(out.shape())), q, kt, v, mask, out, lse,
            self.scale)
        if mask is None:
            return dq, dkt, dv
        dmask = None
        if self.x_needs_grad[3]:       # an additive mask is data, not a weight
            dmask = _scalar_like(mask, 0.0)
        return dq, dkt, dv, dmask


def attention(q, kt, v, mask=None, scale=1.0):
    args = [q, kt, v] + ([mask] if mask is not None else [])
    return Attention(scale)(*args)[0]


class BinaryCrossEntropy(Operator):
    """-mean_batch sum(t log(x + eps) + (1 - t) log(1 - x + eps))."""
    takes_scalar_dy = True

    def __init__(self, t):
        super(BinaryCrossEntropy, self).__init__()
        self.t = t.data

    def forward(self, x):
        posx = singa.AddFloat(x, 0.0001)
        loss = singa.SumAll(singa.__mul__(self.t, singa.Log(posx)))
        negt = singa.AddFloat(singa.MultFloat(self.t, -1.0), 1.0)
        negx = singa.AddFloat(singa.MultFloat(x, -1.0), 1.0001)
        negLoss = singa.SumAll(singa.__mul__(negt, singa.Log(negx)))
        loss = singa.__add__(loss, negLoss)
        loss = singa.MultFloat(loss, -1.0 / x.shape()[0])
        self.x = posx
        return loss

    def backward(self, dy=1.0):
        dx = singa.__div__(self.t, self.x)
        negt = singa.AddFloat(self.t, -1.0)
        # d/dx log(1.0001 - x) = 1 / (x - 1.0001), and self.x is x + 0.0001
        # (the reference subtracts 0.9999, python/singa/autograd.py:1197,
        # which is off by 3e-4 in the denominator)
        negx = singa.AddFloat(self.x, -1.0002)
        dx = singa.__sub__(dx, singa.__div__(negt, negx))
        dx = singa.MultFloat(dx, float(-1.0 / self.x.shape()[0]))
        return _apply_dy(dx, dy)


def binary_cross_entropy(x, t):
    return BinaryCrossEntropy(t)(x)[0]


class CrossEntropy(Operator):
    """-mean_batch sum(t * log(x)) for probabilities x and dense targets t."""
    takes_scalar_dy = True

    def __init__(self, t):
        super(CrossEntropy, self).__init__()
        self.t = t.data

    def forward(self, x):
        loss = singa.SumAll(singa.__mul__(self.t, singa.Log(x)))
        loss = singa.MultFloat(loss, -1.0 / x.shape()[0])
        self.x = x
        return loss

    def backward(self, dy=1.0):
        dx = singa.__div__(self.t, self.x)
        dx = singa.MultFloat(dx, float(-1.0 / self.x.shape()[0]))
        return _apply_dy(dx, dy)


def cross_entropy(x, t):
    assert x.ndim() == 2, "1st arg required 2d tensor. got shape: " + str(
        x.shape)
    assert t.ndim() <= 2, "2nd arg required <=2d tensor. got shape: " + str(
        t.shape)
    return CrossEntropy(t)(x)[0]


class RankingLoss(Operator):
    """mean(max(0, M - pos + neg))."""
    takes_scalar_dy = True

    def __init__(self, M=0.2):
        super().__init__()
        self.M = M

    def forward(self, pos, neg):
        val = singa.AddFloat(singa.__sub__(neg, pos), self.M)
        gt_zero = singa.GTFloat(val, 0.0)
        if training:
            self.inputs = (gt_zero,)
        loss = singa.SumAll(singa.__mul__(gt_zero, val))
        return singa.MultFloat(loss, 1.0 / pos.shape()[0])

    def backward(self, dy=1.0):
        assert training, "enable training mode to do backward"
        gt_zero = self.inputs[0]
        dpos = singa.MultFloat(gt_zero, -1.0 / gt_zero.Size())
        dneg = singa.MultFloat(gt_zero, 1.0 / gt_zero.Size())
        return _apply_dy(dpos, dy), _apply_dy(dneg, dy)


def ranking_loss(pos, neg, M=0.2):
    assert pos.shape == neg.shape, "input and target shape different: %s, %s" % (
        pos.shape, neg.shape)
    return RankingLoss(M)(pos, neg)[0]


class SoftMaxCrossEntropy(Operator):
    """Mean soft-max cross-entropy of logits ``x`` against integer labels or a
    dense target distribution.  One fused kernel forward (soft-max, log,
    label pick, batch mean) and one backward ((p - t) / batch)."""
    takes_scalar_dy = True

    def __init__(self, t):
        super(SoftMaxCrossEntropy, self).__init__()
        self.t = t.data

    def forward(self, x):
        loss, self.p = singa.SoftmaxCrossEntropyFused(x, self.t)
        return loss

    def backward(self, dy=1.0):
        return _apply_dy(singa.SoftmaxCrossEntropyFusedBwd(self.p, self.t), dy)


def softmax_cross_entropy(x, t):
    assert x.ndim() == 2, "1st arg required 2d tensor. got shape: " + str(
        x.shape)
    assert t.ndim() <= 2, "2nd arg required <=2d tensor. got shape: " + str(
        t.shape)
    return SoftMaxCrossEntropy(t)(x)[0]


class MeanSquareError(Operator):
    takes_scalar_dy = True

    def __init__(self, t):
        super(MeanSquareError, self).__init__()
        self.t = t.data

    def forward(self, x):
        self.err = singa.__sub__(x, self.t)
        self.n = 1
        for s in x.shape():
            self.n *= s
        loss = singa.SumAll(singa.Square(self.err))
        return singa.MultFloat(loss, 1.0 / self.n)

    def backward(self, dy=1.0):
        dx = singa.MultFloat(self.err, float(2 / self.n))
        return _apply_dy(dx, dy)


def mse_loss(x, t):
    assert x.shape == t.shape, "input and target shape different: %s, %s" % (
        x.shape, t.shape)
    return MeanSquareError(t)(x)[0]


def ctensor2numpy(x):
    """Copy a core tensor into a numpy array of the same shape."""
    return x.GetFloatValue(int(x.Size())).reshape(x.shape())


class CosSim(Operator):
    """Row-wise cosine similarity of two (batch, dim) matrices."""

    @classmethod
    def dot(cls, a, b):
        """Row-wise dot products of two (batch, dim) CTensors -> (batch,): one
        multiply and one row reduction (the reference slices and multiplies
        row by row, python/singa/autograd.py:4941-4960)."""
        return singa.ReduceND(singa.__mul__(a, b), [1], "sum")

    def forward(self, a, b):
        ad = self.dot(a, a)
        bd = self.dot(b, b)
        ap = singa.PowFloat(ad, 0.5)
        bp = singa.PowFloat(bd, 0.5)
        ret = singa.__div__(self.dot(a, b), singa.__mul__(ap, bp))
        if training:
            self.inputs = (a, b, ad, bd, ap, bp, ret)
        return ret

    def backward(self, dy):
        a, b, ad, bd, ap, bp, ret = self.inputs
        n = a.shape()[0]
        ab = singa.Reshape(singa.__mul__(ap, bp), [n, 1])
        ad_ = singa.Reshape(ad, [n, 1])
        bd_ = singa.Reshape(bd, [n, 1])
        ret_ = singa.Reshape(ret, [n, 1])
        dy_ = singa.Reshape(dy, [n, 1])
        da = singa.__sub__(singa.__div__(b, ab),
                           singa.__div__(singa.__mul__(ret_, a), ad_))
        db = singa.__sub__(singa.__div__(a, ab),
                           singa.__div__(singa.__mul__(ret_, b), bd_))
        return singa.__mul__(dy_, da), singa.__mul__(dy_, db)


def cossim(a, b):
    assert a.shape == b.shape, "shape not match for cossim"
    assert a.ndim() == 2, "shape should be in 2d for cossim"
    assert b.ndim() == 2, "shape should be in 2d for cossim"
    return CosSim()(a, b)[0]


# ---------------------------------------------------------------------------
# shape manipulation
# ---------------------------------------------------------------------------
class Reshape(Operator):
    """np.reshape; 0 keeps the input dimension, -1 is inferred."""

    def __init__(self, shape):
        super(Reshape, self).__init__()
        self.shape = list(shape) if shape is not None else None

    def forward(self, x):
        self._shape = x.shape()
        shape = list(self.shape)
        shape = [
            self._shape[i] if i < len(self._shape) and shape[i] == 0 else
            shape[i] for i in range(len(shape))
        ]
        hidden_shape = int(np.prod(self._shape) // np.abs(np.prod(shape)))
        self.cache = [int(s) if s != -1 else hidden_shape for s in shape]
        return singa.Reshape(x, self.cache)

    def backward(self, dy):
        return singa.Reshape(dy, self._shape)


def reshape(x, shape):
    return Reshape(shape)(x)[0]


class Flatten(Operator):
    """Flatten to 2-d at ``axis``."""

    def __init__(self, axis=1):
        super(Flatten, self).__init__()
        self.axis = axis

    def forward(self, x):
        self.shape = list(x.shape())
        shape, axis = self.shape, self.axis
        if axis < 0:
            axis += len(shape)
        new_shape = (1, int(np.prod(shape))) if axis == 0 else (
            int(np.prod(shape[0:axis])), int(np.prod(shape[axis:])))
        return singa.Reshape(x, new_shape)

    def backward(self, dy):
        return singa.Reshape(dy, self.shape)


def flatten(x, axis=1):
    return Flatten(axis)(x)[0]


class Unsqueeze(Operator):

    def __init__(self, axis):
        super(Unsqueeze, self).__init__()
        if type(axis) is int:
            axis = [axis]
        self.axis = list(axis)

    def forward(self, x):
        self.cache = x.shape()
        cur = list(self.cache)
        nd = len(cur) + len(self.axis)
        for i in sorted(a + nd if a < 0 else a for a in self.axis):
            cur.insert(i, 1)
        return singa.Reshape(x, cur)

    def backward(self, dy):
        return singa.Reshape(dy, self.cache)


def unsqueeze(x, axis=-1):
    return Unsqueeze(axis)(x)[0]


class Squeeze(Operator):

    def __init__(self, axis=[]):
        super(Squeeze, self).__init__()
        self.axis = axis

    def forward(self, x):
        self.cache = x.shape()
        newshape = []
        if self.axis == [] or self.axis is None:
            newshape = [i for i in self.cache if i != 1]
        else:
            axes = self.axis if isinstance(self.axis, (list, tuple)) \
                else [self.axis]
            axes = [a + len(self.cache) if a < 0 else a for a in axes]
            for i, s in enumerate(self.cache):
                if i in axes:
                    assert s == 1, "the length of axis {} is {}, which should be 1".format(i, s)
                else:
                    newshape.append(s)
        if len(newshape) == 0:
            newshape = [1]
        return singa.Reshape(x, newshape)

    def backward(self, dy):
        return singa.Reshape(dy, self.cache)


def squeeze(x, axis=[]):
    return Squeeze(axis)(x)[0]


class Transpose(Operator):
    """Permute the axes; ``perm=None`` reverses them (the ONNX default)."""

    def __init__(self, perm=None):
        super(Transpose, self).__init__()
        self.perm = None if perm is None else list(perm)

    def forward(self, x):
        if self.perm is None:
            self.perm = list(range(len(x.shape()) - 1, -1, -1))
        return singa.Transpose(x, self.perm)

    def backward(self, dy):
        cur = [0] * len(self.perm)
        for i, p in enumerate(self.perm):
            cur[p] = i
        return singa.Transpose(dy, cur)


def transpose(x, shape):
    return Transpose(shape)(x)[0]


class Concat(Operator):

    def __init__(self, axis=0):
        super(Concat, self).__init__()
        self.axis = axis

    def forward(self, *xs):
        if self.axis < 0:
            self.axis = self.axis % len(xs[0].shape())
        offset = 0
        self.slice_point = []
        for t in xs:
            offset += t.shape()[self.axis]
            self.slice_point.append(offset)
        return singa.ConcatOn(singa.VecTensor(list(xs)), self.axis)

    def backward(self, dy):
        assert self.slice_point[-1] == dy.shape()[self.axis], "Shape mismatch."
        ends = self.slice_point
        sizes = [e - b for b, e in zip([0] + ends[:-1], ends)]
        return tuple(singa.SplitOn(dy, sizes, self.axis))


def cat(xs, axis=0):
    return Concat(axis)(*xs)[0]


class Slice(Operator):
    """numpy-style strided slice along ``axes`` (one kernel each way)."""

    def __init__(self, starts, ends, axes=None, steps=None):
        super(Slice, self).__init__()
        self.starts = starts
        self.ends = ends
        self.axes = axes
        self.steps = steps

    def forward(self, x):
        x_shape = list(x.shape())
        nd = len(x_shape)
        axes = list(range(len(self.starts))) if self.axes is None else [
            int(a) + nd if a < 0 else int(a) for a in self.axes]
        steps = [1] * len(axes) if self.steps is None else list(self.steps)
        b, e, s = [0] * nd, list(x_shape), [1] * nd
        flips = []
        for idx, axis in enumerate(axes):
            rng = range(x_shape[axis])[int(self.starts[idx]):int(self.ends[idx]):int(steps[idx])]
            assert len(rng) > 0, "Cannot support empty tensor"
            if rng.step > 0:
                b[axis], e[axis], s[axis] = rng.start, rng.stop, rng.step
            else:
                # walk the same elements forwards, flip afterwards
                last = rng[-1]
                b[axis], e[axis], s[axis] = last, rng.start + 1, -rng.step
                flips.append(axis)
        self.cache = (x_shape, b, e, s, flips)
        y = singa.SliceND(x, b, e, s)
        for axis in flips:
            y = _flip(y, axis)
        return y

    def backward(self, dy):
        x_shape, b, e, s, flips = self.cache
        for axis in flips:
            dy = _flip(dy, axis)
        return singa.EmbedND(dy, x_shape, b, e, s, 0.0)


def _flip(t, axis):
    n = t.shape()[axis]
    idx = tensor.from_numpy(np.arange(n - 1, -1, -1, dtype=np.int32))
    idx.to_device(t.device())
    return singa.IndexSelect(t, axis, idx.data)


def slice(x, starts, ends, axes=None, steps=None):
    return Slice(starts, ends, axes, steps)(x)[0]


class Split(Operator):
    """Cut ``x`` along ``axis`` into pieces of the sizes ``parts`` (or into
    ``num_output`` equal pieces) with one strided-copy launch per piece."""

    def __init__(self, axis, parts, num_output=None):
        super(Split, self).__init__()
        if parts is None and num_output is None:
            raise ValueError("Split needs the piece sizes or their number")
        self.axis, self.parts, self.num_output = axis, parts, num_output

    def forward(self, x):
        shape = list(x.shape())
        self.axis %= len(shape)
        if self.parts is None:
            each, rest = divmod(shape[self.axis], self.num_output)
            assert rest == 0, "Split: %d is not divisible into %d pieces" % (
                shape[self.axis], self.num_output)
            self.parts = [each] * self.num_output
        self._like = (shape, x.device(), x.data_type())
        return tuple(singa.SplitOn(x, self.parts, self.axis))

    def backward(self, *dys):
        # a piece nothing consumed has no gradient: it contributes zeros
        shape, dev, dtype = self._like

        def zeros(extent):
            z = singa.Tensor(shape[:self.axis] + [extent] +
                             shape[self.axis + 1:], dev, dtype)
            z.SetFloatValue(0.0)
            return z

        pieces = [dy if dy is not None else zeros(extent)
                  for dy, extent in zip(dys, self.parts)]
        return singa.ConcatOn(singa.VecTensor(pieces), self.axis)


def split(x, axis, parts, num_output=None):
    return Split(axis, parts, num_output)(x)


class Gather(Operator):
    """ONNX Gather; ``indices`` is a host-side list / array or an int32
    tensor on the device."""

    def __init__(self, axis, indices):
        super(Gather, self).__init__()
        self.axis = axis
        self.indices = indices

    def forward(self, x):
        self.x_shape = list(x.shape())
        self.axis = self.axis % len(self.x_shape)
        n = self.x_shape[self.axis]
        if isinstance(self.indices, Tensor) and \
                self.indices.dtype == tensor.int32:
            # indices computed on the device (token ids fed to the model):
            # read in place at every run, also when a recorded step is
            # replayed on new inputs; the kernels wrap negative entries
            self.idx_shape = list(self.indices.shape)
            self.index = singa.Reshape(self.indices.data,
                                       [int(np.prod(self.idx_shape))])
        else:
            if isinstance(self.indices, Tensor):
                idx = tensor.to_numpy(self.indices)
            else:
                idx = np.asarray(self.indices)
            self.idx_shape = list(idx.shape)
            flat = np.mod(idx.reshape(-1).astype(np.int64),
                          n).astype(np.int32)
            it = tensor.from_numpy(flat)
            it.to_device(x.device())
            self.index = it.data
        y = singa.IndexSelect(x, self.axis, self.index)
        out_shape = (self.x_shape[:self.axis] + self.idx_shape +
                     self.x_shape[self.axis + 1:])
        self.flat_shape = list(y.shape())
        return singa.Reshape(y, out_shape) if out_shape else y

    def backward(self, dy):
        dy = singa.Reshape(dy, self.flat_shape)
        return singa.IndexAdd(dy, self.axis, self.index, self.x_shape)


def gather(x, axis, indices):
    return Gather(axis, indices)(x)[0]


class Tile(Operator):
    """numpy.tile."""

    def __init__(self, repeats):
        super(Tile, self).__init__()
        self.repeats = [repeats] if isinstance(repeats, int) else repeats

    def forward(self, x):
        if isinstance(self.repeats, np.ndarray):
            self.repeats = self.repeats.tolist()
        self.x_shape = list(x.shape())
        reps = [int(r) for r in self.repeats]
        nd = builtins_max(len(reps), len(self.x_shape))
        reps = [1] * (nd - len(reps)) + reps
        shp = [1] * (nd - len(self.x_shape)) + self.x_shape
        self.cache = (shp, reps)
        return singa.TileND(singa.Reshape(x, shp), reps)

    def backward(self, dy):
        shp, reps = self.cache
        dx = singa.UntileND(dy, shp, reps)
        return singa.Reshape(dx, self.x_shape)


def tile(x, repeats):
    return Tile(repeats)(x)[0]


class Expand(Operator):
    """ONNX Expand (numpy broadcast to ``shape``)."""

    def __init__(self, shape):
        super(Expand, self).__init__()
        self.shape = shape

    def forward(self, x):
        if isinstance(self.shape, np.ndarray):
            self.shape = self.shape.tolist()
        else:
            self.shape = [int(s) for s in self.shape]
        self.x_shape = list(x.shape())
        out_shape = list(np.broadcast_shapes(tuple(self.x_shape),
                                             tuple(self.shape)))
        self.out_shape = out_shape
        return singa.BroadcastTo(x, out_shape)

    def backward(self, dy):
        return back_broadcast(self.out_shape, self.x_shape, dy)


def expand(x, shape):
    return Expand(shape)(x)[0]


class Pad(Operator):
    """ONNX Pad: constant, reflect or edge."""

    def __init__(self, mode, pads, constant=0.):
        super(Pad, self).__init__()
        self.mode = mode
        if self.mode not in ("constant", "reflect", "edge"):
            assert False, ('Only support three modes: constant, reflect, edge')
        # an omitted constant_value input means zero
        self.constant = 0. if constant is None else constant
        self.pads = pads
        self.pad_width = ()

    def forward(self, x):
        if not self.pad_width:
            half = len(self.pads) // 2
            self.pad_width = tuple(
                (int(self.pads[i]), int(self.pads[i + half]))
                for i in range(half))
        shape = list(x.shape())
        self.x_shape = shape
        starts = [p[0] for p in self.pad_width]
        out_shape = [s + p[0] + p[1] for s, p in zip(shape, self.pad_width)]
        ends = [a + s for a, s in zip(starts, shape)]
        self.region = (starts, ends)
        if self.mode == "constant":
            return singa.EmbedND(x, out_shape, starts, ends, None,
                                 self.constant)
        # reflect / edge: gather along each padded axis with an index map
        y = x
        self.maps = []
        for axis, (lo, hi) in enumerate(self.pad_width):
            if lo == 0 and hi == 0:
                continue
            n = shape[axis]
            pos = np.arange(-lo, n + hi)
            if self.mode == "edge":
                src = np.clip(pos, 0, n - 1)
            else:
                period = builtins_max(2 * (n - 1), 1)
                m = np.mod(pos, period)
                src = np.where(m >= n, period - m, m)
            it = tensor.from_numpy(src.astype(np.int32))
            it.to_device(x.device())
            self.maps.append((axis, it.data, list(y.shape())))
            y = singa.IndexSelect(y, axis, it.data)
        return y

    def backward(self, dy):
        if self.mode == "constant":
            starts, ends = self.region
            return singa.SliceND(dy, starts, ends)
        for axis, idx, shp in reversed(self.maps):
            dy = singa.IndexAdd(dy, axis, idx, shp)
        return dy


def pad(x, mode, pads, constant=0.):
    return Pad(mode, pads, constant)(x)[0]


class UpSample(Operator):
    """Nearest-neighbour up-sampling by an integer factor per dimension
    (ONNX Upsample): y[..., i * s + j, ...] = x[..., i, ...]."""

    def __init__(self, mode, scales):
        super(UpSample, self).__init__()
        if str(mode).lower() != "nearest":
            raise ValueError("UpSample: only mode 'nearest' is implemented, "
                             "got %r" % (mode,))
        self.mode = "nearest"
        self.scales = scales

    def _factors(self, rank):
        f = [int(v) for v in np.asarray(self.scales).reshape(-1)]
        assert len(f) == rank, "UpSample: one scale per dimension"
        self.scales = f
        return f

    def forward(self, x):
        self.x_shape = list(x.shape())
        for axis, f in enumerate(self._factors(len(self.x_shape))):
            if f > 1:
                x = x.Repeat([f], axis)
        return x

    def backward(self, dy):
        # view dy as (n0, s0, n1, s1, ...) and sum the s axes in one reduction
        split = [v for pair in zip(self.x_shape, self.scales) for v in pair]
        summed = singa.ReduceND(singa.Reshape(dy, split),
                                list(range(1, len(split), 2)), "sum")
        return singa.Reshape(summed, self.x_shape)


def upsample(x, mode, scales):
    return UpSample(mode, scales)(x)[0]


class DepthToSpace(Operator):
    """Rearrange depth blocks into space (DCR or CRD order)."""

    def __init__(self, blocksize, mode="DCR"):
        super(DepthToSpace, self).__init__()
        self.blocksize = blocksize
        self.mode = mode.upper()

    def forward(self, x):
        b, c, h, w = x.shape()
        bs = self.blocksize
        if self.mode == "DCR":
            t = singa.Reshape(x, [b, bs, bs, c // (bs * bs), h, w])
            t = singa.Transpose(t, [0, 3, 4, 1, 5, 2])
            self.perm = [0, 3, 4, 1, 5, 2]
        elif self.mode == "CRD":
            t = singa.Reshape(x, [b, c // (bs * bs), bs, bs, h, w])
            t = singa.Transpose(t, [0, 1, 4, 2, 5, 3])
            self.perm = [0, 1, 4, 2, 5, 3]
        else:
            assert False, "only support two methods: DCR and CRD."
        self.mid_shape = list(t.shape())
        self.x_shape = [b, c, h, w]
        return singa.Reshape(t, [b, c // (bs * bs), h * bs, w * bs])

    def backward(self, dy):
        t = singa.Reshape(dy, self.mid_shape)
        inv = [0] * 6
        for i, p in enumerate(self.perm):
            inv[p] = i
        t = singa.Transpose(t, inv)
        return singa.Reshape(t, self.x_shape)


def depth_to_space(x, blocksize, mode="DCR"):
    return DepthToSpace(blocksize, mode)(x)[0]


class SpaceToDepth(Operator):
    """Inverse of DepthToSpace (DCR)."""

    def __init__(self, blocksize, mode="DCR"):
        super(SpaceToDepth, self).__init__()
        self.blocksize = blocksize
        self.mode = mode.upper()

    def forward(self, x):
        b, c, h, w = x.shape()
        bs = self.blocksize
        t = singa.Reshape(x, [b, c, h // bs, bs, w // bs, bs])
        if self.mode == "DCR":
            self.perm = [0, 3, 5, 1, 2, 4]
        elif self.mode == "CRD":
            self.perm = [0, 1, 3, 5, 2, 4]
        else:
            assert False, "only support two methods: DCR and CRD."
        t = singa.Transpose(t, self.perm)
        self.mid_shape = list(t.shape())
        self.x_shape = [b, c, h, w]
        return singa.Reshape(t, [b, c * bs * bs, h // bs, w // bs])

    def backward(self, dy):
        t = singa.Reshape(dy, self.mid_shape)
        inv = [0] * 6
        for i, p in enumerate(self.perm):
            inv[p] = i
        t = singa.Transpose(t, inv)
        return singa.Reshape(t, self.x_shape)


def space_to_depth(x, blocksize, mode="DCR"):
    return SpaceToDepth(blocksize, mode)(x)[0]


class Shape(_NoGrad):
    """The shape of the input as a 1-d tensor."""

    def forward(self, x):
        cur = list(x.shape())
        t = tensor.from_numpy(np.array(cur, dtype=np.int32))
        t.to_device(x.device())
        return t.data



class ConstantOfShape(_NoGrad):
    """A tensor of the shape given by the (host-readable) input, filled with
    ``value``."""

    def __init__(self, value=0.):
        super(ConstantOfShape, self).__init__()
        self.value = value

    def forward(self, x):
        x_shape = tensor.to_numpy(tensor.from_raw_tensor(x)).astype(
            np.int64).tolist()
        assert np.min(x_shape) >= 0, ('shape cannot be negative')
        out = CTensor(x_shape, x.device())
        out.SetFloatValue(self.value)
        return out


def constant_of_shape(x, value=0):
    return ConstantOfShape(value)(x)[0]


class Cast(_NoGrad):

    def __init__(self, to):
        super(Cast, self).__init__()
        self.to = to

    def forward(self, x):
        if x.data_type() != self.to:
            x = x.AsType(self.to)
        return x


def cast(x, to):
    return Cast(to)(x)[0]


class OneHot(_NoGrad):
    """ONNX OneHot; ``values`` = [off_value, on_value]."""

    def __init__(self, axis, depth, values):
        super(OneHot, self).__init__()
        self.axis = axis
        self.depth = depth
        self.values = values

    def forward(self, indices):
        shape = list(indices.shape())
        rank = len(shape)
        axis = self.axis + rank + 1 if self.axis < 0 else self.axis
        depth = int(np.asarray(self.depth).reshape(-1)[0])
        # compare a broadcast arange against the (wrapped) indices on device
        rng = tensor.from_numpy(np.arange(depth, dtype=np.float32))
        rng.to_device(indices.device())
        rshape = [1] * axis + [depth] + [1] * (rank - axis)
        ishape = shape[:axis] + [1] + shape[axis:]
        idx = indices.AsType(singa.kFloat32)
        # python-style modulo keeps negative indices in range
        idx = singa.__sub__(idx, singa.MultFloat(
            singa.Floor(singa.MultFloat(idx, 1.0 / depth)), float(depth)))
        hit = singa.__eq__(singa.Reshape(rng.data, rshape),
                           singa.Reshape(idx, ishape))
        off, on = float(self.values[0]), float(self.values[1])
        return singa.AddFloat(singa.MultFloat(hit, on - off), off)


def onehot(axis, indices, depth, values):
    return OneHot(axis, depth, values)(indices)[0]


class NonZero(_NoGrad):
    """Indices of the non-zero elements, shape (rank, count).  The output
    shape depends on the data, so this op reads the tensor on the host."""

    def forward(self, x):
        y = ctensor2numpy(x)
        y = np.array(np.nonzero(y)).astype(np.int32)
        t = tensor.from_numpy(y)
        t.to_device(x.device())
        return t.data



class ScatterElements(Operator):
    """ONNX ScatterElements for any rank: ``y = x`` except
    ``y[.., indices[p], ..] = updates[p]`` along ``axis``.  The index tensor
    is host data (an ONNX initializer); it is turned once into flat offsets,
    of which the last write to a cell wins, and the data itself never leaves
    the device: one scatter kernel places the updates, one builds the mask of
    untouched cells, which is also the gradient of ``x``.  (The reference's
    python/singa/autograd.py:5555-5641 copies x to the host and loops over a
    2-d index.)"""

    def __init__(self, indices, updates, axis=0):
        super(ScatterElements, self).__init__()
        self.indices = indices
        self.updates = updates
        self.axis = axis

    def _offsets(self, x_shape):
        """(flat offsets into x, positions in the flattened updates)."""
        idx = self.indices
        if isinstance(idx, Tensor):
            idx = tensor.to_numpy(idx)
        idx = np.asarray(idx).astype(np.int64)
        rank = len(x_shape)
        assert idx.ndim == rank, (
            "indices have rank %d, the input rank %d" % (idx.ndim, rank))
        assert -rank <= self.axis < rank, "axis %d out of range" % self.axis
        axis = self.axis % rank
        extent = x_shape[axis]
        assert ((idx >= -extent) & (idx < extent)).all(), (
            "indices must lie in [%d, %d]" % (-extent, extent - 1))
        assert all(i <= d for i, d in zip(idx.shape, x_shape)), \
            "indices reach outside the input"
        coords = list(np.indices(idx.shape))
        coords[axis] = np.where(idx < 0, idx + extent, idx)
        flat = np.ravel_multi_index(coords, x_shape).reshape(-1)
        # keep the last occurrence of every target cell
        _, first_rev = np.unique(flat[::-1], return_index=True)
        keep = np.sort(flat.size - 1 - first_rev)
        return flat[keep].astype(np.int32), keep.astype(np.int32)

    def forward(self, x):
        shape = list(x.shape())
        n = int(np.prod(shape))
        dev = x.device()
        flat, which = self._offsets(shape)
        upd = self.updates
        if isinstance(upd, Tensor):
            chosen = singa.IndexSelect(
                singa.Reshape(upd.data, [upd.size()]), 0,
                _device_ints(which, dev))
        else:
            chosen = tensor.from_numpy(
                np.asarray(upd, np.float32).reshape(-1)[which]).data
            chosen.ToDevice(dev)
        if chosen.data_type() != x.data_type():
            chosen = chosen.AsType(x.data_type())
        k = int(flat.size)
        where = _device_ints(flat, dev)
        placed = singa.ScatterAddRows(singa.Reshape(chosen, [k, 1]), where,
                                      [n, 1])
        ones = CTensor([k, 1], dev, x.data_type())
        ones.SetFloatValue(1.0)
        hit = singa.ScatterAddRows(ones, where, [n, 1])
        untouched = singa.Reshape(
            singa.AddFloat(singa.MultFloat(hit, -1.0), 1.0), shape)
        if training:
            self.untouched = untouched
        return singa.__add__(singa.__mul__(x, untouched),
                             singa.Reshape(placed, shape))

    def backward(self, dy):
        return singa.__mul__(dy, self.untouched)


def _device_ints(values, dev):
    t = tensor.from_numpy(np.ascontiguousarray(values, dtype=np.int32))
    t.to_device(dev)
    return t.data


def scatter_elements(x, indices, updates, axis=0):
    return ScatterElements(indices, updates, axis)(x)[0]


class Embedding(Operator):
    """Row lookup: out[..., :] = w[x[...], :] with device-side indices; the
    weight gradient is one scatter-add kernel."""

    def forward(self, x, w):
        self.x_shape = list(x.shape())
        self.w_shape = list(w.shape())
        idx = x if x.data_type() == singa.kInt else x.AsType(singa.kInt)
        if training:
            self.idx = idx
        return singa.GatherRows(w, idx, self.x_shape + [self.w_shape[1]])

    def backward(self, dy):
        dw = singa.ScatterAddRows(dy, self.idx, self.w_shape)
        # the index input has no gradient
        return None, dw



# ---------------------------------------------------------------------------
# reductions
# ---------------------------------------------------------------------------
class _Reduce(Operator):
    mode = "sum"

    def __init__(self, axes=None, keepdims=1):
        super(_Reduce, self).__init__()
        self.axes = axes
        self.keepdims = keepdims

    def forward(self, x):
        x_shape = list(x.shape())
        nd = len(x_shape)
        if self.axes is None:
            self.axes = list(range(nd))
        else:
            self.axes = [int(i) if i >= 0 else nd + int(i) for i in self.axes]
        kept = [1 if i in self.axes else s for i, s in enumerate(x_shape)]
        self.cache = (x_shape, kept)
        self.count = 1
        for i in self.axes:
            self.count *= x_shape[i]
        y = singa.ReduceND(x, self.axes, self.mode, keepdims=True)
        if self.keepdims != 1:
            out = [s for i, s in enumerate(x_shape) if i not in self.axes]
            y = singa.Reshape(y, out if out else [1])
        return y

    def backward(self, dy):
        x_shape, kept = self.cache
        dy = singa.Reshape(dy, kept)
        if self.mode == "mean":
            dy = singa.MultFloat(dy, 1.0 / self.count)
        return singa.BroadcastTo(dy, x_shape)


class ReduceSum(_Reduce):
    mode = "sum"


class ReduceMean(_Reduce):
    mode = "mean"


def reduce_sum(x, axes=None, keepdims=1):
    return ReduceSum(axes, keepdims)(x)[0]


def reduce_mean(x, axes=None, keepdims=1):
    return ReduceMean(axes, keepdims)(x)[0]


class GlobalAveragePool(Operator):

    def __init__(self, data_format='channels_first'):
        super(GlobalAveragePool, self).__init__()
        self.data_format = data_format

    def forward(self, x):
        shape = list(x.shape())
        if self.data_format == 'channels_first':
            axes = list(range(2, len(shape)))
        else:
            axes = list(range(1, len(shape) - 1))
        self.x_shape = shape
        self.count = 1
        for i in axes:
            self.count *= shape[i]
        return singa.ReduceND(x, axes, "mean", keepdims=True)

    def backward(self, dy):
        return singa.BroadcastTo(singa.MultFloat(dy, 1.0 / self.count),
                                 self.x_shape)


def globalaveragepool(x, data_format='channels_first'):
    return GlobalAveragePool(data_format)(x)[0]


# ---------------------------------------------------------------------------
# NN ops backed by native handles
# ---------------------------------------------------------------------------
class _Conv2d(Operator):

    def __init__(self, handle, odd_padding=(0, 0, 0, 0)):
        super(_Conv2d, self).__init__()
        self.handle = handle
        self.odd_padding = tuple(odd_padding)
        # the backward-data kernel adds a gradient that already reached x
        # along another path (residual forks) in its epilogue
        self.fuses_dx_addend = self.odd_padding == (0, 0, 0, 0)
        self.dx_addend = None

    def forward(self, x, W, b=None):
        assert x.nDim() == 4, "The dimensions of input should be 4D."
        if self.odd_padding != (0, 0, 0, 0):
            x = utils.handle_odd_pad_fwd(x, self.odd_padding)
        if training:
            self.inputs = (x, W, b) if self.handle.bias_term else (x, W)
        return singa.ConvForward(x, W, b if self.handle.bias_term else None,
                                 self.handle, emit_stats=bool(training))

    def backward(self, dy):
        assert training is True and hasattr(
            self, "inputs"), "Please set training as True before do BP. "
        dx = None
        if self.x_needs_grad[0]:
            dx, dW = singa.ConvBackwardxW(dy, self.inputs[1], self.inputs[0],
                                          self.handle, self.dx_addend)
            self.dx_addend = None
        else:
            dW = singa.ConvBackwardW(dy, self.inputs[0], self.inputs[1],
                                     self.handle)
        db = singa.ConvBackwardb(dy, self.inputs[2], self.handle) \
            if self.handle.bias_term else None
        if dx is not None and self.odd_padding != (0, 0, 0, 0):
            dx = utils.handle_odd_pad_bwd(dx, self.odd_padding)
        if db is not None:
            return dx, dW, db
        return dx, dW


def conv2d(handle, x, W, b=None, odd_padding=(0, 0, 0, 0)):
    if b is None:
        return _Conv2d(handle, odd_padding)(x, W)[0]
    return _Conv2d(handle, odd_padding)(x, W, b)[0]


class _BatchNorm2d(Operator):

    def __init__(self, handle, running_mean, running_var, name=None):
        super(_BatchNorm2d, self).__init__(name)
        self.handle = handle
        self.running_mean = running_mean.data
        self.running_var = running_var.data

    def forward(self, x, scale, bias):
        if training:
            y, mean, var = singa.BatchNormForwardTraining(
                self.handle, x, scale, bias, self.running_mean,
                self.running_var)
            self.cache = (x, scale, mean, var)
        else:
            y = singa.BatchNormForwardInference(self.handle, x, scale, bias,
                                                self.running_mean,
                                                self.running_var)
        return y

    def backward(self, dy):
        assert training is True and hasattr(
            self, "cache"), "Please set training as True before do BP. "
        x, scale, mean, var = self.cache
        dx, ds, db = singa.BatchNormBackward(self.handle, dy, x, scale, mean,
                                             var)
        return dx, ds, db


def batchnorm_2d(handle, x, scale, bias, running_mean, running_var):
    return _BatchNorm2d(handle, running_mean, running_var)(x, scale, bias)[0]


class _BatchNormAct2d(Operator):
    """Batch normalisation with the following ReLU -- and optionally the
    residual addition in front of it -- computed in the normalisation pass:
    y = relu(bn(x)) or y = relu(bn(x) + residual).  The activation costs no
    extra trip through HBM in either direction: backward recomputes the ReLU
    mask from x (or reads the saved y when a residual was added) inside the
    two BN backward passes.  The reference runs these as separate cuDNN /
    elementwise nodes (examples/cnn/model/resnet.py:71-101)."""

    def __init__(self, handle, running_mean, running_var, with_residual=False,
                 name=None):
        super(_BatchNormAct2d, self).__init__(name)
        self.handle = handle
        self.running_mean = running_mean.data
        self.running_var = running_var.data
        self.act = singa.BN_ACT_ADD_RELU if with_residual else \
            singa.BN_ACT_RELU

    def forward(self, x, scale, bias, residual=None):
        if training:
            y, mean, var = singa.BatchNormForwardTraining(
                self.handle, x, scale, bias, self.running_mean,
                self.running_var, self.act, residual)
            self.cache = (x, scale, bias, mean, var, y)
        else:
            y = singa.BatchNormForwardInference(
                self.handle, x, scale, bias, self.running_mean,
                self.running_var, self.act, residual)
        return y

    def backward(self, dy):
        assert training is True and hasattr(
            self, "cache"), "Please set training as True before do BP. "
        x, scale, bias, mean, var, y = self.cache
        if self.act == singa.BN_ACT_RELU:
            dx, ds, db = singa.BatchNormBackward(self.handle, dy, x, scale,
                                                 mean, var, self.act, bias)
            return dx, ds, db
        dx, ds, db, dres = singa.BatchNormBackward(
            self.handle, dy, x, scale, mean, var, self.act, None, y)
        return dx, ds, db, dres


def batchnorm_relu_2d(handle, x, scale, bias, running_mean, running_var):
    """relu(batchnorm(x)) as one operator."""
    return _BatchNormAct2d(handle, running_mean, running_var)(x, scale,
                                                              bias)[0]


def batchnorm_add_relu_2d(handle, x, scale, bias, running_mean, running_var,
                          residual):
    """relu(batchnorm(x) + residual) as one operator."""
    return _BatchNormAct2d(handle, running_mean, running_var, True)(
        x, scale, bias, residual)[0]


class _Pooling2d(Operator):

    def __init__(self, handle, odd_padding=(0, 0, 0, 0)):
        super(_Pooling2d, self).__init__()
        self.handle = handle
        self.odd_padding = tuple(odd_padding)

    def forward(self, x):
        assert x.nDim() == 4, "The dimensions of input should be 4D."
        if self.odd_padding != (0, 0, 0, 0):
            x = utils.handle_odd_pad_fwd(x, self.odd_padding, True,
                                         bool(self.handle.is_max_pooling))
        y = singa.PoolingForward(self.handle, x)
        if training:
            self.cache = (x, y)
        return y

    def backward(self, dy):
        dx = singa.PoolingBackward(self.handle, dy, self.cache[0],
                                   self.cache[1])
        if self.odd_padding != (0, 0, 0, 0):
            dx = utils.handle_odd_pad_bwd(dx, self.odd_padding, True,
                                          bool(self.handle.is_max_pooling))
        return dx


def pooling_2d(handle, x, odd_padding=(0, 0, 0, 0)):
    return _Pooling2d(handle, odd_padding)(x)[0]


class Dropout(Operator):
    """output = scale * data * mask with scale = 1 / (1 - ratio), as the
    reference documents it (python/singa/autograd.py:3939-3943; its code
    computes ``1 / 1 - ratio``, an operator-precedence slip not repeated
    here)."""

    def __init__(self, seed=0, ratio=0.5):
        super(Dropout, self).__init__()
        self.ratio = ratio
        self.seed = int(seed)
        self.init_seed = False

    def forward(self, x):
        if not self.init_seed:
            x.device().SetRandSeed(self.seed)
            self.init_seed = True
        if training:
            self.scale = 1.0 / (1.0 - self.ratio)
            self.mask = CTensor(list(x.shape()), x.device(), x.data_type())
            singa.Bernoulli(1 - self.ratio, self.mask)
            x = singa.MultFloat(singa.__mul__(self.mask, x), self.scale)
        return x

    def backward(self, dy):
        if training:
            dy = singa.MultFloat(singa.__mul__(self.mask, dy), self.scale)
        return dy


def dropout(x, seed=0, ratio=0.5):
    return Dropout(seed, ratio)(x)[0]


class _RNN(Operator):
    """Multi-layer (bi)directional RNN / LSTM / GRU over a flat weight blob,
    executed by the fused recurrent kernels of the handle."""

    def __init__(self, handle, return_sequences=False, use_mask=False,
                 seq_lengths=None):
        super(_RNN, self).__init__()
        self.handle = handle
        self.return_sequences = return_sequences
        self.use_mask = use_mask
        if use_mask:
            assert type(seq_lengths) == Tensor, "wrong type for seq_lengths"
        self.seq_lengths = seq_lengths

    def forward(self, x, hx, cx, w):
        lens = self.seq_lengths.data if self.use_mask else None
        if training:
            y, hy, cy = singa.RNNForwardTraining(x, hx, cx, w, self.handle,
                                                 lens)
            self.inputs = {'x': x, 'hx': hx, 'cx': cx, 'w': w, 'y': y,
                           'hy': hy, 'cy': cy}
        else:
            y, hy, cy = singa.RNNForwardInference(x, hx, cx, w, self.handle,
                                                  lens)
        if self.return_sequences:
            return y
        last_y = CTensor([y.shape()[1], y.shape()[2]], x.device())
        singa.CopyDataToFrom(last_y, y, last_y.Size(), 0,
                             y.Size() - last_y.Size())
        return last_y

    def backward(self, grad):
        assert training is True and hasattr(
            self, "inputs"), "Please set training as True before do BP. "
        y = self.inputs['y']
        if self.return_sequences:
            assert grad.shape() == y.shape(), (
                "grad shape %s != y shape %s" % (grad.shape(), y.shape()))
            dy = grad
        else:
            dy = CTensor(list(y.shape()), grad.device())
            dy.SetFloatValue(0.0)
            singa.CopyDataToFrom(dy, grad, grad.Size(),
                                 dy.Size() - grad.Size(), 0)
        dhy = CTensor(list(self.inputs['hy'].shape()), grad.device())
        dhy.SetFloatValue(0.0)
        dcy = CTensor(list(self.inputs['cy'].shape()), grad.device())
        dcy.SetFloatValue(0.0)
        lens = self.seq_lengths.data if self.use_mask else None
        dx, dhx, dcx, dW = singa.RNNBackward(
            y, dy, dhy, dcy, self.inputs['w'], self.inputs['hx'],
            self.inputs['cx'], self.inputs['x'], self.handle, lens)
        return dx, dhx, dcx, dW


import builtins as _builtins  # noqa: E402

builtins_max = _builtins.max


# ---------------------------------------------------------------------------
# functional forms of the operators that take no attributes:
# name -> (operator class, number of tensor inputs); each call records a
# fresh operator instance and returns its single output
# ---------------------------------------------------------------------------
def _functional(name, cls, arity):
    def fn(*xs):
        if len(xs) != arity:
            raise TypeError("%s() takes %d tensors, got %d" %
                            (name, arity, len(xs)))
        return cls()(*xs)[0]
    fn.__name__ = fn.__qualname__ = name
    fn.__doc__ = cls.__doc__
    return fn


for _name, (_cls, _arity) in {
    "relu": (ReLU, 1), "identity": (Identity, 1), "prelu": (PRelu, 2),
    "tanh": (Tanh, 1), "sigmoid": (Sigmoid, 1), "cos": (Cos, 1),
    "sin": (Sin, 1), "tan": (Tan, 1), "cosh": (Cosh, 1), "sinh": (Sinh, 1),
    "acos": (Acos, 1), "asin": (Asin, 1), "atan": (Atan, 1),
    "acosh": (Acosh, 1), "asinh": (Asinh, 1), "atanh": (Atanh, 1),
    "abs": (Abs, 1), "exp": (Exp, 1), "log": (Log, 1), "sqrt": (Sqrt, 1),
    "softplus": (SoftPlus, 1), "softsign": (SoftSign, 1),
    "reciprocal": (Reciprocal, 1), "erf": (Erf, 1), "sign": (Sign, 1),
    "ceil": (Ceil, 1), "floor": (Floor, 1), "round": (Round, 1),
    "rounde": (Rounde, 1), "negative": (Negative, 1), "add": (Add, 2),
    "sub": (Sub, 2), "mul": (Mul, 2), "div": (Div, 2), "pow": (Pow, 2),
    "less": (Less, 2), "greater": (Greater, 2), "equal": (Equal, 2),
    "_and": (And, 2), "_or": (Or, 2), "_not": (Not, 1), "_xor": (Xor, 2),
    "matmul": (Matmul, 2), "gelu": (Gelu, 1), "shape": (Shape, 1),
    "nonzero": (NonZero, 1), "embedding": (Embedding, 2),
}.items():
    globals()[_name] = _functional(_name, _cls, _arity)


# ---------------------------------------------------------------------------
# names kept for programs written against older releases (reference
# python/singa/autograd.py:5753-5770): the operator base class under its old
# name, and the layer classes that once lived in this module.  The layers are
# looked up on first use: layer.py imports this module.
# ---------------------------------------------------------------------------
Operation = Operator

_LAYER_ALIASES = ("Linear", "Conv2d", "SeparableConv2d", "BatchNorm2d",
                  "Pooling2d", "MaxPool2d", "AvgPool2d", "MaxPool1d",
                  "AvgPool1d", "RNN_Base", "RNN", "LSTM")


def __getattr__(name):
    if name in _LAYER_ALIASES:
        from . import layer
        return getattr(layer, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))

