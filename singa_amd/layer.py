"""Layers: stateful building blocks over autograd operators.

A layer creates its parameters lazily from the first input it sees
(``initialize``), always with graph recording switched off so that parameter
initialisation never becomes part of a recorded training step.

Parameters and persistent states are *registered* when they are created
(``Layer._new_param`` / ``_new_state``): the base class answers ``get_params``,
``set_params``, ``get_states`` and ``set_states`` for every layer from that
registry plus the registered sub-layers, so a layer class only says what it
owns and how it computes.  Tensors and sub-layers are named hierarchically
(``parent.child.W``) as they are assigned.

Layers that merely forward to one autograd function are generated from a
table at the end of the module.

API parity: reference python/singa/layer.py (LayerMeta :29-72, Layer :75-284
and the layer classes).  ``CudnnRNN`` keeps its reference name and is backed by
the fused HIP recurrent kernels.
"""
import math
import os
from collections import OrderedDict
from functools import wraps

from . import singa_wrap as singa
from . import tensor
from . import utils
from .tensor import Tensor


# ---------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------
class _eager(object):
    """Context: graph recording off on ``dev``, previous state restored."""

    def __init__(self, dev):
        self.dev = dev

    def __enter__(self):
        self.was = self.dev.graph_enabled()
        self.dev.EnableGraph(False)

    def __exit__(self, *exc):
        self.dev.EnableGraph(self.was)
        return False


def _lead_tensor(args):
    lead = args[0]
    while isinstance(lead, (list, tuple)):
        lead = lead[0]
    assert isinstance(lead, Tensor), (
        'a layer is initialised from PlaceHolders or Tensors, got %s' %
        type(lead).__name__)
    return lead


def _gaussian(std, mean=0.0):
    return lambda t: t.gaussian(mean, std)


def _constant(value):
    return lambda t: t.set_value(value)


def _keep(obj, values, *names):
    """obj.<name> = values[name] for every name (constructor boilerplate)."""
    for n in names:
        setattr(obj, n, values[n])


class LayerMeta(type):
    """Makes ``initialize`` eager and once-only bookkeeping automatic, and
    makes ``forward`` initialise the layer on its first call."""

    def __init__(cls, name, bases, namespace):
        super(LayerMeta, cls).__init__(name, bases, namespace)
        if 'initialize' in namespace:
            cls.initialize = LayerMeta._eager_initialize(
                namespace['initialize'])
        if 'forward' in namespace:
            cls.forward = LayerMeta._initialising_forward(namespace['forward'])

    @staticmethod
    def _eager_initialize(fn):

        @wraps(fn)
        def initialize(self, *args, **kwargs):
            if not args:
                return
            with _eager(_lead_tensor(args).device):
                fn(self, *args, **kwargs)
                self._initialized = True

        return initialize

    @staticmethod
    def _initialising_forward(fn):

        @wraps(fn)
        def forward(self, *args, **kwargs):
            if not self._initialized:
                self.initialize(*args, **kwargs)
                self._initialized = True
            return fn(self, *args, **kwargs)

        return forward


# ---------------------------------------------------------------------------
# base class
# ---------------------------------------------------------------------------
class Layer(object, metaclass=LayerMeta):

    sep = '.'

    def __init__(self):
        vars(self).update(name=None, _initialized=False, _parent=None,
                          _layers=dict(), _owned={'param': [], 'state': []})

    def initialize(self, *input):
        """Create parameters / states / handles that depend on the input."""

    def forward(self, *input):
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    # ---- registry --------------------------------------------------------------------
    def _register(self, kind, attr, shape, like, fill, dtype, trainable):
        t = Tensor(shape=shape, device=like.device,
                   dtype=tensor.float32 if dtype is None else dtype,
                   requires_grad=trainable, stores_grad=trainable)
        fill(t)
        setattr(self, attr, t)              # also names the tensor
        if attr not in self._owned[kind]:
            self._owned[kind].append(attr)
        return t

    def _new_param(self, attr, shape, like, fill, dtype=None):
        """Create the trainable tensor ``self.<attr>`` on ``like``'s device."""
        return self._register('param', attr, shape, like, fill, dtype, True)

    def _new_state(self, attr, shape, like, fill, dtype=None):
        """Create the persistent, non-trainable tensor ``self.<attr>``."""
        return self._register('state', attr, shape, like, fill, dtype, False)

    def _own(self, kind):
        return [getattr(self, a) for a in self._owned[kind]]

    def _ready_children(self):
        return [c for c in self._layers.values() if c._initialized]

    def get_params(self):
        params = {t.name: t for t in self._own('param')}
        for child in self._ready_children():
            params.update(child.get_params())
        return params

    def set_params(self, parameters):
        for t in self._own('param'):
            t.copy_from(parameters[t.name])
        for child in self._ready_children():
            if self._has_layer_param(child, parameters.keys()):
                child.set_params(parameters)

    def get_states(self):
        states = {}
        for child in self._ready_children():
            states.update(child.get_states())
        states.update(self.get_params())
        states.update((t.name, t) for t in self._own('state'))
        return states

    def set_states(self, states):
        for child in self._ready_children():
            if self._has_layer_param(child, states.keys()):
                child.set_states(states)
        for t in self._own('state'):
            t.copy_from(states[t.name])
        self.set_params(states)

    def set_attribute(self, attribute, attribute_value):
        assert attribute_value.shape == attribute.shape, (
            "cannot load a %s value into a %s tensor" %
            (attribute_value.shape, attribute.shape))
        attribute.copy_from(attribute_value)

    # ---- input hygiene (never recorded) ----------------------------------------------
    def dtype_check(self, *inputs):
        """Cast every later tensor to the dtype of the first one."""
        lead = inputs[0]
        with _eager(lead.device):
            for t in inputs[1:]:
                if t.dtype != lead.dtype:
                    t.to_type(lead.dtype)

    def device_check(self, *inputs):
        """Move every later tensor to the device of the first one."""
        lead = inputs[0]
        with _eager(lead.device):
            for t in inputs[1:]:
                if t.device.id() != lead.device.id():
                    t.to_device(lead.device)

    def _align(self, x, *tensors, dtype=True):
        """device_check (+ dtype_check) of the tensors that exist."""
        present = [t for t in tensors if t is not None]
        self.device_check(x, *present)
        if dtype:
            self.dtype_check(x, *present)

    # ---- hierarchy and naming ----------------------------------------------------------
    def _has_layer_param(self, layer, names):
        return any(n.startswith(layer.name) for n in names)

    def _get_name_prefix(self):
        return self.name + Layer.sep if self.name and self._parent else ''

    def _attach(self, key, child, name):
        self._layers[key] = child
        vars(child)['_parent'] = self
        child.name = name               # re-names the child's own subtree

    def __getattr__(self, name):
        # only reached when normal lookup fails: sub-layers live in _layers
        children = vars(self).get('_layers', {})
        try:
            return children[name]
        except KeyError:
            raise AttributeError("'{}' object has no attribute '{}'".format(
                type(self).__name__, name)) from None

    def __setattr__(self, name, value):
        if isinstance(value, Layer):
            self._attach(name, value, self._get_name_prefix() + name)
            return
        object.__setattr__(self, name, value)
        if isinstance(value, Tensor):
            if value.is_dummy():
                value.name = self._get_name_prefix() + name
        elif name == 'name' and value:
            for key, child in self._layers.items():
                child.name = self._get_name_prefix() + key

    def __delattr__(self, name):
        if name in self._layers:
            del self._layers[name]
        else:
            object.__delattr__(self, name)

    def register_layers(self, *args):
        """Register sub-layers that are not plain attributes (e.g. the blocks
        of a ResNet stage held in a list): either one OrderedDict name ->
        layer, or the layers themselves (named ``<Class>_<position>``)."""
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            named = list(args[0].items())
        else:
            named = [('%s_%d' % (type(v).__name__, i), v)
                     for i, v in enumerate(args)]
        for key, child in named:
            if isinstance(child, Layer):
                self._attach(key, child, key)


# ---------------------------------------------------------------------------
# dense layers
# ---------------------------------------------------------------------------
class Linear(Layer):
    """y = x W + b with W of shape (in_features, out_features); features are
    the last axis of a 2-d or batched (..., in_features) input."""

    def __init__(self, out_features, *args, bias=True, **kwargs):
        super().__init__()
        if args:    # legacy signature Linear(in_features, out_features[, bias])
            self.in_features, out_features = out_features, args[0]
        self.out_features = out_features
        self.bias = args[1] if len(args) > 1 else bias

    def initialize(self, x):
        self.in_features = fan_in = x.shape[-1]
        std = math.sqrt(2.0 / (fan_in + self.out_features))
        self._new_param('W', (fan_in, self.out_features), x, _gaussian(std),
                        x.dtype)
        self.b = None
        if self.bias:
            self._new_param('b', (self.out_features,), x, _constant(0.0),
                            x.dtype)

    def forward(self, x):
        self._align(x, self.W, self.b)
        assert x.shape[-1] == self.W.shape[0], (
            "Linear layer built for %d input features received %d" %
            (self.W.shape[0], x.shape[-1]))
        y = autograd.matmul(x, self.W)
        return autograd.add_bias(y, self.b, axis=0) if self.bias else y


class Gemm(Layer):
    """Y = alpha * A' * B' + beta * C with B the weight and C the bias."""

    def __init__(self, nb_kernels, alpha=1.0, beta=1.0, transA=False,
                 transB=True, bias=True, bias_shape=None):
        super().__init__()
        _keep(self, locals(), 'nb_kernels', 'alpha', 'beta', 'bias',
              'bias_shape')
        self.transA, self.transB = int(bool(transA)), int(bool(transB))

    def _in_features_of(self, x):
        return x.shape[0] if self.transA else x.shape[-1]

    def initialize(self, x):
        self.in_features = fan_in = self._in_features_of(x)
        w_shape = (self.nb_kernels, fan_in) if self.transB \
            else (fan_in, self.nb_kernels)
        self._new_param('W', w_shape, x,
                        _gaussian(math.sqrt(2.0 / (fan_in + self.nb_kernels))))
        self.b = None
        if self.bias:
            self._new_param('b', self.bias_shape or (1, self.nb_kernels), x,
                            _constant(0.0))

    def forward(self, x):
        self._align(x, self.W, self.b, dtype=False)
        expected = self.W.shape[-1] if self.transB else self.W.shape[0]
        assert self._in_features_of(x) == expected, (
            "Gemm layer built for %d input features received %d" %
            (expected, self._in_features_of(x)))
        return autograd.gemm(x, self.W, self.b, self.alpha, self.beta,
                             self.transA, self.transB)


class Embedding(Layer):
    """Table lookup: row ``x[i]`` of an (input_dim, output_dim) matrix."""

    def __init__(self, input_dim, output_dim, initializer="gaussian"):
        super().__init__()
        _keep(self, locals(), 'input_dim', 'output_dim', 'initializer')

    def initialize(self, x):
        fill = (lambda t: t.uniform(-1., 1.)) \
            if self.initializer == 'uniform' else _gaussian(1.)
        self._new_param('W', (self.input_dim, self.output_dim), x, fill)

    def from_pretrained(self, W, freeze=True):
        self.set_params({self.W.name: W})
        self.W.requires_grad = not freeze

    def forward(self, x):
        return autograd.embedding(x, self.W)


# ---------------------------------------------------------------------------
# sliding-window layers
# ---------------------------------------------------------------------------
def _pair(v, what):
    if isinstance(v, int):
        return (v, v)
    if isinstance(v, (tuple, list)):
        return tuple(v)
    raise TypeError("%s is an int or a pair, got %r" % (what, v))


def _split_padding(padding):
    """(symmetric (h, w) padding, odd remainder (top, bottom, left, right)).
    Accepts None, an int, (h, w) or (top, bottom, left, right)."""
    none = (0, 0, 0, 0)
    if padding is None:
        return (0, 0), none
    if isinstance(padding, int):
        return (padding, padding), none
    if isinstance(padding, (tuple, list)) and len(padding) == 2:
        return tuple(padding), none
    if isinstance(padding, (tuple, list)) and len(padding) == 4:
        top, bottom, left, right = padding
        odd = (max(top - bottom, 0), max(bottom - top, 0),
               max(left - right, 0), max(right - left, 0))
        return (top - odd[0], left - odd[2]), odd
    raise TypeError("padding is None, an int, a pair or a 4-tuple, got %r" %
                    (padding,))


class _Windowed(Layer):
    """Shared geometry of convolution and pooling: ``kernel_size``, ``stride``,
    symmetric ``padding`` plus ``odd_padding``, and ``pad_mode``."""

    def _set_window(self, kernel_size, stride, padding, pad_mode):
        self.kernel_size = _pair(kernel_size, "kernel_size")
        self.stride = self.kernel_size if stride is None \
            else _pair(stride, "stride")
        self.padding, self.odd_padding = _split_padding(padding)
        self.pad_mode = pad_mode

    def _padded_probe(self, x):
        """Resolve SAME_* padding against ``x`` and return a core tensor with
        the shape the kernels will see (odd padding applied)."""
        if self.pad_mode in ("SAME_UPPER", "SAME_LOWER"):
            pad, self.odd_padding = utils.get_padding_shape(
                self.pad_mode, x.shape[2:], self.kernel_size, self.stride)
            self.padding = [pad[0], pad[2]]
        top, bottom, left, right = self.odd_padding
        n, c, h, w = x.shape
        return singa.Tensor([n, c, h + top + bottom, w + left + right],
                            x.device, x.dtype)


class Conv2d(_Windowed):
    """2-d convolution (NCHW).  ``pad_mode`` SAME_UPPER / SAME_LOWER derives
    the padding from the input; ``activation="RELU"`` appends a ReLU."""

    def __init__(self, nb_kernels, kernel_size, *args, stride=1, padding=0,
                 dilation=1, group=1, bias=True, pad_mode="NOTSET",
                 activation="NOTSET", **kwargs):
        super().__init__()
        if args:
            # legacy Conv2d(in_channels, out_channels, kernel[, stride[, pad]])
            nb_kernels, kernel_size = kernel_size, args[0]
            stride = args[1] if len(args) > 1 else stride
            padding = args[2] if len(args) > 2 else padding
        if dilation != 1 and list(dilation) != [1, 1]:
            raise ValueError("dilated convolution is not implemented")
        self._set_window(kernel_size, stride, padding, pad_mode)
        _keep(self, locals(), 'nb_kernels', 'dilation', 'group', 'bias',
              'activation')
        # accepted for source compatibility; the HIP kernels pick their own
        # algorithm and workspace
        self.inner_params = {"cudnn_prefer": "fastest",
                             "workspace_MB_limit": 1024}
        unknown = set(kwargs) - set(self.inner_params)
        if unknown:
            raise TypeError("Keyword argument not understood:",
                            sorted(unknown)[0])
        self.inner_params.update(kwargs)

    def initialize(self, x):
        self.in_channels = x.shape[1]
        kh, kw = self.kernel_size
        per_group = self.in_channels // self.group
        std = math.sqrt(2.0 / (per_group * kh * kw + self.nb_kernels))
        self._new_param('W', (self.nb_kernels, per_group, kh, kw), x,
                        _gaussian(std))
        self.b = None
        if self.bias:
            self._new_param('b', (self.nb_kernels,), x, _constant(0.0))
        probe = self._padded_probe(x)
        if not hasattr(self, "handle"):
            self.handle = singa.ConvHandle(
                probe, self.kernel_size, self.stride, self.padding,
                self.in_channels, self.nb_kernels, self.bias, self.group)

    def forward(self, x):
        self._align(x, self.W, self.b)
        assert self.group >= 1 and self.in_channels % self.group == 0, (
            "%d input channels cannot be split into %d groups" %
            (self.in_channels, self.group))
        assert self.nb_kernels >= self.group and \
            self.nb_kernels % self.group == 0, (
                "%d kernels cannot be split into %d groups" %
                (self.nb_kernels, self.group))
        y = autograd.conv2d(self.handle, x, self.W, self.b, self.odd_padding)
        return autograd.relu(y) if self.activation == "RELU" else y


class SeparableConv2d(Layer):
    """Depth-wise convolution followed by a 1x1 point-wise convolution."""

    def __init__(self, nb_kernels, kernel_size, *args, stride=1, padding=0,
                 bias=False):
        super().__init__()
        if args:    # legacy (in_channels, out_channels, kernel[, stride[, pad]])
            nb_kernels, kernel_size = kernel_size, args[0]
            stride = args[1] if len(args) > 1 else stride
            padding = args[2] if len(args) > 2 else padding
        _keep(self, locals(), 'nb_kernels', 'kernel_size', 'stride', 'padding',
              'bias')

    def initialize(self, x):
        self.in_channels = x.shape[1]
        self.depthwise_conv = Conv2d(self.in_channels, self.kernel_size,
                                     stride=self.stride, padding=self.padding,
                                     group=self.in_channels, bias=self.bias)
        self.point_conv = Conv2d(self.nb_kernels, 1, bias=self.bias)

    def forward(self, x):
        return self.point_conv(self.depthwise_conv(x))


class BatchNorm2d(Layer):
    """Batch normalisation over (N, H, W) per channel.  ``momentum`` is the
    weight of the batch statistics in the running average (see
    ``singa_wrap.BatchNormHandle``).  Statistics and affine parameters stay
    in float32 for 16-bit inputs.

    ``activation="RELU"`` computes relu(bn(x)) -- and, when the layer is
    called with a second tensor, relu(bn(x) + residual) -- inside the
    normalisation kernel instead of as separate graph nodes."""

    def __init__(self, *args, momentum=0.9, activation=None):
        super().__init__()
        if args:                         # legacy (channels[, momentum])
            self.channels = args[0]
            momentum = args[1] if len(args) > 1 else momentum
        assert 0 <= momentum <= 1.0, "momentum %r outside [0, 1]" % momentum
        assert activation in (None, "RELU"), "only RELU can be fused"
        _keep(self, locals(), 'momentum', 'activation')

    def initialize(self, x, residual=None):
        self.channels = c = x.shape[1]
        self._new_param('scale', (c,), x, _constant(1.0))
        self._new_param('bias', (c,), x, _constant(0.0))
        self._new_state('running_mean', (c,), x, _constant(0.0))
        self._new_state('running_var', (c,), x, _constant(1.0))
        if not hasattr(self, "handle"):
            self.handle = singa.BatchNormHandle(self.momentum, x.data)

    def forward(self, x, residual=None):
        assert x.shape[1] == self.channels, (
            "built for %d channels, input has %d" % (self.channels,
                                                     x.shape[1]))
        own = (self.scale, self.bias, self.running_mean, self.running_var)
        self.device_check(x, *own)
        if residual is not None:
            assert self.activation == "RELU", \
                "a residual input needs activation='RELU'"
            return autograd.batchnorm_add_relu_2d(self.handle, x, *own,
                                                  residual)
        op = autograd.batchnorm_relu_2d if self.activation == "RELU" \
            else autograd.batchnorm_2d
        return op(self.handle, x, *own)


class Pooling2d(_Windowed):
    """Max / average pooling over (N, C, H, W); (N, C, L) inputs of the 1-d
    variants are pooled as (N, C, 1, L)."""

    def __init__(self, kernel_size, stride=None, padding=0, is_max=True,
                 pad_mode="NOTSET"):
        super().__init__()
        self._set_window(kernel_size, stride, padding, pad_mode)
        self.is_max = is_max

    @staticmethod
    def _as_4d(x):
        if len(x.shape) != 3:
            return x
        n, c, length = x.shape
        return autograd.reshape(x, (n, c, 1, length))

    def initialize(self, x):
        probe = self._padded_probe(self._as_4d(x))
        self.handle = singa.PoolingHandle(probe, self.kernel_size, self.stride,
                                          self.padding, self.is_max)

    def forward(self, x):
        y = autograd.pooling_2d(self.handle, self._as_4d(x), self.odd_padding)
        if len(x.shape) == 3:
            y = autograd.reshape(y, (y.shape[0], y.shape[1], y.shape[3]))
        return y


def _pooling_variant(name, is_max, one_d):
    def __init__(self, kernel_size, stride=None, padding=0, pad_mode="NOTSET"):
        if one_d:
            stride = kernel_size if stride is None else stride
            kernel_size, stride, padding = (1, kernel_size), (1, stride), \
                (0, padding)
        Pooling2d.__init__(self, kernel_size, stride, padding, is_max,
                           pad_mode)
    return LayerMeta(name, (Pooling2d,),
                     {'__init__': __init__, '__module__': __name__})


MaxPool2d = _pooling_variant('MaxPool2d', True, False)
AvgPool2d = _pooling_variant('AvgPool2d', False, False)
MaxPool1d = _pooling_variant('MaxPool1d', True, True)
AvgPool1d = _pooling_variant('AvgPool1d', False, True)


# ---------------------------------------------------------------------------
# recurrent layers
# ---------------------------------------------------------------------------
class RNN_Base(Layer):

    def step_forward(self, x=None, h=None, c=None, Wx=None, Wh=None, Bx=None,
                     Bh=None, b=None):
        raise NotImplementedError

    def _remember(self, values):
        _keep(self, values, 'input_size', 'hidden_size', 'num_layers',
              'nonlinearity', 'bias', 'batch_first', 'dropout',
              'bidirectional')


class RNN(RNN_Base):
    """Vanilla RNN unrolled over a list of per-step inputs."""

    def __init__(self, input_size, hidden_size, num_layers=1,
                 nonlinearity="tanh", bias=True, batch_first=False, dropout=0,
                 bidirectional=False):
        super().__init__()
        self._remember(locals())

    def initialize(self, xs, h0):
        x, n, d = xs[0], self.hidden_size, self.input_size
        self._new_param('Wx', (d, n), x, _gaussian(1.0))
        self._new_param('Wh', (n, n), x, _gaussian(1.0))
        self._new_param('b', (n,), x, _constant(0.0))

    def forward(self, xs, h0):
        xs = tuple(xs)
        self.device_check(*xs, h0)
        self.device_check(xs[0], self.Wx, self.Wh, self.b)
        h, out = h0, []
        for x in xs:
            assert x.shape[0] == xs[0].shape[0], "batch size changed mid-sequence"
            h = self.step_forward(x, h, self.Wx, self.Wh, self.b)
            out.append(h)
        return out, h

    def step_forward(self, x, h, Wx, Wh, b):
        try:
            act = {"tanh": autograd.tanh, "relu": autograd.relu}[
                self.nonlinearity]
        except KeyError:
            raise ValueError("nonlinearity is 'tanh' or 'relu', got %r" %
                             (self.nonlinearity,)) from None
        pre = autograd.add(autograd.matmul(h, Wh), autograd.matmul(x, Wx))
        return act(autograd.add_bias(pre, b, axis=0))


class LSTM(RNN_Base):
    """LSTM unrolled over a list of per-step inputs.  Parameters keep the
    reference layout (8 weight matrices + 8 bias vectors: Wx_/Wh_/Bx_/Bh_ for
    gates i, f, o, g).  ``forward`` runs the whole sequence as one operator;
    ``forward_unrolled`` / ``step_forward`` give the step-by-step forms."""

    _gates = ("i", "f", "o", "g")

    def __init__(self, input_size, hidden_size, nonlinearity="tanh",
                 num_layers=1, bias=True, batch_first=False, dropout=0,
                 bidirectional=False):
        super().__init__()
        self._remember(locals())

    def _family(self, prefix):
        return [getattr(self, prefix + g) for g in self._gates]

    def initialize(self, xs, h0_c0):
        x, n, d = xs[0], self.hidden_size, self.input_size
        for prefix, shape, fill in (("Wx_", (d, n), _gaussian(0.01)),
                                    ("Wh_", (n, n), _gaussian(0.01)),
                                    ("Bx_", (n,), _constant(0.0)),
                                    ("Bh_", (n,), _constant(0.0))):
            for g in self._gates:
                self._new_param(prefix + g, shape, x, fill)

    def forward(self, xs, h0_c0):
        h, c = h0_c0
        xs = list(xs)
        self.device_check(*xs, h, c)
        self.device_check(xs[0], *self._own('param'))
        for x in xs:
            assert x.shape[0] == xs[0].shape[0], "batch size changed mid-sequence"
        if os.environ.get("SINGA_LSTM_UNROLLED") == "1":
            return self.forward_unrolled(xs, (h, c))
        # the whole sequence is one operator (one GEMM for the input
        # projection of all steps, single-launch recurrence on the GPU)
        handles = vars(self).setdefault('_handles', {})
        out, c = rnn_cells.lstm_sequence(xs, h, c, self._own('param'), handles)
        return out, out[-1], c

    def forward_unrolled(self, xs, h0_c0):
        """The same computation as a chain of per-step operators."""
        h, c = h0_c0
        xs = list(xs)
        # gate-concatenated views of the parameters, built once per call
        Wx = autograd.cat(self._family("Wx_"), 1)
        Wh = autograd.cat(self._family("Wh_"), 1)
        B = autograd.add(autograd.cat(self._family("Bx_"), 0),
                         autograd.cat(self._family("Bh_"), 0))
        out = []
        for x in xs:
            h, c = self._fused_step(x, h, c, Wx, Wh, B)
            out.append(h)
        return out, h, c

    def _fused_step(self, x, h, c, Wx, Wh, B):
        gates = autograd.add(autograd.matmul(x, Wx), autograd.matmul(h, Wh))
        return rnn_cells.lstm_cell(autograd.add_bias(gates, B, axis=0), c)

    def step_forward(self, x, h, c):
        """One LSTM step from the un-concatenated parameters."""
        def pre(g):
            from_x = autograd.add_bias(
                autograd.matmul(x, getattr(self, "Wx_" + g)),
                getattr(self, "Bx_" + g), axis=0)
            from_h = autograd.add_bias(
                autograd.matmul(h, getattr(self, "Wh_" + g)),
                getattr(self, "Bh_" + g), axis=0)
            return autograd.add(from_x, from_h)
        i, f, o = (autograd.sigmoid(pre(g)) for g in "ifo")
        cand = autograd.tanh(pre("g"))
        c_next = autograd.add(autograd.mul(f, c), autograd.mul(i, cand))
        return autograd.mul(o, autograd.tanh(c_next)), c_next


class CudnnRNN(Layer):
    """Multi-layer (bi)directional vanilla / LSTM / GRU network with a single
    flat weight tensor, run by fused recurrent HIP kernels on the GPU and by
    the same algorithm on the host."""

    # mode codes of the recurrent kernels
    _modes = {"relu": 0, "vanilla": 1, "tanh": 1, "lstm": 2, "gru": 3}

    def __init__(self, hidden_size, activation="tanh", num_layers=1, bias=True,
                 batch_first=True, dropout=0, bidirectional=False,
                 rnn_mode="lstm", use_mask=False, return_sequences=True):
        assert num_layers > 0, "num_layers must be positive"
        assert 0 <= dropout < 1, "dropout must lie in [0, 1)"
        super().__init__()
        _keep(self, locals(), 'rnn_mode', 'hidden_size', 'num_layers',
              'dropout', 'return_sequences', 'batch_first', 'use_mask')
        self.bidirectional = int(bool(bidirectional))
        self.cudnn_rnn_mode = self._modes[rnn_mode]

    def initialize(self, x, hx=None, cx=None, seq_lengths=None):
        if self.batch_first:
            x = x.transpose((1, 0, 2))
        self.input_size = x.shape[1]
        self.handle = singa.CudnnRNNHandle(
            x.data, self.hidden_size, mode=self.cudnn_rnn_mode,
            num_layers=self.num_layers, dropout=self.dropout,
            bidirectional=self.bidirectional)
        bound = math.sqrt(1 / self.hidden_size)
        self._new_param('W', (self.handle.weights_size,), x,
                        lambda t: t.uniform(-bound, bound))

    def _zero_state(self, x):
        shape = (self.num_layers * (1 + self.bidirectional), x.shape[1],
                 self.hidden_size)
        return Tensor(shape=shape, requires_grad=False, stores_grad=False,
                      device=x.device).set_value(0.0)

    def forward(self, x, hx=None, cx=None, seq_lengths=None):
        self.device_check(x, self.W)
        if self.batch_first:
            x = autograd.transpose(x, (1, 0, 2))
        hx = self._zero_state(x) if hx is None else hx
        cx = self._zero_state(x) if cx is None else cx
        options = dict(return_sequences=self.return_sequences)
        if self.use_mask:
            assert type(seq_lengths) == Tensor, \
                "use_mask needs seq_lengths as a Tensor"
            options.update(use_mask=True, seq_lengths=seq_lengths)
        y = autograd._RNN(self.handle, **options)(x, hx, cx, self.W)[0]
        if self.return_sequences and self.batch_first:
            y = autograd.transpose(y, (1, 0, 2))
        return y

    def set_params(self, parameters):
        self.set_attribute(self.W, parameters[self.W.name])


# ---------------------------------------------------------------------------
# normalisation over the feature axis (an addition over the reference's set)
# ---------------------------------------------------------------------------
class LayerNorm(Layer):
    """Normalisation over the last dimension with a learned scale and bias;
    one fused kernel per direction."""

    def __init__(self, eps=1e-5):
        super().__init__()
        self.eps = eps

    def initialize(self, x):
        d = x.shape[-1]
        self._new_param('scale', (d,), x, _constant(1.0), x.dtype)
        self._new_param('bias', (d,), x, _constant(0.0), x.dtype)

    def forward(self, x):
        self.device_check(x, self.scale, self.bias)
        return autograd.layer_norm(x, self.scale, self.bias, self.eps)


# ---------------------------------------------------------------------------
# layers that only forward to an autograd function
# ---------------------------------------------------------------------------
def _functional(name, function, doc=None, **defaults):
    """A parameterless layer class calling ``autograd.<function>`` with its
    inputs followed by the values of its constructor options (``defaults``
    gives their names, order and default values)."""
    options = list(defaults)

    def __init__(self, *args, **kwargs):
        Layer.__init__(self)
        given = dict(zip(options, args))
        given.update(kwargs)
        unknown = set(given) - set(options)
        if unknown or len(args) > len(options):
            raise TypeError("%s() got unexpected arguments" % name)
        for key in options:
            setattr(self, key, given.get(key, defaults[key]))

    def forward(self, *inputs):
        extra = [getattr(self, key) for key in options]
        return getattr(autograd, function)(*inputs, *extra)

    return LayerMeta(name, (Layer,), {
        '__init__': __init__, 'forward': forward, '__doc__': doc,
        '__module__': __name__})


ReLU = _functional('ReLU', 'relu')
Gelu = _functional('Gelu', 'gelu')
Sigmoid = _functional('Sigmoid', 'sigmoid')
SoftMax = _functional('SoftMax', 'softmax')
Add = _functional('Add', 'add')
Reshape = _functional('Reshape', 'reshape', "forward(x, shape)")
Flatten = _functional('Flatten', 'flatten', axis=1)
Cat = _functional('Cat', 'cat', "forward(list of tensors)", axis=0)
SoftMaxCrossEntropy = _functional('SoftMaxCrossEntropy',
                                  'softmax_cross_entropy')
MeanSquareError = _functional('MeanSquareError', 'mse_loss')
CrossEntropy = _functional('CrossEntropy', 'cross_entropy')
BinaryCrossEntropy = _functional('BinaryCrossEntropy', 'binary_cross_entropy')
Dropout = _functional(
    'Dropout', 'dropout',
    """Like the reference, ``ratio`` is passed positionally into
    ``autograd.dropout(x, seed, ratio)`` and therefore lands in ``seed``; the
    effective drop ratio is the operator's default 0.5 (reference
    python/singa/layer.py:1522).""", ratio=0.5)


# imported last: autograd imports tensor which must not import layer
from . import autograd  # noqa: E402
from .ops import rnn_cells  # noqa: E402
