"""Layers: stateful building blocks over autograd operators.

A layer creates its parameters lazily from the first input it sees
(``initialize``), always with graph recording switched off so that parameter
initialisation never becomes part of a recorded training step.  Parameters are
named hierarchically (``parent.child.W``) through ``__setattr__``.

API parity: reference python/singa/layer.py (LayerMeta :29-72, Layer :75-284
and the layer classes).  ``CudnnRNN`` keeps its reference name and is backed by
the fused HIP recurrent kernels.
"""
import math
from collections import OrderedDict
from functools import wraps

from . import singa_wrap as singa
from . import tensor
from . import utils
from .tensor import Tensor


def _first_device(args):
    a = args[0]
    if isinstance(a, (list, tuple)):
        a = a[0]
    assert isinstance(a, Tensor), (
        'initialize function expects PlaceHolders or Tensors')
    return a.device


class LayerMeta(type):

    def init_wrapper(func):

        @wraps(func)
        def wrapper(self, *args, **kwargs):
            if len(args) == 0:
                return
            dev = _first_device(args)
            prev_state = dev.graph_enabled()
            dev.EnableGraph(False)
            func(self, *args, **kwargs)
            self._initialized = True
            dev.EnableGraph(prev_state)

        return wrapper

    def forward_wrapper(func):

        @wraps(func)
        def wrapper(self, *args, **kwargs):
            if not self._initialized:
                self.initialize(*args, **kwargs)
                self._initialized = True
            return func(self, *args, **kwargs)

        return wrapper

    def __new__(cls, name, bases, attr):
        if 'initialize' in attr:
            attr['initialize'] = LayerMeta.init_wrapper(attr['initialize'])
        if 'forward' in attr:
            attr['forward'] = LayerMeta.forward_wrapper(attr['forward'])
        return super(LayerMeta, cls).__new__(cls, name, bases, attr)


class Layer(object, metaclass=LayerMeta):

    sep = '.'

    def __init__(self):
        self.name = None
        self._initialized = False
        self._parent = None
        self._layers = dict()

    def initialize(self, *input):
        """Create parameters / states / handles that depend on the input."""
        pass

    def forward(self, *input):
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def get_params(self):
        params = dict()
        for name, sublayer in self._layers.items():
            if sublayer._initialized:
                params.update(sublayer.get_params())
        return params

    def set_params(self, parameters):
        names = parameters.keys()
        for name, sublayer in self._layers.items():
            if sublayer._initialized:
                if self._has_layer_param(sublayer, names):
                    sublayer.set_params(parameters)

    def get_states(self):
        states = dict()
        for name, sublayer in self._layers.items():
            if sublayer._initialized:
                states.update(sublayer.get_states())
        states.update(self.get_params())
        return states

    def set_states(self, states):
        names = states.keys()
        for name, sublayer in self._layers.items():
            if sublayer._initialized:
                if self._has_layer_param(sublayer, names):
                    sublayer.set_states(states)
        self.set_params(states)

    def dtype_check(self, *inputs):
        """Cast every input to the dtype of the first one (never recorded)."""
        dev = inputs[0].device
        flag = dev.graph_enabled()
        dev.EnableGraph(False)
        x_dtype = inputs[0].dtype
        for inp in inputs:
            if inp.dtype != x_dtype:
                inp.to_type(x_dtype)
        dev.EnableGraph(flag)

    def device_check(self, *inputs):
        """Move every input to the device of the first one (never recorded)."""
        x_device = inputs[0].device
        prev_state = x_device.graph_enabled()
        x_device.EnableGraph(False)
        x_dev_id = x_device.id()
        for var in inputs:
            if var.device.id() != x_dev_id:
                var.to_device(x_device)
        x_device.EnableGraph(prev_state)

    def _has_layer_param(self, layer, names):
        for name in names:
            if name.startswith(layer.name):
                return True
        return False

    def _get_name_prefix(self):
        if self.name and self._parent:
            return self.name + Layer.sep
        return ''

    def __getattr__(self, name):
        if '_layers' in self.__dict__:
            layers = self.__dict__['_layers']
            if name in layers:
                return layers[name]
        raise AttributeError("'{}' object has no attribute '{}'".format(
            type(self).__name__, name))

    def __setattr__(self, name, value):
        if isinstance(value, Layer):
            self.__dict__['_layers'][name] = value
            value.__dict__['_parent'] = self
            value.name = self._get_name_prefix() + name
        else:
            object.__setattr__(self, name, value)
            if isinstance(value, Tensor) and value.is_dummy():
                value.name = self._get_name_prefix() + name
            elif name == 'name' and value:
                for n, sublayer in self._layers.items():
                    sublayer.name = self._get_name_prefix() + n

    def __delattr__(self, name):
        if name in self._layers:
            del self._layers[name]
        else:
            object.__delattr__(self, name)

    def register_layers(self, *args):
        """Register sublayers that are not plain attributes (e.g. the blocks
        of a ResNet stage held in a list)."""
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            items = args[0].items()
        else:
            items = [(v.__class__.__name__ + '_' + str(idx), v)
                     for idx, v in enumerate(args)]
        for name, value in items:
            if isinstance(value, Layer):
                self._layers[name] = value
                value.__dict__['_parent'] = self
                value.name = name

    def set_attribute(self, attribute, attribute_value):
        assert attribute_value.shape == attribute.shape, (
            "%s and %s do not match" % (attribute_value.shape, attribute.shape))
        attribute.copy_from(attribute_value)


class Linear(Layer):
    """y = x W + b with W of shape (in_features, out_features)."""

    def __init__(self, out_features, *args, bias=True, **kwargs):
        super(Linear, self).__init__()
        self.out_features = out_features
        # legacy signature Linear(in_features, out_features[, bias])
        if len(args) > 0:
            self.in_features = out_features
            self.out_features = args[0]
        if len(args) > 1:
            self.bias = args[1]
        else:
            self.bias = bias

    def initialize(self, x):
        # features are the last axis (2-d input as in the reference, or a
        # batched (..., in_features) input)
        self.in_features = x.shape[-1]
        w_shape = (self.in_features, self.out_features)
        b_shape = (self.out_features,)
        self.W = Tensor(shape=w_shape, dtype=x.dtype, requires_grad=True,
                        stores_grad=True, device=x.device)
        std = math.sqrt(2.0 / (self.in_features + self.out_features))
        self.W.gaussian(0.0, std)
        if self.bias:
            self.b = Tensor(shape=b_shape, dtype=x.dtype, requires_grad=True,
                            stores_grad=True, device=x.device)
            self.b.set_value(0.0)
        else:
            self.b = None

    def forward(self, x):
        if self.b is not None:
            self.device_check(x, self.W, self.b)
            self.dtype_check(x, self.W, self.b)
        else:
            self.device_check(x, self.W)
            self.dtype_check(x, self.W)
        assert x.shape[-1] == self.W.shape[0], (
            "Linear layer expects input features size %d received %d" %
            (self.W.shape[0], x.shape[-1]))
        y = autograd.matmul(x, self.W)
        if self.bias:
            y = autograd.add_bias(y, self.b, axis=0)
        return y

    def get_params(self):
        if self.bias:
            return {self.W.name: self.W, self.b.name: self.b}
        return {self.W.name: self.W}

    def set_params(self, parameters):
        self.W.copy_from(parameters[self.W.name])
        if self.bias:
            self.b.copy_from(parameters[self.b.name])


class Gemm(Layer):
    """Y = alpha * A' * B' + beta * C with B the weight and C the bias."""

    def __init__(self, nb_kernels, alpha=1.0, beta=1.0, transA=False,
                 transB=True, bias=True, bias_shape=None):
        super(Gemm, self).__init__()
        self.nb_kernels = nb_kernels
        self.alpha = alpha
        self.beta = beta
        self.transA = 1 if transA else 0
        self.transB = 1 if transB else 0
        self.bias = bias
        self.bias_shape = bias_shape

    def initialize(self, x):
        self.in_features = x.shape[-1] if self.transA == 0 else x.shape[0]
        if self.transB == 0:
            w_shape = (self.in_features, self.nb_kernels)
        else:
            w_shape = (self.nb_kernels, self.in_features)
        b_shape = self.bias_shape if self.bias_shape else (1, self.nb_kernels)
        self.W = Tensor(shape=w_shape, requires_grad=True, stores_grad=True,
                        device=x.device)
        std = math.sqrt(2.0 / (self.in_features + self.nb_kernels))
        self.W.gaussian(0.0, std)
        if self.bias:
            self.b = Tensor(shape=b_shape, requires_grad=True,
                            stores_grad=True, device=x.device)
            self.b.set_value(0.0)
        else:
            self.b = None

    def forward(self, x):
        if self.b is not None:
            self.device_check(x, self.W, self.b)
        else:
            self.device_check(x, self.W)
        in_features = x.shape[-1] if self.transA == 0 else x.shape[0]
        in_features_w = self.W.shape[0] if self.transB == 0 \
            else self.W.shape[-1]
        assert in_features == in_features_w, (
            "Gemm layer expects input features size %d received %d" %
            (in_features_w, in_features))
        return autograd.gemm(x, self.W, self.b, self.alpha, self.beta,
                             self.transA, self.transB)

    def get_params(self):
        if self.bias:
            return {self.W.name: self.W, self.b.name: self.b}
        return {self.W.name: self.W}

    def set_params(self, parameters):
        self.W.copy_from(parameters[self.W.name])
        if self.bias:
            self.b.copy_from(parameters[self.b.name])


class Embedding(Layer):

    def __init__(self, input_dim, output_dim, initializer="gaussian"):
        super(Embedding, self).__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.initializer = initializer

    def initialize(self, x):
        w_shape = (self.input_dim, self.output_dim)
        self.W = Tensor(shape=w_shape, requires_grad=True, stores_grad=True,
                        device=x.device)
        if self.initializer == 'uniform':
            self.W.uniform(-1., 1.)
        else:
            self.W.gaussian(0., 1.)

    def from_pretrained(self, W, freeze=True):
        self.set_params({self.W.name: W})
        self.W.requires_grad = not freeze

    def forward(self, x):
        return autograd.embedding(x, self.W)

    def get_params(self):
        return {self.W.name: self.W}

    def set_params(self, parameters):
        self.W.copy_from(parameters[self.W.name])


def _norm_pair(v, what):
    if isinstance(v, int):
        return (v, v)
    if isinstance(v, (tuple, list)):
        return tuple(v)
    raise TypeError("Wrong %s type." % what)


def _split_padding(padding):
    """(symmetric (h, w) padding, odd remainder (top, bottom, left, right))."""
    odd = (0, 0, 0, 0)
    if isinstance(padding, int):
        return (padding, padding), odd
    if isinstance(padding, (tuple, list)):
        if len(padding) == 2:
            return tuple(padding), odd
        if len(padding) == 4:
            hm = padding[0] - padding[1]
            wm = padding[2] - padding[3]
            odd = (max(hm, 0), max(-hm, 0), max(wm, 0), max(-wm, 0))
            return (padding[0] - odd[0], padding[2] - odd[2]), odd
    if padding is None:
        return (0, 0), odd
    raise TypeError("Wrong padding value.")


class Conv2d(Layer):
    """2-d convolution (NCHW).  ``pad_mode`` SAME_UPPER / SAME_LOWER derives
    the padding from the input; ``activation="RELU"`` appends a ReLU."""

    def __init__(self, nb_kernels, kernel_size, *args, stride=1, padding=0,
                 dilation=1, group=1, bias=True, pad_mode="NOTSET",
                 activation="NOTSET", **kwargs):
        super(Conv2d, self).__init__()
        # legacy signature Conv2d(in_channels, out_channels, kernel[, stride
        # [, padding]])
        if len(args) > 0:
            nb_kernels = kernel_size
            kernel_size = args[0]
        if len(args) > 1:
            stride = args[1]
        if len(args) > 2:
            padding = args[2]

        self.nb_kernels = nb_kernels
        self.kernel_size = _norm_pair(kernel_size, "kernel_size")
        self.stride = _norm_pair(stride, "stride")
        self.padding, self.odd_padding = _split_padding(padding)
        self.dilation = dilation
        self.group = group
        self.bias = bias
        self.pad_mode = pad_mode
        self.activation = activation
        if dilation != 1 and list(dilation) != [1, 1]:
            raise ValueError("Not implemented yet")
        self.inner_params = {
            "cudnn_prefer": "fastest",
            "workspace_MB_limit": 1024,
        }
        for kwarg in kwargs:
            if kwarg not in self.inner_params:
                raise TypeError("Keyword argument not understood:", kwarg)
            self.inner_params[kwarg] = kwargs[kwarg]

    def initialize(self, x):
        self.in_channels = x.shape[1]
        w_shape = (self.nb_kernels, int(self.in_channels / self.group),
                   self.kernel_size[0], self.kernel_size[1])
        self.W = Tensor(shape=w_shape, requires_grad=True, stores_grad=True,
                        device=x.device)
        std = math.sqrt(
            2.0 / (w_shape[1] * self.kernel_size[0] * self.kernel_size[1] +
                   self.nb_kernels))
        self.W.gaussian(0.0, std)
        if self.bias:
            self.b = Tensor(shape=(self.nb_kernels,), requires_grad=True,
                            stores_grad=True, device=x.device)
            self.b.set_value(0.0)
        else:
            self.b = None

        if self.pad_mode in ("SAME_UPPER", "SAME_LOWER"):
            pad, self.odd_padding = utils.get_padding_shape(
                self.pad_mode, x.shape[2:], self.kernel_size, self.stride)
            self.padding = [pad[0], pad[2]]

        x_shape = list(x.shape)
        if tuple(self.odd_padding) != (0, 0, 0, 0):
            x_shape[2] += (self.odd_padding[0] + self.odd_padding[1])
            x_shape[3] += (self.odd_padding[2] + self.odd_padding[3])
        probe = singa.Tensor(x_shape, x.device, x.dtype)
        if not hasattr(self, "handle"):
            self.handle = singa.ConvHandle(probe, self.kernel_size,
                                           self.stride, self.padding,
                                           self.in_channels, self.nb_kernels,
                                           self.bias, self.group)

    def forward(self, x):
        self.device_check(x, *[s for k, s in self.get_states().items()])
        self.dtype_check(x, *[s for k, s in self.get_states().items()])
        assert (self.group >= 1 and self.in_channels % self.group
                == 0), "please set reasonable group."
        assert (self.nb_kernels >= self.group and self.nb_kernels % self.group
                == 0), "nb_kernels and group dismatched."
        y = autograd.conv2d(self.handle, x, self.W, self.b, self.odd_padding)
        if self.activation == "RELU":
            y = autograd.relu(y)
        return y

    def get_params(self):
        if self.bias:
            return {self.W.name: self.W, self.b.name: self.b}
        return {self.W.name: self.W}

    def set_params(self, parameters):
        self.W.copy_from(parameters[self.W.name])
        if self.bias:
            self.b.copy_from(parameters[self.b.name])


class SeparableConv2d(Layer):
    """Depth-wise convolution followed by a 1x1 point-wise convolution."""

    def __init__(self, nb_kernels, kernel_size, *args, stride=1, padding=0,
                 bias=False):
        super(SeparableConv2d, self).__init__()
        if len(args) > 0:
            nb_kernels = kernel_size
            kernel_size = args[0]
        if len(args) > 1:
            stride = args[1]
        if len(args) > 2:
            padding = args[2]
        self.nb_kernels = nb_kernels
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = padding
        self.bias = bias

    def initialize(self, x):
        self.in_channels = x.shape[1]
        self.depthwise_conv = Conv2d(self.in_channels, self.kernel_size,
                                     stride=self.stride, padding=self.padding,
                                     group=self.in_channels, bias=self.bias)
        self.point_conv = Conv2d(self.nb_kernels, 1, bias=self.bias)

    def forward(self, x):
        y = self.depthwise_conv(x)
        y = self.point_conv(y)
        return y


class BatchNorm2d(Layer):
    """Batch normalisation over (N, H, W) per channel.  ``momentum`` is the
    weight of the batch statistics in the running average (see
    ``singa_wrap.BatchNormHandle``)."""

    def __init__(self, *args, momentum=0.9, activation=None):
        """``activation="RELU"`` computes relu(bn(x)) -- and, when the layer
        is called with a second tensor, relu(bn(x) + residual) -- inside the
        normalisation kernel instead of as separate graph nodes."""
        super(BatchNorm2d, self).__init__()
        if len(args) > 0:
            self.channels = args[0]
        if len(args) > 1:
            momentum = args[1]
        self.momentum = momentum
        assert 0 <= momentum <= 1.0, ("Illegal momentum")
        assert activation in (None, "RELU"), "only RELU can be fused"
        self.activation = activation

    def initialize(self, x, residual=None):
        self.channels = x.shape[1]
        param_shape = (self.channels,)
        self.scale = Tensor(shape=param_shape, requires_grad=True,
                            stores_grad=True, device=x.device)
        self.scale.set_value(1.0)
        self.bias = Tensor(shape=param_shape, requires_grad=True,
                           stores_grad=True, device=x.device)
        self.bias.set_value(0.0)
        self.running_mean = Tensor(shape=param_shape, requires_grad=False,
                                   stores_grad=False, device=x.device)
        self.running_mean.set_value(0.0)
        self.running_var = Tensor(shape=param_shape, requires_grad=False,
                                  stores_grad=False, device=x.device)
        self.running_var.set_value(1.0)
        if not hasattr(self, "handle"):
            self.handle = singa.BatchNormHandle(self.momentum, x.data)

    def forward(self, x, residual=None):
        assert x.shape[1] == self.channels, (
            "number of channels dismatched. %d vs %d" %
            (x.shape[1], self.channels))
        self.device_check(x, self.scale, self.bias, self.running_mean,
                          self.running_var)
        # statistics and affine parameters stay in float32 for half inputs
        if residual is not None:
            assert self.activation == "RELU", \
                "a residual input needs activation='RELU'"
            return autograd.batchnorm_add_relu_2d(
                self.handle, x, self.scale, self.bias, self.running_mean,
                self.running_var, residual)
        if self.activation == "RELU":
            return autograd.batchnorm_relu_2d(
                self.handle, x, self.scale, self.bias, self.running_mean,
                self.running_var)
        return autograd.batchnorm_2d(self.handle, x, self.scale, self.bias,
                                     self.running_mean, self.running_var)

    def get_params(self):
        return {self.scale.name: self.scale, self.bias.name: self.bias}

    def set_params(self, parameters):
        self.scale.copy_from(parameters[self.scale.name])
        self.bias.copy_from(parameters[self.bias.name])

    def get_states(self):
        ret = self.get_params()
        ret[self.running_mean.name] = self.running_mean
        ret[self.running_var.name] = self.running_var
        return ret

    def set_states(self, states):
        self.set_params(states)
        self.running_mean.copy_from(states[self.running_mean.name])
        self.running_var.copy_from(states[self.running_var.name])


class Pooling2d(Layer):

    def __init__(self, kernel_size, stride=None, padding=0, is_max=True,
                 pad_mode="NOTSET"):
        super(Pooling2d, self).__init__()
        self.kernel_size = _norm_pair(kernel_size, "kernel_size")
        if stride is None:
            self.stride = self.kernel_size
        else:
            self.stride = _norm_pair(stride, "stride")
        self.padding, self.odd_padding = _split_padding(padding)
        self.is_max = is_max
        self.pad_mode = pad_mode

    @staticmethod
    def _as_4d(x):
        """(N, C, L) inputs of the 1-d poolings are pooled as (N, C, 1, L)."""
        if len(x.shape) == 3:
            return autograd.reshape(x, (x.shape[0], x.shape[1], 1, x.shape[2]))
        return x

    def initialize(self, x):
        x = self._as_4d(x)
        if self.pad_mode in ("SAME_UPPER", "SAME_LOWER"):
            pad, self.odd_padding = utils.get_padding_shape(
                self.pad_mode, x.shape[2:], self.kernel_size, self.stride)
            self.padding = [pad[0], pad[2]]
        x_shape = list(x.shape)
        if tuple(self.odd_padding) != (0, 0, 0, 0):
            x_shape[2] += (self.odd_padding[0] + self.odd_padding[1])
            x_shape[3] += (self.odd_padding[2] + self.odd_padding[3])
        probe = singa.Tensor(x_shape, x.device, x.dtype)
        self.handle = singa.PoolingHandle(probe, self.kernel_size, self.stride,
                                          self.padding, self.is_max)

    def forward(self, x):
        if len(x.shape) == 3:
            y = autograd.pooling_2d(self.handle, self._as_4d(x),
                                    self.odd_padding)
            return autograd.reshape(y, (y.shape[0], y.shape[1], y.shape[3]))
        return autograd.pooling_2d(self.handle, x, self.odd_padding)


class MaxPool2d(Pooling2d):

    def __init__(self, kernel_size, stride=None, padding=0, pad_mode="NOTSET"):
        super(MaxPool2d, self).__init__(kernel_size, stride, padding, True,
                                        pad_mode)


class AvgPool2d(Pooling2d):

    def __init__(self, kernel_size, stride=None, padding=0, pad_mode="NOTSET"):
        super(AvgPool2d, self).__init__(kernel_size, stride, padding, False,
                                        pad_mode)


class MaxPool1d(Pooling2d):

    def __init__(self, kernel_size, stride=None, padding=0, pad_mode="NOTSET"):
        if stride is None:
            stride = kernel_size
        super(MaxPool1d, self).__init__((1, kernel_size), (1, stride),
                                        (0, padding), True, pad_mode)


class AvgPool1d(Pooling2d):

    def __init__(self, kernel_size, stride=None, padding=0, pad_mode="NOTSET"):
        if stride is None:
            stride = kernel_size
        super(AvgPool1d, self).__init__((1, kernel_size), (1, stride),
                                        (0, padding), False, pad_mode)


class RNN_Base(Layer):

    def step_forward(self, x=None, h=None, c=None, Wx=None, Wh=None, Bx=None,
                     Bh=None, b=None):
        raise NotImplementedError


class RNN(RNN_Base):
    """Vanilla RNN unrolled over a list of per-step inputs."""

    def __init__(self, input_size, hidden_size, num_layers=1,
                 nonlinearity="tanh", bias=True, batch_first=False, dropout=0,
                 bidirectional=False):
        super(RNN, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.nonlinearity = nonlinearity
        self.bias = bias
        self.batch_first = batch_first
        self.dropout = dropout
        self.bidirectional = bidirectional

    def initialize(self, xs, h0):
        self.Wx = Tensor(shape=(self.input_size, self.hidden_size),
                         requires_grad=True, stores_grad=True, device=xs[0].device)
        self.Wx.gaussian(0.0, 1.0)
        self.Wh = Tensor(shape=(self.hidden_size, self.hidden_size),
                         requires_grad=True, stores_grad=True, device=xs[0].device)
        self.Wh.gaussian(0.0, 1.0)
        self.b = Tensor(shape=(self.hidden_size,), requires_grad=True,
                        stores_grad=True, device=xs[0].device)
        self.b.set_value(0.0)

    def forward(self, xs, h0):
        if not isinstance(xs, tuple):
            xs = tuple(xs)
        inputs = xs + (h0,)
        self.device_check(*inputs)
        self.device_check(inputs[0], self.Wx, self.Wh, self.b)
        batchsize = xs[0].shape[0]
        out = []
        h = self.step_forward(xs[0], h0, self.Wx, self.Wh, self.b)
        out.append(h)
        for x in xs[1:]:
            assert x.shape[0] == batchsize
            h = self.step_forward(x, h, self.Wx, self.Wh, self.b)
            out.append(h)
        return out, h

    def step_forward(self, x, h, Wx, Wh, b):
        y2 = autograd.matmul(h, Wh)
        y1 = autograd.matmul(x, Wx)
        y = autograd.add(y2, y1)
        y = autograd.add_bias(y, b, axis=0)
        if self.nonlinearity == "tanh":
            y = autograd.tanh(y)
        elif self.nonlinearity == "relu":
            y = autograd.relu(y)
        else:
            raise ValueError
        return y

    def get_params(self):
        return {self.Wx.name: self.Wx, self.Wh.name: self.Wh,
                self.b.name: self.b}

    def set_params(self, parameters):
        self.Wx.copy_from(parameters[self.Wx.name])
        self.Wh.copy_from(parameters[self.Wh.name])
        self.b.copy_from(parameters[self.b.name])


class LSTM(RNN_Base):
    """LSTM unrolled over a list of per-step inputs.  Parameters keep the
    reference layout (8 weight matrices + 8 bias vectors: Wx_/Wh_/Bx_/Bh_ for
    gates i, f, o, g); each time step runs as two GEMMs over the concatenated
    gate weights plus one fused cell kernel."""

    _gates = ("i", "f", "o", "g")

    def __init__(self, input_size, hidden_size, nonlinearity="tanh",
                 num_layers=1, bias=True, batch_first=False, dropout=0,
                 bidirectional=False):
        super(LSTM, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.nonlinearity = nonlinearity
        self.bias = bias
        self.batch_first = batch_first
        self.dropout = dropout
        self.bidirectional = bidirectional

    def _all(self):
        ws = [getattr(self, "Wx_" + g) for g in self._gates] + \
             [getattr(self, "Wh_" + g) for g in self._gates]
        bs = [getattr(self, "Bx_" + g) for g in self._gates] + \
             [getattr(self, "Bh_" + g) for g in self._gates]
        return ws, bs

    def initialize(self, xs, h0_c0):
        for g in self._gates:
            setattr(self, "Wx_" + g,
                    Tensor(shape=(self.input_size, self.hidden_size),
                           requires_grad=True, stores_grad=True, device=xs[0].device))
        for g in self._gates:
            setattr(self, "Wh_" + g,
                    Tensor(shape=(self.hidden_size, self.hidden_size),
                           requires_grad=True, stores_grad=True, device=xs[0].device))
        for g in self._gates:
            setattr(self, "Bx_" + g, Tensor(shape=(self.hidden_size,),
                                            requires_grad=True,
                                            stores_grad=True, device=xs[0].device))
        for g in self._gates:
            setattr(self, "Bh_" + g, Tensor(shape=(self.hidden_size,),
                                            requires_grad=True,
                                            stores_grad=True, device=xs[0].device))
        ws, bs = self._all()
        for w in ws:
            w.gaussian(0.0, 0.01)
        for b in bs:
            b.set_value(0.0)

    def forward(self, xs, h0_c0):
        h0, c0 = h0_c0
        if not isinstance(xs, list):
            xs = list(xs)
        inputs = xs + [h0, c0]
        self.device_check(*inputs)
        self.device_check(inputs[0], *[s for k, s in self.get_states().items()])
        batchsize = xs[0].shape[0]
        # gate-concatenated views of the parameters, built once per call
        Wx = autograd.cat([getattr(self, "Wx_" + g) for g in self._gates], 1)
        Wh = autograd.cat([getattr(self, "Wh_" + g) for g in self._gates], 1)
        Bx = autograd.cat([getattr(self, "Bx_" + g) for g in self._gates], 0)
        Bh = autograd.cat([getattr(self, "Bh_" + g) for g in self._gates], 0)
        B = autograd.add(Bx, Bh)
        out = []
        h, c = h0, c0
        for x in xs:
            assert x.shape[0] == batchsize
            h, c = self._fused_step(x, h, c, Wx, Wh, B)
            out.append(h)
        return out, h, c

    def _fused_step(self, x, h, c, Wx, Wh, B):
        gates = autograd.add(autograd.matmul(x, Wx), autograd.matmul(h, Wh))
        gates = autograd.add_bias(gates, B, axis=0)
        return rnn_cells.lstm_cell(gates, c)

    def step_forward(self, x, h, c):
        """One LSTM step from the un-concatenated parameters."""
        def gate(g):
            y1 = autograd.add_bias(
                autograd.matmul(x, getattr(self, "Wx_" + g)),
                getattr(self, "Bx_" + g), axis=0)
            y2 = autograd.add_bias(
                autograd.matmul(h, getattr(self, "Wh_" + g)),
                getattr(self, "Bh_" + g), axis=0)
            return autograd.add(y1, y2)
        i = autograd.sigmoid(gate("i"))
        f = autograd.sigmoid(gate("f"))
        o = autograd.sigmoid(gate("o"))
        g = autograd.tanh(gate("g"))
        cout = autograd.add(autograd.mul(f, c), autograd.mul(i, g))
        hout = autograd.mul(o, autograd.tanh(cout))
        return hout, cout

    def get_params(self):
        ws, bs = self._all()
        return {t.name: t for t in ws + bs}

    def set_params(self, parameters):
        ws, bs = self._all()
        for t in ws + bs:
            t.copy_from(parameters[t.name])


# ---- layers without parameters --------------------------------------------
class LayerNorm(Layer):
    """Normalisation over the last dimension with a learned scale and bias
    (an addition over the reference's layer set; one fused kernel per
    direction)."""

    def __init__(self, eps=1e-5):
        super(LayerNorm, self).__init__()
        self.eps = eps

    def initialize(self, x):
        d = x.shape[-1]
        self.scale = Tensor(shape=(d,), requires_grad=True, stores_grad=True,
                            device=x.device, dtype=x.dtype)
        self.scale.set_value(1.0)
        self.bias = Tensor(shape=(d,), requires_grad=True, stores_grad=True,
                           device=x.device, dtype=x.dtype)
        self.bias.set_value(0.0)

    def forward(self, x):
        self.device_check(x, self.scale, self.bias)
        return autograd.layer_norm(x, self.scale, self.bias, self.eps)

    def get_params(self):
        return {self.scale.name: self.scale, self.bias.name: self.bias}

    def set_params(self, parameters):
        self.scale.copy_from(parameters[self.scale.name])
        self.bias.copy_from(parameters[self.bias.name])


class Gelu(Layer):

    def forward(self, x):
        return autograd.gelu(x)


class ReLU(Layer):

    def forward(self, x):
        return autograd.relu(x)


class Sigmoid(Layer):

    def forward(self, x):
        return autograd.sigmoid(x)


class Add(Layer):

    def forward(self, a, b):
        return autograd.add(a, b)


class Flatten(Layer):

    def __init__(self, axis=1):
        super(Flatten, self).__init__()
        self.axis = axis

    def forward(self, x):
        return autograd.flatten(x, self.axis)


class SoftMaxCrossEntropy(Layer):

    def forward(self, x, t):
        return autograd.softmax_cross_entropy(x, t)


class SoftMax(Layer):

    def forward(self, x):
        return autograd.softmax(x)


class MeanSquareError(Layer):

    def forward(self, x, t):
        return autograd.mse_loss(x, t)


class CrossEntropy(Layer):

    def forward(self, x, t):
        return autograd.cross_entropy(x, t)


class BinaryCrossEntropy(Layer):

    def forward(self, x, t):
        return autograd.binary_cross_entropy(x, t)


class Dropout(Layer):
    """Like the reference, ``ratio`` is passed positionally into
    ``autograd.dropout(x, seed, ratio)`` and therefore lands in ``seed``; the
    effective drop ratio is the operator's default 0.5 (reference
    python/singa/layer.py:1522)."""

    def __init__(self, ratio=0.5):
        super(Dropout, self).__init__()
        self.ratio = ratio

    def forward(self, x):
        return autograd.dropout(x, self.ratio)


class Cat(Layer):

    def __init__(self, axis=0):
        super(Cat, self).__init__()
        self.axis = axis

    def forward(self, xs):
        return autograd.cat(xs, self.axis)


class Reshape(Layer):

    def forward(self, x, shape):
        return autograd.reshape(x, shape)


class CudnnRNN(Layer):
    """Multi-layer (bi)directional vanilla / LSTM / GRU network with a single
    flat weight tensor, run by fused recurrent HIP kernels on the GPU and by
    the same algorithm on the host."""

    def __init__(self, hidden_size, activation="tanh", num_layers=1, bias=True,
                 batch_first=True, dropout=0, bidirectional=False,
                 rnn_mode="lstm", use_mask=False, return_sequences=True):
        assert num_layers > 0, "num layers should be > 0"
        assert 0 <= dropout < 1, "dropout shouldbe >=0 and <1"
        super(CudnnRNN, self).__init__()
        self.rnn_mode = rnn_mode
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.dropout = dropout
        self.bidirectional = 1 if bidirectional else 0
        self.return_sequences = return_sequences
        self.batch_first = batch_first
        self.use_mask = use_mask
        # mode codes: 0 relu rnn, 1 tanh rnn, 2 lstm, 3 gru
        self.cudnn_rnn_mode = {"lstm": 2, "vanilla": 1, "tanh": 1, "relu": 0,
                               "gru": 3}[self.rnn_mode]

    def initialize(self, x, hx=None, cx=None, seq_lengths=None):
        if self.batch_first:
            x = x.transpose((1, 0, 2))
        self.input_size = x.shape[1]
        self.handle = singa.CudnnRNNHandle(x.data, self.hidden_size,
                                           mode=self.cudnn_rnn_mode,
                                           num_layers=self.num_layers,
                                           dropout=self.dropout,
                                           bidirectional=self.bidirectional)
        self.W = Tensor(shape=(self.handle.weights_size,), requires_grad=True,
                        stores_grad=True, device=x.device)
        k = 1 / self.hidden_size
        self.W.uniform(-math.sqrt(k), math.sqrt(k))

    def forward(self, x, hx=None, cx=None, seq_lengths=None):
        self.device_check(x, self.W)
        if self.batch_first:
            x = autograd.transpose(x, (1, 0, 2))
        batch_size = x.shape[1]
        directions = 2 if self.bidirectional else 1
        state_shape = (self.num_layers * directions, batch_size,
                       self.hidden_size)
        if hx is None:
            hx = Tensor(shape=state_shape, requires_grad=False,
                        stores_grad=False, device=x.device).set_value(0.0)
        if cx is None:
            cx = Tensor(shape=state_shape, requires_grad=False,
                        stores_grad=False, device=x.device).set_value(0.0)
        if self.use_mask:
            assert type(seq_lengths) == Tensor, "wrong type for seq_lengths"
            y = autograd._RNN(self.handle,
                              return_sequences=self.return_sequences,
                              use_mask=self.use_mask,
                              seq_lengths=seq_lengths)(x, hx, cx, self.W)[0]
        else:
            y = autograd._RNN(self.handle,
                              return_sequences=self.return_sequences)(
                                  x, hx, cx, self.W)[0]
        if self.return_sequences and self.batch_first:
            y = autograd.transpose(y, (1, 0, 2))
        return y

    def get_params(self):
        return {self.W.name: self.W}

    def set_params(self, parameters):
        self.set_attribute(self.W, parameters[self.W.name])


# imported last: autograd imports tensor which must not import layer
from . import autograd  # noqa: E402
from .ops import rnn_cells  # noqa: E402
