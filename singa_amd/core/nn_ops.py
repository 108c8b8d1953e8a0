"""Convolution, batch-norm and pooling handles plus their forward / backward
functions.  One implementation serves both devices: the op body asks the
device's math backend (torch on the host, the hand-written MFMA / LDS kernels
on the GPU), so ``Cpu*`` and ``Gpu*`` names are aliases of the same function.

Behaviour parity: reference src/model/operation/{convolution,batchnorm,
pooling}.{h,cc}; binding names from src/api/model_operation.i:37-191.
"""
import os

from . import dtypes
from .tensor_core import Tensor, _submit, Contiguous

# SINGA_CONV_BN_STATS=1: a convolution takes the statistics of the batch
# normalisation that reads its output (ChannelStatLink).  Off by default: on
# MI355X the per-tile atomics of the epilogue cost more than the statistics
# pass they replace (measured per layer in profiles/bn_stats_in_conv_epilogue.md).
CONV_BN_STATS = os.environ.get("SINGA_CONV_BN_STATS", "0") == "1"


def _pair(v):
    if isinstance(v, (list, tuple)):
        return int(v[0]), int(v[1])
    return int(v), int(v)


class ConvHandle(object):
    """Geometry of one 2-d convolution (NCHW, cross-correlation, dilation 1,
    optional groups).  reference convolution.cc:29-70."""

    def __init__(self, x, kernel_size, stride, padding, in_channels,
                 out_channels, bias, groups=1,
                 workspace_byte_limit=1024 * 1024 * 1024, prefer="fastest"):
        self.kernel_h, self.kernel_w = _pair(kernel_size)
        self.pad_h, self.pad_w = _pair(padding)
        self.stride_h, self.stride_w = _pair(stride)
        self.channels = int(in_channels)
        self.num_filters = int(out_channels)
        self.group = int(groups)
        self.bias_term = bool(bias)
        self.batchsize = x.shape(0)
        assert x.shape(1) == self.channels, (
            "the number of input channels mismatched: %d vs %d" %
            (x.shape(1), self.channels))
        self.height = x.shape(2)
        self.width = x.shape(3)
        self.conv_height = 1
        if self.stride_h > 0:
            self.conv_height = (self.height + 2 * self.pad_h -
                                self.kernel_h) // self.stride_h + 1
        self.conv_width = (self.width + 2 * self.pad_w -
                           self.kernel_w) // self.stride_w + 1
        self.col_height = self.channels * self.kernel_w * self.kernel_h
        self.col_width = self.conv_height * self.conv_width
        self.imagesize = x.Size() // self.batchsize
        self.workspace_byte_limit = workspace_byte_limit
        self.prefer = prefer
        # set by the batch normalisation that reads this convolution's output
        self.stat_link = None
        self.stat_link_refused = False

    def out_shape(self, batch):
        return [batch, self.num_filters, self.conv_height, self.conv_width]


CudnnConvHandle = ConvHandle


class ChannelStatLink(object):
    """Hand-over of batch-norm statistics from a convolution to the batch
    normalisation that reads its output: the convolution kernel's epilogue
    adds the per-channel sums of (y - shift) and (y - shift)^2 into ``stats``
    while y is still in registers, and the normalisation takes them from there
    instead of reading y once more for a statistics pass.  ``shift`` is the
    normalisation's running mean as it is before the step (any value works;
    one near the batch mean keeps E[d^2] - E[d]^2 free of cancellation).  The
    normalisation clears the buffer once it has read it.

    ``token`` counts emissions and ``pending`` says that the last one has not
    been consumed: a normalisation only trusts the sums of the very tensor it
    was handed, and a convolution whose previous sums nobody took clears them
    before adding new ones."""

    def __init__(self, dev, channels, shift):
        math = dev.math
        self.stats = Tensor([(2 * math.STAT_COPIES + 1) * channels], dev,
                            dtypes.kFloat32)
        self.shift = shift
        self.token = 0
        self.pending = False
        self.clear()

    def clear(self):
        dev = self.stats.device()
        was = dev.graph_enabled()
        dev.EnableGraph(False)      # parameters-like state: set right now
        try:
            self.stats.SetValue(0.0)
        finally:
            dev.EnableGraph(was)


def stat_link_for(x, running_mean, channels):
    """The link through which the producer of ``x`` handed statistics to the
    batch normalisation owning ``running_mean``, or None.  A convolution seen
    for the first time is asked to emit them from its next run on."""
    source = getattr(x, "_stat_source", None)
    if source is None or not CONV_BN_STATS:
        return None
    handle, token = source
    link = handle.stat_link
    if link is None:
        if not handle.stat_link_refused and \
                running_mean.data_type() == dtypes.kFloat32 and \
                handle.num_filters == channels:
            handle.stat_link = ChannelStatLink(x.device(), channels,
                                               running_mean)
        return None
    if link.shift is not running_mean:
        # two normalisations read this convolution: keep both on their own pass
        handle.stat_link, handle.stat_link_refused = None, True
        return None
    if token != link.token or not link.pending:
        return None
    link.pending = False
    return link


def _check_conv_input(x, ch):
    assert x.nDim() == 4 and x.shape(1) == ch.channels and \
        x.shape(2) == ch.height and x.shape(3) == ch.width, "input sample not match"


def ConvForward(x, W, b, ch, emit_stats=False):
    """``emit_stats``: this is a training step; when a batch normalisation
    linked itself to the handle (``ChannelStatLink``) the launch also produces
    its statistics, and the output remembers where it came from so that the
    normalisation can find them."""
    _check_conv_input(x, ch)
    xs = x if x.is_contiguous() else Contiguous(x)
    y = Tensor(ch.out_shape(x.shape(0)), x.device(), x.data_type())
    st, pd, g = (ch.stride_h, ch.stride_w), (ch.pad_h, ch.pad_w), ch.group
    bias = b if ch.bias_term and b is not None and b.Size() > 0 else None
    link = ch.stat_link if emit_stats else None
    if link is not None and link.stats.device() is not x.device():
        link = None
    if link is not None:
        if link.pending:
            link.stats.SetValue(0.0)       # sums nobody consumed
        link.token += 1
        link.pending = True
    y._stat_source = (ch, link.token if link is not None else -1)
    reads = [xs, W] + ([bias] if bias is not None else [])
    nb = bias is not None
    if link is None:
        def body(a, w, *rest):
            bb, o = (rest[0], rest[1]) if nb else (None, rest[0])
            a._dev.math.conv_fwd(a.view(), w.view(),
                                 bb.view() if nb else None, o.view(), st, pd,
                                 g)
        _submit(x.device(), "ConvForward", body, reads, [y])
    else:
        def body(a, w, *rest):
            bb = rest[0] if nb else None
            sh, _, o, sums = rest[-4:]
            a._dev.math.conv_fwd(a.view(), w.view(),
                                 bb.view() if nb else None, o.view(), st, pd,
                                 g, sums.view(), sh.view())
        _submit(x.device(), "ConvForward", body,
                reads + [link.shift, link.stats], [y, link.stats])
    return y


def ConvBackwardx(dy, W, x, ch, addend=None):
    """Gradient w.r.t. the convolution input; ``addend`` (a tensor of x's
    shape) is added to it inside the kernel."""
    d = dy if dy.is_contiguous() else Contiguous(dy)
    dx = Tensor(x.shape(), x.device(), x.data_type())
    st, pd, g = (ch.stride_h, ch.stride_w), (ch.pad_h, ch.pad_w), ch.group
    if addend is None:
        _submit(x.device(), "ConvBackwardx",
                lambda a, w, o: a._dev.math.conv_bwd_data(
                    a.view(), w.view(), o.view(), st, pd, g),
                [d, W], [dx])
    else:
        ad = addend if addend.is_contiguous() else Contiguous(addend)
        _submit(x.device(), "ConvBackwardxAcc",
                lambda a, w, p, o: a._dev.math.conv_bwd_data(
                    a.view(), w.view(), o.view(), st, pd, g, p.view()),
                [d, W, ad], [dx])
    return dx


def ConvBackwardxW(dy, W, x, ch, addend=None):
    """(dx, dW) of one layer as one recorded operation: the data-gradient
    launch clears dW on the side where the backend supports it, so the
    weight-gradient launch accumulates its partial sums without a zero-fill
    pass of its own."""
    d = dy if dy.is_contiguous() else Contiguous(dy)
    xs = x if x.is_contiguous() else Contiguous(x)
    dx = Tensor(x.shape(), x.device(), x.data_type())
    dW = Tensor(W.shape(), W.device(), W.data_type())
    st, pd, g = (ch.stride_h, ch.stride_w), (ch.pad_h, ch.pad_w), ch.group
    reads = [d, W, xs]
    if addend is not None:
        reads.append(addend if addend.is_contiguous() else Contiguous(addend))

    def body(a, w, xx, *rest):
        p, odx, odw = (rest[0], rest[1], rest[2]) if len(rest) == 3 else \
            (None, rest[0], rest[1])
        m = a._dev.math
        if _PAIR_STREAMS and a._dev.lang() != 0 and \
                a._dtype in _SIXTEEN_BIT:
            _backward_pair_on_two_streams(a, w, xx, p, odx, odw, st, pd, g)
            return
        cleared = m.conv_bwd_data(a.view(), w.view(), odx.view(), st, pd, g,
                                  None if p is None else p.view(),
                                  side_zero=odw.view())
        m.conv_bwd_weight(a.view(), xx.view(), odw.view(), st, pd, g,
                          zeroed=bool(cleared))
    _submit(x.device(), "ConvBackwardxW", body, reads, [dx, dW])
    return dx, dW


# 16-bit layers at small batch: each of the two gradient kernels of a layer
# leaves compute units idle (10-50 us launches of a few hundred workgroups),
# so the weight gradient runs on a second stream next to the data gradient
# and is joined before the operation ends -- inside a captured step the two
# become parallel branches of the hipGraph.  Measured on the ResNet-50 layers
# at batch 32: 3.32 -> 3.06 ms for the pairs in fp16, no gain in float32 (whose
# kernels fill the chip), hence 16-bit only.  SINGA_CONV_PAIR_STREAMS=0 keeps
# one stream.
_PAIR_STREAMS = os.environ.get("SINGA_CONV_PAIR_STREAMS", "1") != "0"
_SIXTEEN_BIT = (dtypes.kFloat16, dtypes.kBFloat16)


def _backward_pair_on_two_streams(a, w, xx, p, odx, odw, st, pd, g):
    import torch
    dev = a._dev
    m = dev.math
    side = getattr(dev, "_stream_wgrad", None)
    if side is None:
        side = dev._stream_wgrad = torch.cuda.Stream(device=dev.torch_device)
    # operands and results are bound (and the results' storage allocated) on
    # the operation's own stream: only the launches move to the second one
    av, wv, xv = a.view(), w.view(), xx.view()
    dxv, dwv = odx.view(), odw.view()
    pv = None if p is None else p.view()
    cur = torch.cuda.current_stream(dev.torch_device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        m.conv_bwd_weight(av, xv, dwv, st, pd, g, zeroed=False)
    m.conv_bwd_data(av, wv, dxv, st, pd, g, pv, side_zero=None)
    cur.wait_stream(side)


def ConvBackwardW(dy, x, W, ch):
    d = dy if dy.is_contiguous() else Contiguous(dy)
    xs = x if x.is_contiguous() else Contiguous(x)
    dW = Tensor(W.shape(), W.device(), W.data_type())
    st, pd, g = (ch.stride_h, ch.stride_w), (ch.pad_h, ch.pad_w), ch.group
    _submit(x.device(), "ConvBackwardW",
            lambda a, xx, o: a._dev.math.conv_bwd_weight(
                a.view(), xx.view(), o.view(), st, pd, g),
            [d, xs], [dW])
    return dW


def ConvBackwardb(dy, b, ch):
    d = dy if dy.is_contiguous() else Contiguous(dy)
    db = Tensor(b.shape(), b.device(), b.data_type())
    _submit(dy.device(), "ConvBackwardb",
            lambda a, o: a._dev.math.conv_bwd_bias(a.view(), o.view()),
            [d], [db])
    return db


CpuConvForward = GpuConvForward = ConvForward
CpuConvBackwardx = GpuConvBackwardx = ConvBackwardx
CpuConvBackwardW = GpuConvBackwardW = ConvBackwardW
CpuConvBackwardxW = GpuConvBackwardxW = ConvBackwardxW
CpuConvBackwardb = GpuConvBackwardb = ConvBackwardb


class BatchNormHandle(object):
    """``factor`` is the weight of the *batch* statistics in the running
    average (running = (1 - factor) * running + factor * batch), i.e. the
    constructor argument is used verbatim, and the variance that enters the
    running average is the unbiased one.  reference batchnorm.cc:27-60 and
    test/python/test_api.py:31-54."""

    eps = 1e-5

    def __init__(self, momentum, x):
        self.factor = float(momentum)
        self.batchsize = x.shape(0)
        self.channels = x.shape(1)
        if x.nDim() == 4:
            self.height, self.width, self.is_2d = x.shape(2), x.shape(3), False
        elif x.nDim() == 2:
            self.height, self.width, self.is_2d = 1, 1, True
        else:
            raise ValueError("Expecting input tensor to be of dimension 2 or 4")


CudnnBatchNormHandle = BatchNormHandle


BN_ACT_NONE, BN_ACT_RELU, BN_ACT_ADD_RELU = 0, 1, 2


def _contig(t):
    return t if t is None or t.is_contiguous() else Contiguous(t)


def BatchNormForwardTraining(bnh, x, scale, bias, running_mean, running_var,
                             act=BN_ACT_NONE, residual=None):
    """Returns [y, batch_mean, batch_inv_std]; updates the running statistics
    in place.  ``act`` fuses an activation into the normalisation pass:
    BN_ACT_RELU gives relu(bn(x)), BN_ACT_ADD_RELU relu(bn(x) + residual).
    When x comes from a convolution that already took the statistics
    (``ChannelStatLink``) the statistics pass over x is skipped."""
    link = stat_link_for(x, running_mean, bnh.channels)
    xs, res = _contig(x), _contig(residual)
    if xs is not x:
        link = None
    dev = x.device()
    y = Tensor(x.shape(), dev, x.data_type())
    mean = Tensor([bnh.channels], dev, dtypes.kFloat32)
    invstd = Tensor([bnh.channels], dev, dtypes.kFloat32)
    f, eps = bnh.factor, bnh.eps
    with_res = act == BN_ACT_ADD_RELU
    reads = [xs, scale, bias, running_mean, running_var]
    if with_res:
        reads.append(res)
    if link is not None:
        reads.append(link.stats)
    n_in = len(reads)

    def body(*t):
        a, s, b = t[0], t[1], t[2]
        r = t[5] if with_res else None
        sums = t[n_in - 1] if link is not None else None
        o, m, iv, rm, rv = t[n_in:n_in + 5]
        a._dev.math.bn_fwd_train(
            a.view(), s.view(), b.view(), rm.view(), rv.view(), f, eps,
            o.view(), m.view(), iv.view(), act,
            r.view() if r is not None else None,
            **({"stats": sums.view()} if sums is not None else {}))

    writes = [y, mean, invstd, running_mean, running_var]
    if link is not None:
        writes.append(link.stats)
    _submit(dev, "BatchNormForwardTraining", body, reads, writes)
    return [y, mean, invstd]


def BatchNormForwardInference(bnh, x, scale, bias, running_mean, running_var,
                              act=BN_ACT_NONE, residual=None):
    # a first (dry or evaluation) pass already tells the convolution in front
    # that a normalisation follows
    stat_link_for(x, running_mean, bnh.channels)
    xs, res = _contig(x), _contig(residual)
    dev = x.device()
    y = Tensor(x.shape(), dev, x.data_type())
    eps = bnh.eps
    if act == BN_ACT_ADD_RELU:
        _submit(dev, "BatchNormForwardInference",
                lambda a, s, b, rm, rv, r, o: a._dev.math.bn_fwd_infer(
                    a.view(), s.view(), b.view(), rm.view(), rv.view(), eps,
                    o.view(), act, r.view()),
                [xs, scale, bias, running_mean, running_var, res], [y])
    else:
        _submit(dev, "BatchNormForwardInference",
                lambda a, s, b, rm, rv, o: a._dev.math.bn_fwd_infer(
                    a.view(), s.view(), b.view(), rm.view(), rv.view(), eps,
                    o.view(), act),
                [xs, scale, bias, running_mean, running_var], [y])
    return y


def BatchNormBackward(bnh, dy, x, scale, mean, invstd, act=BN_ACT_NONE,
                      bias=None, y=None):
    """Returns [dx, dscale, dbias] (+ [dresidual] for BN_ACT_ADD_RELU).  With
    a fused activation ``dy`` is the gradient of the activated output:
    BN_ACT_RELU needs ``bias`` (the mask is recomputed from x), BN_ACT_ADD_RELU
    needs the saved output ``y``."""
    d, xs = _contig(dy), _contig(x)
    dev = x.device()
    dx = Tensor(x.shape(), dev, x.data_type())
    dscale = Tensor(scale.shape(), dev, scale.data_type())
    dbias = Tensor(scale.shape(), dev, scale.data_type())
    if act == BN_ACT_RELU:
        _submit(dev, "BatchNormBackward",
                lambda a, xx, s, b, m, iv, o, ds, db: a._dev.math.bn_bwd(
                    a.view(), xx.view(), s.view(), m.view(), iv.view(),
                    o.view(), ds.view(), db.view(), act, b.view()),
                [d, xs, scale, bias, mean, invstd], [dx, dscale, dbias])
        return [dx, dscale, dbias]
    if act == BN_ACT_ADD_RELU:
        dres = Tensor(x.shape(), dev, x.data_type())
        ys = _contig(y)
        _submit(dev, "BatchNormBackward",
                lambda a, xx, yy, s, m, iv, o, ds, db, dr:
                a._dev.math.bn_bwd(
                    a.view(), xx.view(), s.view(), m.view(), iv.view(),
                    o.view(), ds.view(), db.view(), act, None, yy.view(),
                    dr.view()),
                [d, xs, ys, scale, mean, invstd], [dx, dscale, dbias, dres])
        return [dx, dscale, dbias, dres]
    _submit(dev, "BatchNormBackward",
            lambda a, xx, s, m, iv, o, ds, db: a._dev.math.bn_bwd(
                a.view(), xx.view(), s.view(), m.view(), iv.view(), o.view(),
                ds.view(), db.view()),
            [d, xs, scale, mean, invstd], [dx, dscale, dbias])
    return [dx, dscale, dbias]


GpuBatchNormForwardTraining = CpuBatchNormForwardTraining = BatchNormForwardTraining
GpuBatchNormForwardInference = CpuBatchNormForwardInference = BatchNormForwardInference
GpuBatchNormBackward = BatchNormBackward


def CpuBatchNormBackwardx(bnh, y, dy, x, scale, bias, mean, var):
    return BatchNormBackward(bnh, dy, x, scale, mean, var)


class PoolingHandle(object):
    """2-d max / average pooling; average excludes the padding from the
    divisor and the output size is rounded down.  reference pooling.cc:27-60."""

    def __init__(self, x, kernel_size, stride, padding, is_max=True):
        self.kernel_h, self.kernel_w = _pair(kernel_size)
        self.pad_h, self.pad_w = _pair(padding)
        self.stride_h, self.stride_w = _pair(stride)
        self.batchsize = x.shape(0)
        self.channels = x.shape(1)
        self.height = x.shape(2)
        self.width = x.shape(3)
        self.pooled_height = 1
        if self.stride_h > 0:
            self.pooled_height = (self.height + 2 * self.pad_h -
                                  self.kernel_h) // self.stride_h + 1
        self.pooled_width = (self.width + 2 * self.pad_w -
                             self.kernel_w) // self.stride_w + 1
        self.is_max_pooling = bool(is_max)
        assert self.pooled_height > 0 and self.pooled_width > 0, (
            "pooling window %dx%d does not fit the %dx%d input (padding "
            "%d,%d)" % (self.kernel_h, self.kernel_w, self.height, self.width,
                        self.pad_h, self.pad_w))

    def _args(self):
        sh = self.stride_h if self.stride_h > 0 else 1
        return ((self.kernel_h, self.kernel_w), (sh, self.stride_w),
                (self.pad_h, self.pad_w), self.is_max_pooling)


CudnnPoolingHandle = PoolingHandle


def PoolingForward(ph, x):
    xs = x if x.is_contiguous() else Contiguous(x)
    y = Tensor([x.shape(0), x.shape(1), ph.pooled_height, ph.pooled_width],
               x.device(), x.data_type())
    k, s, p, mx = ph._args()
    _submit(x.device(), "PoolingForward",
            lambda a, o: a._dev.math.pool_fwd(a.view(), o.view(), k, s, p, mx),
            [xs], [y])
    return y


def PoolingBackward(ph, dy, x, y):
    d = dy if dy.is_contiguous() else Contiguous(dy)
    xs = x if x.is_contiguous() else Contiguous(x)
    dx = Tensor(x.shape(), x.device(), x.data_type())
    k, s, p, mx = ph._args()
    _submit(x.device(), "PoolingBackward",
            lambda a, xx, yy, o: a._dev.math.pool_bwd(
                a.view(), xx.view(), yy.view(), o.view(), k, s, p, mx),
            [d, xs, y], [dx])
    return dx


GpuPoolingForward = CpuPoolingForward = PoolingForward
GpuPoolingBackward = CpuPoolingBackward = PoolingBackward


# ---------------------------------------------------------------------------
# layer normalisation over the last dimension, erf-form GELU
# ---------------------------------------------------------------------------
def LayerNormForward(x, gamma, beta, eps=1e-5):
    """-> (y, mean, rstd); mean / rstd are float32 vectors with one entry per
    row (all leading dimensions flattened), kept for the backward pass."""
    xs = _contig(x)
    dev = x.device()
    D = x.shape()[-1]
    rows = max(1, x.Size() // max(D, 1))
    y = Tensor(x.shape(), dev, x.data_type())
    mean = Tensor([rows], dev, dtypes.kFloat32)
    rstd = Tensor([rows], dev, dtypes.kFloat32)
    eps = float(eps)
    ins = [xs] + [t for t in (gamma, beta) if t is not None]
    has_g, has_b = gamma is not None, beta is not None

    def body(*a):
        xx = a[0]
        k = 1
        g = a[k].view() if has_g else None
        k += has_g
        b = a[k].view() if has_b else None
        k += has_b
        o, m, r = a[k:k + 3]
        xx._dev.math.layernorm_fwd(xx.view(), g, b, o.view(), m.view(),
                                   r.view(), eps)

    _submit(dev, "LayerNormForward", body, ins, [y, mean, rstd])
    return y, mean, rstd


def LayerNormBackward(dy, x, gamma, mean, rstd, with_params=True):
    """-> [dx, dgamma, dbeta] (the last two None unless ``with_params``)."""
    d, xs = _contig(dy), _contig(x)
    dev = x.device()
    D = x.shape()[-1]
    dx = Tensor(x.shape(), dev, x.data_type())
    pdt = gamma.data_type() if gamma is not None else x.data_type()
    outs = [dx]
    dgamma = dbeta = None
    if with_params:
        dgamma = Tensor([D], dev, pdt)
        dbeta = Tensor([D], dev, pdt)
        outs += [dgamma, dbeta]
    has_g = gamma is not None
    ins = [d, xs] + ([gamma] if has_g else []) + [mean, rstd]

    def body(*a):
        dd, xx = a[0], a[1]
        g = a[2].view() if has_g else None
        m, r = a[2 + has_g], a[3 + has_g]
        o = a[4 + has_g:]
        xx._dev.math.layernorm_bwd(
            dd.view(), xx.view(), g, m.view(), r.view(), o[0].view(),
            o[1].view() if with_params else None,
            o[2].view() if with_params else None)

    _submit(dev, "LayerNormBackward", body, ins, outs)
    return [dx, dgamma, dbeta]


def GeluForward(x):
    xs = _contig(x)
    y = Tensor(x.shape(), x.device(), x.data_type())
    _submit(x.device(), "GeluForward",
            lambda a, o: a._dev.math.gelu_fwd(a.view(), o.view()), [xs], [y])
    return y


def GeluBackward(dy, x):
    d, xs = _contig(dy), _contig(x)
    dx = Tensor(x.shape(), x.device(), x.data_type())
    _submit(x.device(), "GeluBackward",
            lambda a, b, o: a._dev.math.gelu_bwd(a.view(), b.view(), o.view()),
            [d, xs], [dx])
    return dx


# ---------------------------------------------------------------------------
# fused scaled-dot-product attention
# ---------------------------------------------------------------------------
def AttentionForward(q, kt, v, mask, scale):
    """softmax(scale * q @ kt + mask) @ v for q (B, H, Sq, D), kt (B, H, D,
    Sk), v (B, H, Sk, D) -- typically transposed *views* of (B, S, H, D)
    activations, which the kernel reads in place.  Returns (out, lse): out is
    (B, Sq, H, D) contiguous (the layout the following merge-heads reshape
    wants), lse (B, H, Sq) float32 for the backward pass."""
    dev = q.device()
    B, H, Sq, D = q.shape()
    out = Tensor([B, Sq, H, D], dev, q.data_type())
    lse = Tensor([B, H, Sq], dev, dtypes.kFloat32)
    scale = float(scale)
    if mask is None:
        _submit(dev, "AttentionForward",
                lambda a, b, c, o, l: a._dev.math.attention_fwd(
                    a.view(), b.view(), c.view(), None, scale, o.view(),
                    l.view()),
                [q, kt, v], [out, lse])
    else:
        _submit(dev, "AttentionForward",
                lambda a, b, c, m, o, l: a._dev.math.attention_fwd(
                    a.view(), b.view(), c.view(), m.view(), scale, o.view(),
                    l.view()),
                [q, kt, v, mask], [out, lse])
    return out, lse


def AttentionBackward(dout, q, kt, v, mask, out, lse, scale):
    """-> (dq, dkt, dv) shaped like q, kt, v.  They are transposed views of
    contiguous (B, S, H, D) buffers, so the transposes upstream undo them
    without a copy."""
    from .tensor_core import Transpose
    dev = q.device()
    B, H, Sq, D = q.shape()
    Sk = v.shape()[2]
    dq = Tensor([B, Sq, H, D], dev, q.data_type())
    dk = Tensor([B, Sk, H, D], dev, q.data_type())
    dv = Tensor([B, Sk, H, D], dev, q.data_type())
    scale = float(scale)
    d = _contig(dout)
    if mask is None:
        _submit(dev, "AttentionBackward",
                lambda g, a, b, c, o, l, x, y, z: a._dev.math.attention_bwd(
                    g.view(), a.view(), b.view(), c.view(), None, scale,
                    o.view(), l.view(), x.view(), y.view(), z.view()),
                [d, q, kt, v, out, lse], [dq, dk, dv])
    else:
        _submit(dev, "AttentionBackward",
                lambda g, a, b, c, m, o, l, x, y, z:
                a._dev.math.attention_bwd(
                    g.view(), a.view(), b.view(), c.view(), m.view(), scale,
                    o.view(), l.view(), x.view(), y.view(), z.view()),
                [d, q, kt, v, mask, out, lse], [dq, dk, dv])
    return (Transpose(dq, [0, 2, 1, 3]), Transpose(dk, [0, 2, 3, 1]),
            Transpose(dv, [0, 2, 1, 3]))
