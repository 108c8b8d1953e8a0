"""Helpers shared by layers, autograd and the ONNX bridge: odd / SAME padding
arithmetic, handle re-creation, tape traversal for export, progress bar.

API parity: reference python/singa/utils.py:27-267.  Odd padding is applied
with one embed / slice kernel per call instead of a chain of concatenations.
"""
import sys
from collections import OrderedDict

import numpy as np

from .. import singa_wrap as singa


def update_progress(progress, info):
    """Text progress bar; progress in [0, 1], negative = halt, >= 1 = done."""
    bar_length = 20
    status = ""
    if isinstance(progress, int):
        progress = float(progress)
    if not isinstance(progress, float):
        progress = 0
        status = "error: progress var must be float. "
    if progress < 0:
        progress = 0
        status = "Halt. "
    if progress >= 1:
        progress = 1
        status = "Done. "
    status = status + info
    block = int(round(bar_length * progress))
    text = "[{0}] {1:3.1f}% {2}".format(
        "." * block + " " * (bar_length - block), progress * 100, status)
    sys.stdout.write(text)
    sys.stdout.write('\b' * (9 + bar_length + len(status)))
    sys.stdout.flush()


def handle_odd_pad_fwd(x, odd_padding, is_pool=False):
    """Apply the asymmetric part of a padding: (top, bottom, left, right)
    extra rows / columns.  Convolution pads with zeros; pooling replicates the
    border so the extra cells never win a max or bias an average."""
    top, bottom, left, right = [int(p) for p in odd_padding]
    shape = list(x.shape())
    if not is_pool:
        out_shape = [shape[0], shape[1], shape[2] + top + bottom,
                     shape[3] + left + right]
        starts = [0, 0, top, left]
        ends = [shape[0], shape[1], top + shape[2], left + shape[3]]
        return singa.EmbedND(x, out_shape, starts, ends, None, 0.0)
    flags = [(2, True), (2, False), (3, True), (3, False)]
    for (axis, is_left), pad in zip(flags, (top, bottom, left, right)):
        if pad == 0:
            continue
        n = list(x.shape())[axis]
        if is_left:
            padding = singa.SliceOn(x, 0, pad, axis)
            x = singa.ConcatOn(singa.VecTensor([padding, x]), axis)
        else:
            padding = singa.SliceOn(x, n - pad, n, axis)
            x = singa.ConcatOn(singa.VecTensor([x, padding]), axis)
    return x


def handle_odd_pad_bwd(dx, odd_padding):
    """Drop the rows / columns that handle_odd_pad_fwd added."""
    top, bottom, left, right = [int(p) for p in odd_padding]
    shape = list(dx.shape())
    starts = [0, 0, top, left]
    ends = [shape[0], shape[1], shape[2] - bottom, shape[3] - right]
    return singa.SliceND(dx, starts, ends)


def same_pad_shape_check(handle, pad_mode, x):
    """Validate a handle's padding against SAME_* mode; returns the full
    (un-halved) padding."""
    _kernel = [handle.kernel_h, handle.kernel_w]
    _stride = [handle.stride_h, handle.stride_w]
    _padding = [handle.pad_h, handle.pad_w]
    _padding_correct, _ = get_padding_shape(pad_mode, x.shape()[2:], _kernel,
                                            _stride)
    _padding_crop = [_padding_correct[0], _padding_correct[2]]
    assert _padding == _padding_crop, (
        'For a same mode, the given padding %s is wrong, the correct one '
        'should be %s.' % (_padding, _padding_crop))
    return _padding_correct


def re_new_handle(handle, x, is_pool=False):
    """Build a handle with the same hyper-parameters for a new input shape."""
    kernel_size = [handle.kernel_h, handle.kernel_w]
    stride = [handle.stride_h, handle.stride_w]
    padding = [handle.pad_h, handle.pad_w]
    if is_pool:
        return singa.PoolingHandle(x, kernel_size, stride, padding,
                                   handle.is_max_pooling)
    return singa.ConvHandle(x, kernel_size, stride, padding, handle.channels,
                            handle.num_filters, handle.bias_term, handle.group)


def get_padding_shape(pad_mode, input_spatial_shape, kernel_spatial_shape,
                      strides_spatial):
    """(symmetric padding per side, odd remainder per side) for SAME_UPPER /
    SAME_LOWER; both lists are [h_begin, h_end, w_begin, w_end]."""
    output_spatial_shape = get_output_shape(pad_mode, input_spatial_shape,
                                            kernel_spatial_shape,
                                            strides_spatial)
    n = len(input_spatial_shape)
    pad_shape = [0] * n * 2
    odd_padd_shape = [0] * n * 2
    for i in range(n):
        whole_pad = (output_spatial_shape[i] - 1) * strides_spatial[i] + \
            kernel_spatial_shape[i] - input_spatial_shape[i]
        whole_pad = max(whole_pad, 0)
        pad_shape[2 * i] = pad_shape[2 * i + 1] = whole_pad // 2
        if whole_pad % 2 != 0:
            if pad_mode == "SAME_UPPER":
                odd_padd_shape[2 * i + 1] += 1
            else:
                odd_padd_shape[2 * i] += 1
    return pad_shape, tuple(odd_padd_shape)


def get_output_shape(auto_pad, input_spatial_shape, kernel_spatial_shape,
                     strides_spatial):
    """Spatial output shape of conv / pool under an ONNX auto_pad mode."""
    out_shape = [0] * len(input_spatial_shape)
    if auto_pad in ('SAME_UPPER', 'SAME_LOWER'):
        for i in range(len(input_spatial_shape)):
            out_shape[i] = int(
                np.ceil(float(input_spatial_shape[i]) /
                        float(strides_spatial[i])))
    elif auto_pad == 'VALID':
        for i in range(len(input_spatial_shape)):
            out_shape[i] = int(
                np.ceil(float(input_spatial_shape[i] -
                              (kernel_spatial_shape[i] - 1)) /
                        float(strides_spatial[i])))
    return out_shape


def force_unicode(s):
    try:
        return s.decode('utf-8')
    except AttributeError:
        return s


def post_order_recursive(root, root_t):
    """Post-order walk of the autograd tape ending at op ``root``.  Returns
    (ops, weights, inputs) as ordered dicts keyed by name: leaf tensors that
    were captured by an op (they store a gradient) are weights, the other
    leaves are graph inputs."""
    nodes = OrderedDict()
    weights = OrderedDict()
    inputs = OrderedDict()
    seen = set()

    def visit(op, t):
        if op is None:
            return
        if type(op).__name__ != 'Dummy':
            if id(op) in seen:
                return
            seen.add(id(op))
        for srcop, _, y, _ in op.src:
            visit(srcop, y)
        if type(op).__name__ == 'Dummy':
            if t is not None:
                weights[op.name] = t
            else:
                inputs[op.name] = t
        else:
            nodes[op.name] = op

    visit(root, root_t)
    return nodes, weights, inputs
