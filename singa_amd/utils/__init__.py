"""Helpers shared by layers, autograd and the ONNX bridge: odd / SAME padding
arithmetic, handle re-creation, tape traversal for export, progress bar.

API parity: reference python/singa/utils.py:27-267.  Odd padding is applied
with one embed / slice kernel per call instead of a chain of concatenations.
"""
import sys
from collections import OrderedDict

from .. import singa_wrap as singa


def update_progress(progress, info):
    """Text progress bar; progress in [0, 1], negative = halt, >= 1 = done."""
    width = 20
    note = ""
    if isinstance(progress, bool) or not isinstance(progress, (int, float)):
        progress, note = 0.0, "error: progress var must be float. "
    elif progress < 0:
        progress, note = 0.0, "Halt. "
    elif progress >= 1:
        progress, note = 1.0, "Done. "
    note += info
    filled = int(round(width * progress))
    sys.stdout.write("[%s] %3.1f%% %s" % (
        "." * filled + " " * (width - filled), progress * 100, note))
    sys.stdout.write('\b' * (9 + width + len(note)))
    sys.stdout.flush()


def _replicate_border(x, axis, count, at_start):
    """Prepend / append ``count`` copies' worth of border rows on ``axis``."""
    extent = list(x.shape())[axis]
    if at_start:
        parts = [singa.SliceOn(x, 0, count, axis), x]
    else:
        parts = [x, singa.SliceOn(x, extent - count, extent, axis)]
    return singa.ConcatOn(singa.VecTensor(parts), axis)


def handle_odd_pad_fwd(x, odd_padding, is_pool=False, is_max=False):
    """Apply the asymmetric part of a padding: (top, bottom, left, right)
    extra rows / columns.  Convolution pads with zeros (one embed kernel); max
    pooling with the most negative number, so a padded cell never wins and
    never receives gradient; average pooling replicates the border so the
    extra cells do not bias the mean."""
    top, bottom, left, right = (int(p) for p in odd_padding)
    if not is_pool or is_max:
        n, c, h, w = x.shape()
        return singa.EmbedND(x, [n, c, h + top + bottom, w + left + right],
                             [0, 0, top, left], [n, c, top + h, left + w],
                             None, -3.0e38 if is_pool else 0.0)
    for axis, at_start, count in ((2, True, top), (2, False, bottom),
                                  (3, True, left), (3, False, right)):
        if count:
            x = _replicate_border(x, axis, count, at_start)
    return x


def _fold_border(dx, axis, count, at_start):
    """Reverse of ``_replicate_border`` for gradients: cut the replicated
    cells off and add their gradient onto the border cells they copied."""
    shape = list(dx.shape())
    n = shape[axis]
    if at_start:
        extra, body = singa.SliceOn(dx, 0, count, axis), \
            singa.SliceOn(dx, count, n, axis)
        first = 0
    else:
        extra, body = singa.SliceOn(dx, n - count, n, axis), \
            singa.SliceOn(dx, 0, n - count, axis)
        first = n - 2 * count
    kept = list(body.shape())
    starts, ends = [0] * len(kept), list(kept)
    starts[axis], ends[axis] = first, first + count
    return singa.__add__(body, singa.EmbedND(extra, kept, starts, ends, None,
                                             0.0))


def handle_odd_pad_bwd(dx, odd_padding, is_pool=False, is_max=False):
    """Drop the rows / columns that handle_odd_pad_fwd added.  For average
    pooling the added cells were copies of border cells, so their gradient
    belongs to those cells."""
    top, bottom, left, right = (int(p) for p in odd_padding)
    if is_pool and not is_max:
        for axis, at_start, count in ((3, False, right), (3, True, left),
                                      (2, False, bottom), (2, True, top)):
            if count:
                dx = _fold_border(dx, axis, count, at_start)
        return dx
    n, c, h, w = dx.shape()
    return singa.SliceND(dx, [0, 0, top, left], [n, c, h - bottom, w - right])


def _window_of(handle):
    return ([handle.kernel_h, handle.kernel_w],
            [handle.stride_h, handle.stride_w], [handle.pad_h, handle.pad_w])


def same_pad_shape_check(handle, pad_mode, x):
    """Validate a handle's padding against SAME_* mode; returns the full
    (un-halved) padding."""
    kernel, stride, given = _window_of(handle)
    full, _ = get_padding_shape(pad_mode, x.shape()[2:], kernel, stride)
    wanted = [full[0], full[2]]
    assert given == wanted, (
        '%s needs padding %s for this input, the handle has %s' %
        (pad_mode, wanted, given))
    return full


def re_new_handle(handle, x, is_pool=False):
    """Build a handle with the same hyper-parameters for a new input shape."""
    kernel, stride, padding = _window_of(handle)
    if is_pool:
        return singa.PoolingHandle(x, kernel, stride, padding,
                                   handle.is_max_pooling)
    return singa.ConvHandle(x, kernel, stride, padding, handle.channels,
                            handle.num_filters, handle.bias_term, handle.group)


def get_padding_shape(pad_mode, input_spatial_shape, kernel_spatial_shape,
                      strides_spatial):
    """(symmetric padding per side, odd remainder per side) for SAME_UPPER /
    SAME_LOWER; both are [h_begin, h_end, w_begin, w_end]: the odd cell goes
    to the end under SAME_UPPER and to the beginning under SAME_LOWER."""
    outputs = get_output_shape(pad_mode, input_spatial_shape,
                               kernel_spatial_shape, strides_spatial)
    symmetric, odd = [], []
    for size, kernel, stride, out in zip(input_spatial_shape,
                                         kernel_spatial_shape,
                                         strides_spatial, outputs):
        total = max((out - 1) * stride + kernel - size, 0)
        symmetric += [total // 2] * 2
        extra = total % 2
        odd += [0, extra] if pad_mode == "SAME_UPPER" else [extra, 0]
    return symmetric, tuple(odd)


def get_output_shape(auto_pad, input_spatial_shape, kernel_spatial_shape,
                     strides_spatial):
    """Spatial output shape of conv / pool under an ONNX auto_pad mode (zeros
    for modes that carry explicit padding)."""
    dims = list(zip(input_spatial_shape, kernel_spatial_shape,
                    strides_spatial))
    if auto_pad in ('SAME_UPPER', 'SAME_LOWER'):
        return [-(-size // stride) for size, _, stride in dims]
    if auto_pad == 'VALID':
        return [-(-(size - kernel + 1) // stride)
                for size, kernel, stride in dims]
    return [0] * len(dims)


def force_unicode(s):
    return s.decode('utf-8') if isinstance(s, (bytes, bytearray)) else s


def post_order_recursive(root, root_t):
    """Post-order walk of the autograd tape ending at op ``root``.  Returns
    (ops, weights, inputs) as ordered dicts keyed by name: leaf tensors that
    were captured by an op (they store a gradient) are weights, the other
    leaves are graph inputs.  Iterative, so deep models do not hit the
    interpreter's recursion limit."""
    ops, weights, inputs = OrderedDict(), OrderedDict(), OrderedDict()
    seen = set()
    stack = [(root, root_t, False)]
    while stack:
        op, t, expanded = stack.pop()
        if op is None:
            continue
        leaf = type(op).__name__ == 'Dummy'
        if expanded:
            if not leaf:
                ops[op.name] = op
            elif t is not None:
                weights[op.name] = t
            else:
                # the leaf's own tensor: a fed input, or a constant the
                # computation captured (exported as an initializer)
                inputs[op.name] = op.__dict__.get("tensor")
            continue
        if not leaf:
            if id(op) in seen:
                continue
            seen.add(id(op))
        stack.append((op, t, True))
        for src_op, _, y, _ in reversed(op.src):
            stack.append((src_op, y, False))
    return ops, weights, inputs
