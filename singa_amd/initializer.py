"""Parameter initialisers operating in place on a Tensor.

API parity: reference python/singa/initializer.py:41-335.  ``eye`` and
``orthogonal`` go through numpy on the host and must not be called while a
graph is being recorded.  (The reference rejects 2-d tensors in ``eye`` /
``orthogonal`` because of an inverted test; here exactly 2-d tensors are
accepted, which is what its documentation states.)
"""
import math

import numpy as np


def eye(t):
    """Ones on the diagonal, zeros elsewhere (2-d tensors)."""
    if len(t.shape) != 2:
        raise ValueError("Only tensors with 2 dimensions are supported")
    t.copy_from(np.eye(t.shape[0], t.shape[1], dtype=np.float32))


def orthogonal(t, gain=1.0):
    """Random (semi-)orthogonal matrix scaled by ``gain``."""
    if len(t.shape) != 2:
        raise ValueError("Only tensors with 2 dimensions are supported")
    a = np.random.normal(0.0, 1.0, t.shape).astype(np.float32)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == tuple(t.shape) else v
    t.copy_from(np.ascontiguousarray(q * gain, dtype=np.float32))


def lecun_uniform(t):
    """U(-l, l), l = sqrt(3 / fan_in)."""
    _random_fill(t, scale=1., mode='fan_in', distribution='uniform')


def glorot_normal(t):
    """N(0, 2 / (fan_in + fan_out))."""
    _random_fill(t, scale=1., mode='fan_avg', distribution='normal')


def glorot_uniform(t):
    """U(-l, l), l = sqrt(6 / (fan_in + fan_out))."""
    _random_fill(t, scale=1., mode='fan_avg', distribution='uniform')


def he_normal(t):
    """N(0, 2 / fan_in)."""
    _random_fill(t, scale=2., mode='fan_in', distribution='normal')


def lecun_normal(t):
    """N(0, 1 / fan_in)."""
    _random_fill(t, scale=1., mode='fan_in', distribution='normal')


def he_uniform(t):
    """U(-l, l), l = sqrt(6 / fan_in)."""
    _random_fill(t, scale=2., mode='fan_in', distribution='uniform')


def gaussian(t, fan_in=0, fan_out=0):
    """Deprecated: N(0, std) with std from fan_in / fan_out (He et al.)."""
    assert fan_in > 0 or fan_out > 0, \
        'fan_in and fan_out cannot be 0 at the same time'
    avg = 2
    if fan_in * fan_out == 0:
        avg = 1
    std = math.sqrt(2.0 * avg / (fan_in + fan_out))
    t.gaussian(0, std)


def xavier(t):
    """Deprecated: U(-s, s), s = sqrt(6 / (rows + cols))."""
    scale = math.sqrt(6.0 / (t.shape[0] + t.shape[1]))
    t.uniform(-scale, scale)


def glorot(t):
    """Deprecated: N(0, 1) * sqrt(2 / (rows + cols))."""
    scale = math.sqrt(2.0 / (t.shape[0] + t.shape[1]))
    t.gaussian(0, 1)
    t *= scale


def msra(t):
    """Deprecated: N(0, sqrt(2 / rows))."""
    t.gaussian(0, math.sqrt(2.0 / t.shape[0]))


def _compute_fans(shape, data_format='channels_first'):
    """(fan_in, fan_out) of a dense (2-d) or convolution (3-5 d) weight."""
    if len(shape) == 2:
        fan_in, fan_out = shape[0], shape[1]
    elif len(shape) in {3, 4, 5}:
        if data_format == 'channels_first':
            rf = np.prod(shape[2:])
            fan_in, fan_out = shape[1] * rf, shape[0] * rf
        elif data_format == 'channels_last':
            rf = np.prod(shape[:-2])
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
        else:
            raise ValueError('Invalid data_format: ' + data_format)
    else:
        fan_in = fan_out = np.sqrt(np.prod(shape))
    return fan_in, fan_out


def _random_fill(t, scale, mode, distribution):
    """normal: stddev = sqrt(scale / n); uniform: limit = sqrt(3 scale / n),
    with n = fan_in, fan_out or their mean."""
    if scale <= 0.:
        raise ValueError('`scale` must be a positive float. Got:', scale)
    mode = mode.lower()
    if mode not in {'fan_in', 'fan_out', 'fan_avg'}:
        raise ValueError('Invalid `mode` argument: expected on of '
                         '{"fan_in", "fan_out", "fan_avg"} but got', mode)
    distribution = distribution.lower()
    if distribution not in {'normal', 'uniform'}:
        raise ValueError('Invalid `distribution` argument: expected one of '
                         '{"normal", "uniform"} but got', distribution)
    fan_in, fan_out = _compute_fans(t.shape)
    if mode == 'fan_in':
        scale /= max(1., fan_in)
    elif mode == 'fan_out':
        scale /= max(1., fan_out)
    else:
        scale /= max(1., float(fan_in + fan_out) / 2)
    if distribution == 'normal':
        t.gaussian(0., float(np.sqrt(scale)))
    else:
        limit = float(np.sqrt(3. * scale))
        t.uniform(-limit, limit)
