"""Parameter initialisers operating in place on a Tensor.

Every variance-scaling scheme is one row of ``_SCHEMES`` (gain, which fan to
divide by, distribution) applied by ``variance_scaling``; the public functions
are generated from that table.

API parity: reference python/singa/initializer.py:41-335.  ``eye`` and
``orthogonal`` go through numpy on the host and must not be called while a
graph is being recorded.  (The reference rejects 2-d tensors in ``eye`` /
``orthogonal`` because of an inverted test; here exactly 2-d tensors are
accepted, which is what its documentation states.)
"""
import math

import numpy as np


def _matrix_only(t):
    if len(t.shape) != 2:
        raise ValueError("expected a 2-d tensor, got shape %s" % (t.shape,))


def eye(t):
    """Ones on the diagonal, zeros elsewhere (2-d tensors)."""
    _matrix_only(t)
    t.copy_from(np.eye(*t.shape, dtype=np.float32))


def orthogonal(t, gain=1.0):
    """Random (semi-)orthogonal matrix scaled by ``gain``: the orthonormal
    factor of a Gaussian matrix's SVD that has the tensor's shape."""
    _matrix_only(t)
    u, _, vt = np.linalg.svd(np.random.normal(size=t.shape), full_matrices=False)
    q = u if u.shape == tuple(t.shape) else vt
    t.copy_from(np.ascontiguousarray(gain * q, dtype=np.float32))


def _compute_fans(shape, data_format='channels_first'):
    """(fan_in, fan_out) of a dense (2-d) or convolution (3-5 d) weight; any
    other rank counts sqrt(volume) for both."""
    rank = len(shape)
    if rank == 2:
        return shape[0], shape[1]
    if rank not in (3, 4, 5):
        side = math.sqrt(math.prod(shape))
        return side, side
    if data_format == 'channels_first':
        field = math.prod(shape[2:])
        return shape[1] * field, shape[0] * field
    if data_format == 'channels_last':
        field = math.prod(shape[:-2])
        return shape[-2] * field, shape[-1] * field
    raise ValueError('unknown data_format %r' % (data_format,))


def variance_scaling(t, gain, fan, distribution):
    """Fill ``t`` with variance gain / n, n = fan_in, fan_out or their mean
    (``fan`` = 'fan_in' | 'fan_out' | 'fan_avg'): a normal with that variance,
    or the uniform distribution on +-sqrt(3 gain / n)."""
    if gain <= 0.:
        raise ValueError('gain must be positive, got %r' % (gain,))
    fan_in, fan_out = _compute_fans(t.shape)
    try:
        n = {'fan_in': fan_in, 'fan_out': fan_out,
             'fan_avg': (fan_in + fan_out) / 2.0}[fan.lower()]
    except KeyError:
        raise ValueError("fan is 'fan_in', 'fan_out' or 'fan_avg', got %r" %
                         (fan,)) from None
    variance = gain / max(1., n)
    kind = distribution.lower()
    if kind == 'normal':
        t.gaussian(0., math.sqrt(variance))
    elif kind == 'uniform':
        bound = math.sqrt(3. * variance)
        t.uniform(-bound, bound)
    else:
        raise ValueError("distribution is 'normal' or 'uniform', got %r" %
                         (distribution,))


# kept under its earlier private name
_random_fill = variance_scaling

_SCHEMES = {
    # name: (gain, fan, distribution, summary)
    'lecun_uniform': (1., 'fan_in', 'uniform', "U(-l, l), l = sqrt(3 / fan_in)"),
    'lecun_normal': (1., 'fan_in', 'normal', "N(0, 1 / fan_in)"),
    'glorot_uniform': (1., 'fan_avg', 'uniform',
                       "U(-l, l), l = sqrt(6 / (fan_in + fan_out))"),
    'glorot_normal': (1., 'fan_avg', 'normal', "N(0, 2 / (fan_in + fan_out))"),
    'he_uniform': (2., 'fan_in', 'uniform', "U(-l, l), l = sqrt(6 / fan_in)"),
    'he_normal': (2., 'fan_in', 'normal', "N(0, 2 / fan_in)"),
}


def _scheme(name):
    gain, fan, distribution, summary = _SCHEMES[name]

    def init(t):
        variance_scaling(t, gain, fan, distribution)
    init.__name__ = name
    init.__doc__ = summary
    return init


for _name in _SCHEMES:
    globals()[_name] = _scheme(_name)


# ---- deprecated spellings kept for old scripts ---------------------------------------
def gaussian(t, fan_in=0, fan_out=0):
    """N(0, std) with std = sqrt(2 / fan) when one fan is given and
    sqrt(4 / (fan_in + fan_out)) when both are."""
    assert fan_in > 0 or fan_out > 0, 'give fan_in, fan_out or both'
    fans_given = 2 if fan_in * fan_out else 1
    t.gaussian(0, math.sqrt(2.0 * fans_given / (fan_in + fan_out)))


def xavier(t):
    """U(-s, s), s = sqrt(6 / (rows + cols))."""
    bound = math.sqrt(6.0 / sum(t.shape[:2]))
    t.uniform(-bound, bound)


def glorot(t):
    """N(0, 1) * sqrt(2 / (rows + cols))."""
    t.gaussian(0, 1)
    t *= math.sqrt(2.0 / sum(t.shape[:2]))


def msra(t):
    """N(0, sqrt(2 / rows))."""
    t.gaussian(0, math.sqrt(2.0 / t.shape[0]))
