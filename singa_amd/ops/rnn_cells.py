"""Fused recurrent cell operators (autograd level).

``lstm_cell`` takes the pre-activation gates (batch, 4*H), ordered i, f, o, g,
and the previous cell state and produces (h, c) with one point-wise kernel in
each direction, replacing the ~20 element-wise ops of the composed cell
(reference python/singa/layer.py:1330-1368).
"""
from .. import autograd
from .. import singa_wrap as singa
from ..core.tensor_core import Tensor as CTensor, _submit


class _LSTMCell(autograd.Operator):

    def forward(self, gates, c_prev):
        dev = gates.device()
        batch, four_h = gates.shape()
        H = four_h // 4
        c = CTensor([batch, H], dev, gates.data_type())
        h = CTensor([batch, H], dev, gates.data_type())
        acts = CTensor([batch, four_h], dev, gates.data_type())
        g = gates if gates.is_contiguous() else singa.Contiguous(gates)
        cp = c_prev if c_prev.is_contiguous() else singa.Contiguous(c_prev)
        _submit(dev, "LSTMCellForward",
                lambda a, b, cc, hh, ac: a._dev.math.lstm_cell_fwd(
                    a.view(), b.view(), cc.view(), hh.view(), ac.view()),
                [g, cp], [c, h, acts])
        if autograd.training:
            self.cache = (acts, cp, c)
        return h, c

    def backward(self, dh, dc):
        acts, c_prev, c = self.cache
        dev = acts.device()
        if dh is None:
            dh = CTensor(c.shape(), dev, c.data_type())
            dh.SetFloatValue(0.0)
        if dc is None:
            dc = CTensor(c.shape(), dev, c.data_type())
            dc.SetFloatValue(0.0)
        dgates = CTensor(acts.shape(), dev, acts.data_type())
        dc_prev = CTensor(c.shape(), dev, c.data_type())
        dhc = dh if dh.is_contiguous() else singa.Contiguous(dh)
        dcc = dc if dc.is_contiguous() else singa.Contiguous(dc)
        _submit(dev, "LSTMCellBackward",
                lambda a, b, ac, cp, cc, dg, dcp: a._dev.math.lstm_cell_bwd(
                    a.view(), b.view(), ac.view(), cp.view(), cc.view(),
                    dg.view(), dcp.view()),
                [dhc, dcc, acts, c_prev, c], [dgates, dc_prev])
        return dgates, dc_prev


def lstm_cell(gates, c_prev):
    """(h, c) of one LSTM step from gate pre-activations and the old cell."""
    h, c = _LSTMCell()(gates, c_prev)
    return h, c
