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


class _LSTMSequence(autograd.Operator):
    """A whole unrolled LSTM layer as ONE operator.

    Inputs: x_0 .. x_{T-1} (batch, features), h0, c0 and the 16 parameters in
    the order Wx_{i,f,o,g} (features, H), Wh_{i,f,o,g} (H, H), Bx_{i,f,o,g},
    Bh_{i,f,o,g} (H,).  Outputs: h_0 .. h_{T-1} and the final cell state.

    The steps are stacked into one (T, batch, features) tensor, the parameters
    are laid out as the flat weight block of a one-layer recurrent stack, and
    the sequence runs through the same code as ``CudnnRNN``: one GEMM for the
    input projection of all steps and, on the GPU, the single-launch LSTM
    kernels.  Two stacking kernels and 16 small parameter copies replace
    ~5 graph nodes per time step in each direction."""

    def __init__(self, steps, handles):
        super(_LSTMSequence, self).__init__()
        self.steps = steps
        self.handles = handles          # (T, B, features, H) -> handle

    def _handle(self, x0, H):
        from ..core import rnn_ops
        B, n_in = x0.shape()
        key = (self.steps, B, n_in, H, id(x0.device()))
        if key not in self.handles:
            probe = CTensor([self.steps, B, n_in], x0.device(), x0.data_type())
            self.handles[key] = rnn_ops.CudnnRNNHandle(probe, H, mode=2)
        return self.handles[key]

    @staticmethod
    def _blocks(handle, flat):
        """Views of the flat weight / gradient block: input matrix (4H, in),
        recurrent matrix (4H, H) and the two bias vectors (4H,)."""
        H = handle.hidden_size
        w_off, n_in, r_off, bw_off, br_off = handle._blocks[0]
        return (flat[w_off:w_off + 4 * H * n_in].view(4 * H, n_in),
                flat[r_off:r_off + 4 * H * H].view(4 * H, H),
                flat[bw_off:bw_off + 4 * H], flat[br_off:br_off + 4 * H])

    def forward(self, *ins):
        from ..core import rnn_ops
        T = self.steps
        ins = [t if t.is_contiguous() else singa.Contiguous(t) for t in ins]
        xs, params = ins[:T], ins[T + 2:]
        assert len(params) == 16, "expects 4 x (Wx, Wh, Bx, Bh) parameters"
        dev = xs[0].device()
        H = params[4].shape()[0]
        B = xs[0].shape()[0]
        handle = self._handle(xs[0], H)
        dt = xs[0].data_type()
        outs = [CTensor([B, H], dev, dt) for _ in range(T)]
        c_last = CTensor([B, H], dev, dt)
        training = autograd.training
        if getattr(handle, "_bar", None) is None:
            import torch
            handle._bar = torch.zeros(2, dtype=torch.int32,
                                      device=dev.torch_device)

        def body(*ts):
            import torch
            m = dev.math
            x_v = [t.view() for t in ts[:T]]
            h0, c0 = ts[T].view(), ts[T + 1].view()
            p_v = [t.view() for t in ts[T + 2:T + 18]]
            o_v = [t.view() for t in ts[T + 18:T + 18 + T]]
            n_in = x_v[0].shape[1]
            X = torch.empty((T, B, n_in), dtype=torch.float32,
                            device=dev.torch_device)
            m.pack(m.pack_plan(x_v), x_v, X.view(-1))
            W = torch.empty(handle.weights_size, dtype=torch.float32,
                            device=dev.torch_device)
            Wm, R, bW, bR = self._blocks(handle, W)
            for s in range(4):
                rows = slice(s * H, (s + 1) * H)
                m.copy(p_v[s].t(), Wm[rows])
                m.copy(p_v[4 + s].t(), R[rows])
                m.copy(p_v[8 + s], bW[rows])
                m.copy(p_v[12 + s], bR[rows])
            Y, _, cy = rnn_ops._forward(handle, X, h0.reshape(1, B, H),
                                        c0.reshape(1, B, H), W, None, training)
            m.unpack(m.pack_plan(o_v), Y.view(-1), o_v, 1.0)
            m.copy(cy[0], ts[-1].view())
            handle._flat_weights = W

        _submit(dev, "LSTMSequenceForward", body, ins, outs + [c_last])
        if training:
            self.cache = (handle, params, [x.shape() for x in xs], dt)
        return tuple(outs) + (c_last,)

    def backward(self, *dys):
        from ..core import rnn_ops
        T = self.steps
        handle, params, x_shapes, dt = self.cache
        dev = params[0].device()
        H, B = handle.hidden_size, handle.batch_size

        def given(dy):
            if dy is None:
                dy = CTensor([B, H], dev, dt)
                dy.SetFloatValue(0.0)
            return dy if dy.is_contiguous() else singa.Contiguous(dy)

        dys = [given(d) for d in dys]
        want_dx = any(self.x_needs_grad[i] for i in range(T))
        dxs = [CTensor(list(s), dev, dt) for s in x_shapes] if want_dx else []
        dh0, dc0 = CTensor([B, H], dev, dt), CTensor([B, H], dev, dt)
        dparams = [CTensor(list(p.shape()), dev, p.data_type())
                   for p in params]

        def body(*ts):
            import torch
            m = dev.math
            dh_v = [t.view() for t in ts[:T]]
            dc_last = ts[T].view()
            outs = [t.view() for t in ts[T + 1 + 16:]]
            dx_v, rest = outs[:len(dxs)], outs[len(dxs):]
            dY = torch.empty((T, B, H), dtype=torch.float32,
                             device=dev.torch_device)
            m.pack(m.pack_plan(dh_v), dh_v, dY.view(-1))
            zero = torch.empty((1, B, H), dtype=torch.float32,
                               device=dev.torch_device)
            m.fill(zero, 0.0)
            W = handle._flat_weights
            dW = torch.empty_like(W)
            dX, dhx, dcx = rnn_ops._backward(handle, dY, zero,
                                             dc_last.reshape(1, B, H), W, dW)
            if dx_v:
                m.unpack(m.pack_plan(dx_v), dX.view(-1), dx_v, 1.0)
            m.copy(dhx[0], rest[0])
            m.copy(dcx[0], rest[1])
            gW, gR, gbW, gbR = self._blocks(handle, dW)
            grads = rest[2:]
            for s in range(4):
                rows = slice(s * H, (s + 1) * H)
                m.copy(gW[rows].t(), grads[s])
                m.copy(gR[rows].t(), grads[4 + s])
                m.copy(gbW[rows], grads[8 + s])
                m.copy(gbR[rows], grads[12 + s])

        _submit(dev, "LSTMSequenceBackward", body, dys + list(params),
                dxs + [dh0, dc0] + dparams)
        dx_out = dxs if want_dx else [None] * T
        return tuple(dx_out) + (dh0, dc0) + tuple(dparams)


def lstm_sequence(xs, h0, c0, params, handles):
    """Run an unrolled LSTM layer over the step inputs ``xs``; ``params`` are
    the 16 parameter tensors (see ``_LSTMSequence``), ``handles`` a dict the
    caller keeps for the per-shape geometry.  Returns ([h_t], c_T)."""
    outs = _LSTMSequence(len(xs), handles)(*xs, h0, c0, *params)
    return list(outs[:-1]), outs[-1]
