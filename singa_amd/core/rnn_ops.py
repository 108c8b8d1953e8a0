"""Multi-layer (bi)directional recurrent networks over one flat weight tensor:
ReLU / tanh RNN, LSTM and GRU, with optional per-sample sequence lengths.

Takes the place of the reference's cuDNN RNN wrapper (src/model/operation/
rnn.h:38-131, rnn.cc:27-806: ``CudnnRNNHandle``, ``GpuRNNForwardTraining /
Inference``, ``GpuRNNBackwardx / BackwardW``, the ``...Ex`` variants with
``seq_lengths``, ``GpuRNNSetParam / GetParamCopy``).  The algorithm is laid
out for large GEMMs on the matrix cores:

* the input projection of a whole layer is ONE GEMM over all time steps,
  (T*B, in) x (in, G*H); only the recurrent (B, H) x (H, G*H) product and
  the fused point-wise cell kernel run per time step;
* backward computes the weight gradients of a layer with two GEMMs over all
  time steps (dW = dG^T X, dR = dG^T H_prev) and the input gradient with one;
* everything a pass needs is launched from one graph node, so under hipGraph
  replay a sequence costs no host work per step;
* each step is one recurrent GEMM plus ONE fused cell kernel (LSTM, GRU,
  tanh / relu) in either direction;
* a float32 LSTM without ``seq_lengths`` whose hidden size is a multiple of
  32 up to 256 (batch <= 128) runs the whole time loop of a layer and
  direction in ONE persistent launch, forward and backward.

Weight layout (element offsets are published through ``weights_mapping`` with
cuDNN's numbering: ``(linLayerID, pseudoLayer, is_bias) -> (offset, size)``,
linLayerID < G input matrices, >= G recurrent ones; LSTM gates i, f, g, o;
GRU gates r, z, n): for every pseudo layer (layer * directions + direction)
the input matrix block (G*H, in) then the recurrent block (G*H, H); after all
matrices, for every pseudo layer the two bias blocks (G*H each).
"""
import os

import numpy as np
import torch

from . import dtypes
from .tensor_core import Tensor, _submit

_GATES = {0: 1, 1: 1, 2: 4, 3: 3}
# position of cuDNN's gate number inside a block (the LSTM cell kernel wants
# i, f, o, g)
_SLOT = {0: [0], 1: [0], 2: [0, 1, 3, 2], 3: [0, 1, 2]}


class CudnnRNNHandle(object):
    """Geometry, weight layout and saved activations of one recurrent stack.
    ``x`` is a (seq, batch, feature) tensor used for its shape and device."""

    def __init__(self, x, hidden_size, mode=0, num_layers=1, bias=1,
                 dropout=0.0, bidirectional=0):
        assert mode in _GATES, "mode: 0 relu, 1 tanh, 2 lstm, 3 gru"
        assert x.nDim() == 3, "x must be (seq, batch, feature)"
        self.dev = x.device()
        self.mode = int(mode)
        self.bias = int(bias)
        self.dropout = float(dropout)
        self.bidirectional = 1 if bidirectional else 0
        self.hidden_size = int(hidden_size)
        self.num_layers = int(num_layers)
        self.seq_length, self.batch_size, self.feature_size = x.shape()
        self.batch_first = 0
        self.gates = _GATES[self.mode]
        self.directions = 2 if self.bidirectional else 1
        H, G, D = self.hidden_size, self.gates, self.directions
        self.weights_mapping = {}
        self._blocks = []  # per pseudo layer: (w_off, in, r_off, bw_off, br_off)
        off = 0
        mats = []
        for layer in range(self.num_layers):
            n_in = self.feature_size if layer == 0 else H * D
            for d in range(D):
                w_off = off
                off += G * H * n_in
                r_off = off
                off += G * H * H
                mats.append((w_off, n_in, r_off))
        for p, (w_off, n_in, r_off) in enumerate(mats):
            bw_off = off
            off += G * H
            br_off = off
            off += G * H
            self._blocks.append((w_off, n_in, r_off, bw_off, br_off))
            for lin in range(G):
                s = _SLOT[self.mode][lin]
                self.weights_mapping[(lin, p, False)] = (
                    w_off + s * H * n_in, H * n_in)
                self.weights_mapping[(lin + G, p, False)] = (
                    r_off + s * H * H, H * H)
                self.weights_mapping[(lin, p, True)] = (bw_off + s * H, H)
                self.weights_mapping[(lin + G, p, True)] = (br_off + s * H, H)
        self.weights_size = off
        self.weights_size_bytes = off * 4
        self.reserve = None  # activations of the last training forward


# ---------------------------------------------------------------------------
# tiny expression helpers over a device's math backend (torch views in / out)
# ---------------------------------------------------------------------------
class _X(object):

    def __init__(self, dev):
        self.m = dev.math
        self.tdev = dev.torch_device

    def new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.tdev)

    def zeros(self, *shape):
        t = self.new(*shape)
        self.m.fill(t, 0.0)
        return t

    def bin(self, op, a, b, out=None):
        if out is None:
            out = self.new(*torch.broadcast_shapes(a.shape, b.shape))
        self.m.binary(op, a, b, out)
        return out

    def un(self, op, a, out=None):
        if out is None:
            out = self.new(*a.shape)
        self.m.unary(op, a, out)
        return out

    def sc(self, op, a, s, out=None):
        if out is None:
            out = self.new(*a.shape)
        self.m.binary_scalar(op, a, s, out)
        return out

    def copy(self, src, dst):
        self.m.copy(src, dst)
        return dst

    def mm(self, a, b, c, beta=0.0):
        self.m.gemm(1.0, a, b, beta, c)
        return c

    def colsum(self, a2d, out):
        self.m.sum_axis(a2d, 0, out)
        return out


def _weights(h, W):
    """Views of the flat weight tensor, per pseudo layer."""
    H, G = h.hidden_size, h.gates
    out = []
    for (w_off, n_in, r_off, bw_off, br_off) in h._blocks:
        out.append((W[w_off:w_off + G * H * n_in].view(G * H, n_in),
                    W[r_off:r_off + G * H * H].view(G * H, H),
                    W[bw_off:bw_off + G * H], W[br_off:br_off + G * H]))
    return out


def _mask(X_, h, lens):
    """(T, B, 1) float mask, 1 where step t is inside sample b's sequence."""
    if lens is None:
        return None
    T, B = h.seq_length, h.batch_size
    steps = h._steps  # (T, 1) step numbers, made when the op was recorded
    lf = X_.new(1, B)
    X_.m.copy(lens.reshape(1, B), lf)
    return X_.bin("lt", steps, lf).view(T, B, 1)


def _cell_fwd(X_, h, gx_t, hp, cp, R, bR, save, t):
    """One time step.  gx_t: (B, G*H) input projection (+ biases); returns
    (h_new, c_new)."""
    H, mode = h.hidden_size, h.mode
    if mode == 2:
        X_.mm(hp, R.t(), gx_t, 1.0)
        c_new, h_new = X_.new(*hp.shape), X_.new(*hp.shape)
        X_.m.lstm_cell_fwd(gx_t, cp, c_new, h_new, save["acts"][t])
        return h_new, c_new
    if mode in (0, 1):
        X_.mm(hp, R.t(), gx_t, 1.0)
        h_new = X_.un("relu" if mode == 0 else "tanh", gx_t)
        return h_new, None
    # GRU: n = tanh(Wn x + bWn + r * (Rn h + bRn)), one fused cell kernel
    gh = X_.new(hp.shape[0], 3 * H)
    X_.mm(hp, R.t(), gh)
    h_new = X_.new(*hp.shape)
    X_.m.gru_cell_fwd(gx_t, gh, bR[2 * H:] if h.bias else None, hp, h_new,
                      save["acts"][t], save["ghn"][t])
    return h_new, None


def _contig(X_, t):
    return t if t.is_contiguous() else X_.copy(t, X_.new(*t.shape))


def _whole_sequence(X_, h, mask, dtype):
    """True when the single-launch LSTM kernels cover this configuration
    (``SINGA_LSTM_SEQ=0`` forces the per-step path)."""
    return h.mode == 2 and mask is None and \
        os.environ.get("SINGA_LSTM_SEQ", "1") != "0" and \
        X_.m.lstm_seq_supported(h.batch_size, h.hidden_size, dtype)


def _forward(h, x, hx, cx, W, lens, training):
    X_ = _X(h.dev)
    T, B, H, G, D = (h.seq_length, h.batch_size, h.hidden_size, h.gates,
                     h.directions)
    assert tuple(x.shape) == (T, B, h.feature_size), \
        "input shape %s does not match the handle" % (tuple(x.shape),)
    wts = _weights(h, W)
    mask = _mask(X_, h, lens)
    res = {"layers": [], "mask": mask, "drop": []}
    hy = X_.new(h.num_layers * D, B, H)
    cy = X_.new(h.num_layers * D, B, H) if h.mode == 2 else None
    inp = x if x.is_contiguous() else X_.copy(x, X_.new(*x.shape))
    for layer in range(h.num_layers):
        n_in = inp.shape[2]
        Y = X_.new(T, B, D * H)
        per_dir = []
        for d in range(D):
            p = layer * D + d
            Wm, R, bW, bR = wts[p]
            gx = X_.new(T, B, G * H)
            X_.mm(inp.view(T * B, n_in), Wm.t(), gx.view(T * B, G * H))
            if h.bias:
                if h.mode == 3:
                    # bRn stays inside the reset-gated term
                    X_.bin("add", gx, bW.view(1, 1, G * H), gx)
                    X_.bin("add", gx[:, :, :2 * H], bR[:2 * H].view(1, 1, -1),
                           gx[:, :, :2 * H])
                else:
                    X_.bin("add", gx, X_.bin("add", bW, bR).view(1, 1, G * H),
                           gx)
            save = {"acts": X_.new(T, B, G * H) if h.mode >= 2 else None,
                    "ghn": X_.new(T, B, H) if h.mode == 3 else None,
                    "hseq": X_.new(T, B, H),
                    "cseq": X_.new(T, B, H) if h.mode == 2 else None,
                    "hnew": X_.new(T, B, H) if mask is not None else None,
                    "cnew": X_.new(T, B, H)
                    if (mask is not None and h.mode == 2) else None,
                    "h0": hx[p], "c0": cx[p] if h.mode == 2 else None,
                    "inp": inp, "gx": gx if h.mode in (0, 1) else None}
            hp = hx[p]
            cp = cx[p] if h.mode == 2 else None
            order = range(T) if d == 0 else range(T - 1, -1, -1)
            if _whole_sequence(X_, h, mask, gx.dtype):
                save["h0"], save["c0"] = _contig(X_, hp), _contig(X_, cp)
                save["seq"] = True
                X_.m.lstm_seq_fwd(gx, _contig(X_, R), save["h0"], save["c0"],
                                  save["hseq"], save["cseq"], save["acts"],
                                  h._bar, d == 1)
                last = order[-1]
                hp, cp = save["hseq"][last], save["cseq"][last]
                order = ()
            for t in order:
                h_new, c_new = _cell_fwd(X_, h, gx[t], hp, cp, R, bR, save, t)
                if mask is not None:
                    m = mask[t]
                    X_.copy(h_new, save["hnew"][t])
                    hs = X_.bin("add", hp, X_.bin(
                        "mul", m, X_.bin("sub", h_new, hp)), save["hseq"][t])
                    X_.bin("mul", m, h_new, Y[t, :, d * H:(d + 1) * H])
                    if c_new is not None:
                        X_.copy(c_new, save["cnew"][t])
                        cp = X_.bin("add", cp, X_.bin(
                            "mul", m, X_.bin("sub", c_new, cp)),
                            save["cseq"][t])
                    hp = hs
                else:
                    hp = X_.copy(h_new, save["hseq"][t])
                    if c_new is not None:
                        cp = X_.copy(c_new, save["cseq"][t])
            if mask is None:
                X_.copy(save["hseq"], Y[:, :, d * H:(d + 1) * H])
            X_.copy(hp, hy[p])
            if cy is not None:
                X_.copy(cp, cy[p])
            per_dir.append(save)
        res["layers"].append(per_dir)
        inp = Y
        if training and h.dropout > 0 and layer < h.num_layers - 1:
            dm = X_.new(*Y.shape)
            X_.m.bernoulli(dm, 1.0 - h.dropout)
            X_.sc("mul", dm, 1.0 / (1.0 - h.dropout), dm)
            inp = X_.bin("mul", Y, dm)
            res["drop"].append(dm)
        else:
            res["drop"].append(None)
    if training:
        h.reserve = res
    return Y, hy, cy


def _cell_bwd(X_, h, dh_new, dc_new, save, t, hp, cp, dgx_t, dgh_t):
    """Gradients of one step, one fused kernel: fills the gate-gradient rows
    ``dgx_t`` (and ``dgh_t`` for GRU, whose recurrent projection sees a
    different gradient for the n gate); returns (dh_prev from the point-wise
    part or None, dc_prev or None)."""
    mode = h.mode
    if mode == 2:
        c_t = save["cnew"][t] if save["cnew"] is not None else save["cseq"][t]
        dcp = X_.new(*hp.shape)
        X_.m.lstm_cell_bwd(dh_new, dc_new, save["acts"][t], cp, c_t, dgx_t, dcp)
        return None, dcp
    if mode in (0, 1):
        h_new = save["hnew"][t] if save["hnew"] is not None \
            else save["hseq"][t]
        X_.m.rnn_act_bwd(mode, dh_new, h_new, dgx_t)
        return None, None
    dhp = X_.new(*hp.shape)
    X_.m.gru_cell_bwd(dh_new, save["acts"][t], save["ghn"][t], hp, dgx_t,
                      dgh_t, dhp)
    return dhp, None


def _backward(h, dy, dhy, dcy, W, dW):
    X_ = _X(h.dev)
    res = h.reserve
    assert res is not None, "RNN backward needs a training forward first"
    T, B, H, G, D = (h.seq_length, h.batch_size, h.hidden_size, h.gates,
                     h.directions)
    wts = _weights(h, W)
    gwts = _weights(h, dW)
    mask = res["mask"]
    X_.m.fill(dW, 0.0)
    dhx = X_.new(h.num_layers * D, B, H)
    dcx = X_.new(h.num_layers * D, B, H) if h.mode == 2 else None
    dY = dy if dy.is_contiguous() else X_.copy(dy, X_.new(*dy.shape))
    for layer in range(h.num_layers - 1, -1, -1):
        if res["drop"][layer] is not None:
            dY = X_.bin("mul", dY, res["drop"][layer])
        n_in = res["layers"][layer][0]["inp"].shape[2]
        dX = X_.new(T, B, n_in)
        for d in range(D):
            p = layer * D + d
            save = res["layers"][layer][d]
            Wm, R, bW, bR = wts[p]
            gW, gR, gbW, gbR = gwts[p]
            dYd = X_.copy(dY[:, :, d * H:(d + 1) * H], X_.new(T, B, H))
            dGx = X_.new(T, B, G * H)
            dGh = X_.new(T, B, G * H) if h.mode == 3 else dGx
            dh_state = X_.copy(dhy[p], X_.new(B, H))
            dc_state = X_.copy(dcy[p], X_.new(B, H)) if h.mode == 2 else None
            order = list(range(T)) if d == 0 else list(range(T - 1, -1, -1))
            steps = range(T - 1, -1, -1)
            if save.get("seq"):
                X_.m.lstm_seq_bwd(_contig(X_, R), save["c0"], save["cseq"],
                                  save["acts"], dY[:, :, d * H:(d + 1) * H],
                                  _contig(X_, dhy[p]), _contig(X_, dcy[p]),
                                  dGx, dhx[p], dcx[p], h._bar, d == 1)
                steps = ()
            for i in steps:
                t = order[i]
                t_prev = order[i - 1] if i > 0 else None
                hp = save["hseq"][t_prev] if t_prev is not None else save["h0"]
                cp = None
                if h.mode == 2:
                    cp = save["cseq"][t_prev] if t_prev is not None \
                        else save["c0"]
                if mask is not None:
                    m = mask[t]
                    dh_new = X_.bin("mul", m, X_.bin("add", dh_state, dYd[t]))
                    dc_new = X_.bin("mul", m, dc_state) \
                        if dc_state is not None else None
                else:
                    dh_new = X_.bin("add", dh_state, dYd[t])
                    dc_new = dc_state
                dhp_pt, dcp = _cell_bwd(X_, h, dh_new, dc_new, save, t, hp, cp,
                                        dGx[t], dGh[t])
                dh_prev = X_.new(B, H)
                X_.mm(dGh[t], R, dh_prev)
                if dhp_pt is not None:
                    X_.bin("add", dh_prev, dhp_pt, dh_prev)
                if mask is not None:
                    keep = X_.sc("add", X_.sc("mul", m, -1.0), 1.0)  # 1 - m
                    X_.bin("add", dh_prev, X_.bin("mul", keep, dh_state),
                           dh_prev)
                    if dcp is not None:
                        X_.bin("add", dcp, X_.bin("mul", keep, dc_state), dcp)
                dh_state = dh_prev
                if dcp is not None:
                    dc_state = dcp
            if not save.get("seq"):
                X_.copy(dh_state, dhx[p])
                if dcx is not None:
                    X_.copy(dc_state, dcx[p])
            inp = save["inp"]
            g2 = dGx.view(T * B, G * H)
            X_.mm(g2.t(), inp.view(T * B, n_in), gW)
            # dR = sum_t dG_h[t]^T h_prev[t]
            gh2 = dGh
            if d == 0:
                X_.mm(gh2[0].t(), save["h0"], gR)
                if T > 1:
                    X_.mm(gh2[1:].reshape(-1, G * H).t(),
                          save["hseq"][:-1].reshape(-1, H), gR, 1.0)
            else:
                X_.mm(gh2[T - 1].t(), save["h0"], gR)
                if T > 1:
                    X_.mm(gh2[:-1].reshape(-1, G * H).t(),
                          save["hseq"][1:].reshape(-1, H), gR, 1.0)
            if h.bias:
                X_.colsum(g2, gbW)
                X_.colsum(dGh.view(T * B, G * H), gbR)
            X_.mm(g2, Wm, dX.view(T * B, n_in), 0.0 if d == 0 else 1.0)
        dY = dX
    return dY, dhx, dcx


# ---------------------------------------------------------------------------
# public functions (reference rnn.h:95-131)
# ---------------------------------------------------------------------------
def _states(h, hx, cx):
    shape = [h.num_layers * h.directions, h.batch_size, h.hidden_size]
    assert list(hx.shape()) == shape, "hx must be %s" % shape
    return shape


def _run_forward(x, hx, cx, W, h, lens, training):
    shape = _states(h, hx, cx)
    dev = x.device()
    y = Tensor([h.seq_length, h.batch_size, h.directions * h.hidden_size], dev,
               x.data_type())
    hy = Tensor(shape, dev, x.data_type())
    cy = Tensor(shape, dev, x.data_type())
    reads = [x, hx, cx, W] + ([lens] if lens is not None else [])
    if getattr(h, "_bar", None) is None:
        # barrier words of the single-launch LSTM kernels: arrival counter,
        # give-up flag of the last launch, and a sticky give-up flag that no
        # launch clears.  A wait that hit its bound means the workgroups were
        # not all resident (e.g. another process shares the GPU) and the
        # sequence was computed from incomplete data: the device raises at the
        # next read of device memory by the host.
        h._bar = torch.zeros(4, dtype=torch.int32, device=dev.torch_device)
        if hasattr(dev, "watch_flag"):
            dev.watch_flag(
                h._bar[2:3],
                "a persistent LSTM kernel gave up waiting at its grid barrier "
                "(its workgroups were not all resident); results since the "
                "last check are invalid.  SINGA_LSTM_SEQ=0 selects the "
                "per-step kernels.")
    if lens is not None and getattr(h, "_steps", None) is None:
        h._steps = torch.arange(h.seq_length, dtype=torch.float32).view(
            h.seq_length, 1).to(dev.torch_device)

    def body(*ts):
        xx, hh, cc, ww = ts[0], ts[1], ts[2], ts[3]
        ll = ts[4] if lens is not None else None
        yy, hyy, cyy = ts[-3], ts[-2], ts[-1]
        out, hn, cn = _forward(h, xx.view(), hh.view(), cc.view(),
                               ww.flat()[:h.weights_size],
                               ll.view() if ll is not None else None, training)
        m = xx._dev.math
        m.copy(out, yy.view())
        m.copy(hn, hyy.view())
        if cn is not None:
            m.copy(cn, cyy.view())
        else:
            m.fill(cyy.view(), 0.0)

    _submit(dev, "RNNForward", body, reads, [y, hy, cy])
    return y, hy, cy


def RNNForwardTraining(x, hx, cx, W, h, seq_lengths=None):
    """-> (y, hy, cy); keeps the activations backward needs in the handle."""
    return _run_forward(x, hx, cx, W, h, seq_lengths, True)


def RNNForwardInference(x, hx, cx, W, h, seq_lengths=None):
    return _run_forward(x, hx, cx, W, h, seq_lengths, False)


def RNNBackward(y, dy, dhy, dcy, W, hx, cx, x, h, seq_lengths=None):
    """-> (dx, dhx, dcx, dW): the reference's BackwardData + BackwardWeights
    as one graph node."""
    dev = x.device()
    shape = _states(h, hx, cx)
    dx = Tensor(list(x.shape()), dev, x.data_type())
    dhx = Tensor(shape, dev, x.data_type())
    dcx = Tensor(shape, dev, x.data_type())
    dW = Tensor([h.weights_size], dev, x.data_type())

    def body(*ts):
        dyy, dhh, dcc, ww = ts[0], ts[1], ts[2], ts[3]
        ddx, ddhx, ddcx, ddw = ts[-4], ts[-3], ts[-2], ts[-1]
        gx, ghx, gcx = _backward(h, dyy.view(), dhh.view(), dcc.view(),
                                 ww.flat()[:h.weights_size],
                                 ddw.flat()[:h.weights_size])
        m = ww._dev.math
        m.copy(gx, ddx.view())
        m.copy(ghx, ddhx.view())
        if gcx is not None:
            m.copy(gcx, ddcx.view())
        else:
            m.fill(ddcx.view(), 0.0)

    _submit(dev, "RNNBackward", body, [dy, dhy, dcy, W, y, hx, cx, x],
            [dx, dhx, dcx, dW])
    return dx, dhx, dcx, dW


def RNNBackwardx(y, dy, dhy, dcy, W, hx, cx, h, seq_lengths=None, x=None):
    """(dx, dhx, dcx); the weight gradient of the same pass is kept for
    ``RNNBackwardW``."""
    assert x is not None or h.reserve is not None
    xin = x
    if xin is None:
        xin = Tensor([h.seq_length, h.batch_size, h.feature_size], y.device(),
                     y.data_type())
    dx, dhx, dcx, dW = RNNBackward(y, dy, dhy, dcy, W, hx, cx, xin, h,
                                   seq_lengths)
    h._last_dW = dW
    return dx, dhx, dcx


def RNNBackwardW(x, hx, y, h, seq_lengths=None):
    assert getattr(h, "_last_dW", None) is not None, \
        "call RNNBackwardx first: both gradients come from one pass"
    return h._last_dW


def RNNSetParam(linLayerID, pseudoLayer, weights, paramValues, is_bias, h):
    """Copy ``paramValues`` into its slot of the flat weight tensor."""
    off, size = h.weights_mapping[(linLayerID, pseudoLayer, bool(is_bias))]
    assert paramValues.Size() == size, "parameter size mismatch"
    from . import tensor_core as tc
    tc.CopyDataToFrom(weights, paramValues, size, off, 0)


def RNNGetParamCopy(linLayerID, pseudoLayer, weights, is_bias, h):
    off, size = h.weights_mapping[(linLayerID, pseudoLayer, bool(is_bias))]
    from . import tensor_core as tc
    out = Tensor([size], weights.device(), weights.data_type())
    tc.CopyDataToFrom(out, weights, size, 0, off)
    return out


# names the reference exports for the GPU build
GpuRNNForwardTraining = RNNForwardTraining
GpuRNNForwardInference = RNNForwardInference
GpuRNNForwardTrainingEx = RNNForwardTraining
GpuRNNForwardInferenceEx = RNNForwardInference
GpuRNNBackwardx = RNNBackwardx
GpuRNNBackwardxEx = RNNBackwardx
GpuRNNBackwardW = RNNBackwardW
GpuRNNBackwardWEx = RNNBackwardW
GpuRNNSetParam = RNNSetParam
GpuRNNGetParamCopy = RNNGetParamCopy

__all__ = [
    "CudnnRNNHandle", "RNNForwardTraining", "RNNForwardInference",
    "RNNBackward", "RNNBackwardx", "RNNBackwardW", "RNNSetParam",
    "RNNGetParamCopy", "GpuRNNForwardTraining", "GpuRNNForwardInference",
    "GpuRNNForwardTrainingEx", "GpuRNNForwardInferenceEx", "GpuRNNBackwardx",
    "GpuRNNBackwardxEx", "GpuRNNBackwardW", "GpuRNNBackwardWEx",
    "GpuRNNSetParam", "GpuRNNGetParamCopy",
]
