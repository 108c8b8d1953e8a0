"""Host math backend.

Every function receives torch *views* of the operand blocks (already carrying
the tensor's shape and strides, including stride-0 broadcast dimensions) and
writes its result into the output view.  The host path is functional rather
than fast: it exists for plumbing tests, data preparation and as the numerical
oracle of the HIP kernels.

Contract parity: reference src/core/tensor/tensor_math.h:86-485 (the set of
primitives every backend must provide) and tensor_math_cpp.h.
"""
import math

import torch
import torch.nn.functional as F

_UNARY = {
    "abs": torch.abs, "exp": torch.exp, "log": torch.log, "sqrt": torch.sqrt,
    "square": torch.square, "sign": torch.sign, "ceil": torch.ceil,
    "floor": torch.floor, "sigmoid": torch.sigmoid, "tanh": torch.tanh,
    "relu": torch.relu, "erf": torch.erf,
    "cos": torch.cos, "cosh": torch.cosh, "acos": torch.acos,
    "acosh": torch.acosh, "sin": torch.sin, "sinh": torch.sinh,
    "asin": torch.asin, "asinh": torch.asinh, "tan": torch.tan,
    "atan": torch.atan, "atanh": torch.atanh,
    "softplus": lambda x: torch.log1p(torch.exp(x)),
    "softsign": lambda x: x / (1 + torch.abs(x)),
    # round half away from zero (C round()) vs round half to even
    "round": lambda x: torch.sign(x) * torch.floor(torch.abs(x) + 0.5),
    "rounde": torch.round,
}

_BINARY = {
    "add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.div,
    "pow": torch.pow,
    "lt": torch.lt, "le": torch.le, "gt": torch.gt, "ge": torch.ge,
    "eq": torch.eq, "max": torch.maximum, "min": torch.minimum,
    # dy where x > 0 else 0  (operands: dy, x)
    "relu_bwd": lambda dy, x: dy * (x > 0).to(dy.dtype),
}


class CpuMath(object):
    name = "cpp"

    def __init__(self, dev):
        self.dev = dev

    # ---- element-wise ------------------------------------------------------
    def fill(self, out, value):
        out.fill_(value)

    def copy(self, src, dst):
        dst.copy_(src)

    def unary(self, op, x, out):
        out.copy_(_UNARY[op](x))

    def clamp(self, x, lo, hi, out):
        out.copy_(torch.clamp(x, lo, hi))

    def binary(self, op, a, b, out):
        out.copy_(_BINARY[op](a, b).to(out.dtype))

    def binary_scalar(self, op, x, s, out):
        if op == "rdiv":  # s / x
            out.copy_(torch.div(s, x))
        elif op in ("max", "min"):
            out.copy_(torch.clamp(x, min=s) if op == "max"
                      else torch.clamp(x, max=s))
        else:
            out.copy_(_BINARY[op](x, s).to(out.dtype))

    def axpy(self, alpha, x, y):
        """y += alpha * x; alpha is a python float or a one-element tensor."""
        if torch.is_tensor(alpha):
            y.add_(x * alpha.reshape(()).to(x.dtype))
        else:
            y.add_(x, alpha=alpha)

    # ---- random -------------------------------------------------------------
    def uniform(self, out, low, high):
        out.copy_(torch.empty(out.shape, dtype=torch.float32).uniform_(
            low, high, generator=self.dev.generator))

    def gaussian(self, out, mean, std):
        out.copy_(torch.empty(out.shape, dtype=torch.float32).normal_(
            mean, std, generator=self.dev.generator))

    def bernoulli(self, out, p):
        r = torch.empty(out.shape, dtype=torch.float32).uniform_(
            0, 1, generator=self.dev.generator)
        out.copy_((r <= p).to(out.dtype))

    # ---- reductions -----------------------------------------------------------
    def sum_all(self, x, out):
        out.copy_(x.sum().reshape(out.shape))

    def sum_axis(self, x, axis, out):
        """x is 2-d; axis 0 sums the rows (-> ncol), axis 1 the columns."""
        out.copy_(x.sum(dim=axis).reshape(out.shape))

    def reduce(self, op, x, axes, out_flat):
        if op == "sum":
            r = x.sum(dim=axes)
        elif op == "mean":
            r = x.mean(dim=axes)
        elif op == "max":
            r = x.amax(dim=axes)
        elif op == "min":
            r = x.amin(dim=axes)
        else:
            raise ValueError(op)
        out_flat[:r.numel()].copy_(r.reshape(-1))

    def index_select(self, x, axis, idx, out):
        idx = idx.long()
        idx = torch.where(idx < 0, idx + x.shape[axis], idx)
        out.copy_(torch.index_select(x, axis, idx))

    def index_add(self, dy, axis, idx, out):
        idx = idx.long()
        idx = torch.where(idx < 0, idx + out.shape[axis], idx)
        out.zero_()
        out.index_add_(axis, idx, dy)

    def row_max(self, x, out):
        out.copy_(x.max(dim=1).values.reshape(out.shape))

    def asum(self, x):
        return float(x.abs().sum())

    def nrm2(self, x):
        return float(x.float().norm())

    def sum_float(self, x):
        return float(x.sum())

    # ---- linear algebra ---------------------------------------------------------
    def gemm(self, alpha, a, b, beta, c, bias=None, act=0):
        """c = act(alpha * a @ b + bias + beta * c) for 2-d (or batched 3-d)
        views; ``bias`` has one value per column, ``act`` 1 is ReLU."""
        r = torch.matmul(a, b)
        if alpha != 1.0:
            r = r * alpha
        if bias is not None:
            r = r + bias
        if act == 1:
            assert not beta
            r = torch.relu(r)
        if beta:
            c.mul_(beta).add_(r)
        else:
            c.copy_(r)

    def gemv(self, alpha, a, v, beta, out):
        r = torch.mv(a, v)
        if alpha != 1.0:
            r = r * alpha
        if beta:
            out.mul_(beta).add_(r)
        else:
            out.copy_(r)

    def dgmm_rows(self, m, v, out):
        """out[i, j] = m[i, j] * v[j]"""
        out.copy_(m * v.reshape(1, -1))

    def dgmm_cols(self, m, v, out):
        """out[i, j] = m[i, j] * v[i]"""
        out.copy_(m * v.reshape(-1, 1))

    # ---- softmax / losses -----------------------------------------------------------
    def softmax(self, x, out):
        out.copy_(torch.softmax(x.float(), dim=1).to(out.dtype))

    def softmax_bwd(self, dy, y, dx):
        s = (dy * y).sum(dim=1, keepdim=True)
        dx.copy_((dy - s) * y)

    def cross_entropy_fwd(self, p, t, loss):
        """p (batch, dim) probabilities; t int labels (batch) or a (batch, dim)
        distribution that is normalised per row."""
        batch = p.shape[0]
        tiny = torch.finfo(torch.float32).tiny
        pf = p.float()
        if t.numel() == batch:
            idx = t.reshape(batch).long()
            picked = pf[torch.arange(batch), idx]
            loss.copy_((-torch.log(torch.clamp(picked, min=tiny))).to(loss.dtype))
        else:
            tf = t.reshape(p.shape).float()
            w = tf / tf.sum(dim=1, keepdim=True)
            loss.copy_((-(w * torch.log(torch.clamp(pf, min=tiny))).sum(dim=1)
                        ).to(loss.dtype))

    def softmax_ce_bwd(self, p, t):
        """in place on p: p - onehot(t)  (or p - t for dense targets)."""
        batch = p.shape[0]
        if t.numel() == batch:
            idx = t.reshape(batch).long()
            p[torch.arange(batch), idx] -= 1.0
        else:
            p.sub_(t.reshape(p.shape).to(p.dtype))

    def softmax_ce_fused_fwd(self, x, t, p, loss):
        """p = softmax(x); loss = mean over the batch of the cross entropy."""
        self.softmax(x, p)
        per = torch.empty(x.shape[0], dtype=torch.float32)
        self.cross_entropy_fwd(p, t, per)
        loss.copy_((per.sum() / x.shape[0]).reshape(loss.shape))

    def softmax_ce_fused_bwd(self, p, t, dx):
        dx.copy_(p)
        self.softmax_ce_bwd(dx, t)
        dx.div_(p.shape[0])

    # ---- convolution ------------------------------------------------------------------
    STAT_COPIES = 1

    def conv_fwd(self, x, w, b, y, stride, pad, groups, stats=None,
                 shift=None):
        """``stats`` (float[STAT_COPIES * 2 * K + K]) receives the output's
        per-channel sums of y - shift and (y - shift)^2, for a following batch
        normalisation that then skips its own statistics pass."""
        y.copy_(F.conv2d(x, w, b, stride=stride, padding=pad, groups=groups))
        if stats is not None:
            K = y.shape[1]
            d = y.float() - shift.float().reshape(1, K, 1, 1)
            stats[:K] += d.sum(dim=(0, 2, 3))
            stats[K:2 * K] += (d * d).sum(dim=(0, 2, 3))

    def conv_bwd_data(self, dy, w, dx, stride, pad, groups, addend=None,
                      side_zero=None):
        r = torch.nn.grad.conv2d_input(dx.shape, w, dy, stride=stride,
                                       padding=pad, groups=groups)
        dx.copy_(r if addend is None else r + addend)
        return False

    def conv_bwd_weight(self, dy, x, dw, stride, pad, groups, zeroed=False):
        r = torch.nn.grad.conv2d_weight(x, dw.shape, dy, stride=stride,
                                        padding=pad, groups=groups)
        dw.copy_(r)

    def conv_bwd_bias(self, dy, db):
        db.copy_(dy.sum(dim=(0, 2, 3)))

    # ---- batch norm ------------------------------------------------------------------------
    @staticmethod
    def _bn_act(z, act, residual):
        if act == 2:
            z = z + residual.float()
        if act >= 1:
            z = torch.relu(z)
        return z

    def bn_fwd_train(self, x, scale, bias, rmean, rvar, factor, eps, y, mean,
                     invstd, act=0, residual=None, stats=None):
        """Channel statistics over (N, H, W) (or N for 2-d input).  Running
        stats follow running = (1 - factor) * running + factor * batch with the
        unbiased batch variance.  With ``stats`` (filled by the producer of x,
        see conv_fwd) the statistics are taken from there and the buffer is
        cleared for the next step."""
        dims = [0] + list(range(2, x.dim()))
        m = x.numel() // x.shape[1]
        xf = x.float()
        if stats is not None:
            C = x.shape[1]
            dm = stats[:C] / m
            mu = rmean.float() + dm
            var = (stats[C:2 * C] / m - dm * dm).clamp_(min=0)
            stats.zero_()
        else:
            mu = xf.mean(dim=dims)
            var = xf.var(dim=dims, unbiased=False)
        istd = torch.rsqrt(var + eps)
        shp = [1, -1] + [1] * (x.dim() - 2)
        y.copy_(self._bn_act(
            (xf - mu.reshape(shp)) * istd.reshape(shp) *
            scale.float().reshape(shp) + bias.float().reshape(shp), act,
            residual).to(y.dtype))
        mean.copy_(mu)
        invstd.copy_(istd)
        unbiased = var * (m / max(m - 1, 1))
        rmean.mul_(1 - factor).add_(mu.to(rmean.dtype), alpha=factor)
        rvar.mul_(1 - factor).add_(unbiased.to(rvar.dtype), alpha=factor)

    def bn_fwd_infer(self, x, scale, bias, rmean, rvar, eps, y, act=0,
                     residual=None):
        shp = [1, -1] + [1] * (x.dim() - 2)
        istd = torch.rsqrt(rvar.float() + eps)
        y.copy_(self._bn_act(
            (x.float() - rmean.float().reshape(shp)) * istd.reshape(shp) *
            scale.float().reshape(shp) + bias.float().reshape(shp), act,
            residual).to(y.dtype))

    def bn_bwd(self, dy, x, scale, mean, invstd, dx, dscale, dbias, act=0,
               bias=None, y=None, dres=None):
        dims = [0] + list(range(2, x.dim()))
        shp = [1, -1] + [1] * (x.dim() - 2)
        m = x.numel() // x.shape[1]
        xhat = (x.float() - mean.reshape(shp)) * invstd.reshape(shp)
        dyf = dy.float()
        if act == 1:
            z = xhat * scale.float().reshape(shp) + bias.float().reshape(shp)
            dyf = dyf * (z > 0)
        elif act == 2:
            dyf = dyf * (y.float() > 0)
            dres.copy_(dyf.to(dres.dtype))
        db = dyf.sum(dim=dims)
        ds = (dyf * xhat).sum(dim=dims)
        g = scale.float().reshape(shp) * invstd.reshape(shp)
        dx.copy_((g * (dyf - db.reshape(shp) / m - xhat * ds.reshape(shp) / m)
                  ).to(dx.dtype))
        dscale.copy_(ds.to(dscale.dtype))
        dbias.copy_(db.to(dbias.dtype))

    # ---- pooling ----------------------------------------------------------------------------------
    def pool_fwd(self, x, y, kernel, stride, pad, is_max):
        if is_max:
            y.copy_(F.max_pool2d(x, kernel, stride, pad))
        else:
            y.copy_(F.avg_pool2d(x, kernel, stride, pad,
                                 count_include_pad=False))

    def pool_bwd(self, dy, x, y, dx, kernel, stride, pad, is_max):
        xr = x.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            if is_max:
                yy = F.max_pool2d(xr, kernel, stride, pad)
            else:
                yy = F.avg_pool2d(xr, kernel, stride, pad,
                                  count_include_pad=False)
            yy.backward(dy)
        dx.copy_(xr.grad)

    # ---- index / gather ------------------------------------------------------------------------------
    def gather_rows(self, table, idx, out):
        out.copy_(table[idx.reshape(-1).long()].reshape(out.shape))

    def scatter_add_rows(self, grad, idx, dtable):
        dtable.zero_()
        dtable.index_add_(0, idx.reshape(-1).long(),
                          grad.reshape(-1, dtable.shape[1]))

    # ---- collectives support (reference src/io/communicator.cc) ----------------------------
    def pack_plan(self, views):
        offs, off = [], 0
        for v in views:
            offs.append(off)
            off += v.numel()
        return offs

    def pack(self, plan, views, bucket):
        for off, v in zip(plan, views):
            bucket[off:off + v.numel()].copy_(v.reshape(-1))

    def unpack(self, plan, bucket, views, scale):
        for off, v in zip(plan, views):
            src = bucket[off:off + v.numel()].view(v.shape)
            v.copy_(src if scale == 1.0 else src.to(torch.float32) * scale)

    def sparsify_threshold(self, x, thr):
        idx = torch.nonzero(x.abs() >= thr).flatten().to(torch.int32)
        n = x.numel()
        vals = torch.zeros(n, dtype=torch.float32)
        full = torch.zeros(n, dtype=torch.int32)
        vals[:idx.numel()] = x[idx.long()]
        full[:idx.numel()] = idx
        return vals, full, torch.tensor([idx.numel()], dtype=torch.int32)

    def topk_abs(self, x, k):
        _, idx = torch.topk(x.abs(), k)
        return x[idx].contiguous(), idx.to(torch.int32)

    def scatter_add_sparse(self, vals, idx, cnt, dense):
        c = vals.numel() if cnt is None else min(int(cnt.item()), vals.numel())
        dense.index_add_(0, idx[:c].long(), vals[:c])

    def scatter_sub(self, vals, idx, cnt, dense):
        c = vals.numel() if cnt is None else min(int(cnt.item()), vals.numel())
        dense.index_add_(0, idx[:c].long(), -vals[:c])

    def sparse_select(self, x, thr, topk, payload, cap, work):
        n = x.numel()
        if topk:
            k = min(cap, max(1, int(math.ceil(thr * n))))
            _, idx = torch.topk(x.abs(), k)
            count = k
        else:
            idx = torch.nonzero(x.abs() >= thr).flatten()
            count = idx.numel()
            idx = idx[:cap]
        m = idx.numel()
        payload[0] = count
        payload[1:1 + m] = idx.to(torch.int32)
        payload[1 + cap:1 + cap + m] = x[idx].contiguous().view(torch.int32)

    def sparse_workspace(self):
        return None

    def sparse_accumulate(self, payloads, nseg, cap, dense, sign=1.0):
        stride = 1 + 2 * cap
        for r in range(nseg):
            p = payloads[r * stride:(r + 1) * stride]
            c = min(int(p[0]), cap)
            vals = p[1 + cap:1 + cap + c].contiguous().view(torch.float32)
            dense.index_add_(0, p[1:1 + c].long(), vals * sign)

    def adaptive_update_multi(self, rule, ps, gs, s1, s2, lr, wd, eps, a, b,
                              step, scale):
        f = lambda t: float(t.reshape(-1)[0]) if t is not None else 0.0
        lr, wd, eps, a, b = f(lr), f(wd), f(eps), f(a), f(b)
        t = f(step) + 1.0
        for i, (p, g) in enumerate(zip(ps, gs)):
            pv = p.float()
            gv = g.float() * scale + wd * pv
            u = s1[i].float()
            if rule == 0:
                u = a * u + (1.0 - a) * gv * gv
                upd = gv / torch.sqrt(u + eps)
            elif rule == 1:
                u = u + gv * gv
                upd = gv / torch.sqrt(u + eps)
            else:
                w = b * s2[i].float() + (1.0 - b) * gv * gv
                u = a * u + (1.0 - a) * gv
                s2[i].copy_(w)
                upd = (u / (1.0 - a ** t)) / (torch.sqrt(w / (1.0 - b ** t)) +
                                              eps)
            s1[i].copy_(u)
            p.copy_(pv - lr * upd)

    # ---- fused attention (same contract as the HIP backend) ------------------------
    @staticmethod
    def _attn_probs(q, kt, mask, scale):
        sc = torch.matmul(q.float(), kt.float()) * scale
        if mask is not None:
            sc = sc + mask.float()
        return sc

    def attention_fwd(self, q, kt, v, mask, scale, out, lse):
        sc = self._attn_probs(q, kt, mask, scale)
        lse.copy_(torch.logsumexp(sc, dim=-1))
        p = torch.softmax(sc, dim=-1)
        out.copy_(torch.matmul(p, v.float()).permute(0, 2, 1, 3))

    def attention_bwd(self, dout, q, kt, v, mask, scale, out, lse, dq, dk,
                      dv):
        sc = self._attn_probs(q, kt, mask, scale)
        p = torch.exp(sc - lse.unsqueeze(-1))
        do = dout.float().permute(0, 2, 1, 3)            # (B, H, Sq, D)
        o = out.float().permute(0, 2, 1, 3)
        delta = (do * o).sum(-1, keepdim=True)
        dv.copy_(torch.matmul(p.transpose(-1, -2), do).permute(0, 2, 1, 3))
        dp = torch.matmul(do, v.float().transpose(-1, -2))
        ds = p * (dp - delta) * scale
        dq.copy_(torch.matmul(ds, kt.float().transpose(-1, -2))
                 .permute(0, 2, 1, 3))
        dk.copy_(torch.matmul(ds.transpose(-1, -2), q.float())
                 .permute(0, 2, 1, 3))

    # ---- layer norm / gelu ------------------------------------------------------
    def layernorm_fwd(self, x, gamma, beta, y, mean, rstd, eps):
        xf = x.float()
        D = x.shape[-1]
        m = xf.mean(dim=-1, keepdim=True)
        v = ((xf - m) ** 2).mean(dim=-1, keepdim=True)
        rs = torch.rsqrt(v + eps)
        out = (xf - m) * rs
        if gamma is not None:
            out = out * gamma.float()
        if beta is not None:
            out = out + beta.float()
        y.copy_(out)
        mean.copy_(m.reshape(-1))
        rstd.copy_(rs.reshape(-1))

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dx, dgamma, dbeta):
        D = x.shape[-1]
        xf, df = x.float().reshape(-1, D), dy.float().reshape(-1, D)
        xh = (xf - mean.reshape(-1, 1)) * rstd.reshape(-1, 1)
        gd = df * gamma.float().reshape(1, D) if gamma is not None else df
        s1 = gd.mean(dim=1, keepdim=True)
        s2 = (gd * xh).mean(dim=1, keepdim=True)
        dx.copy_((rstd.reshape(-1, 1) * (gd - s1 - xh * s2)).reshape(x.shape))
        if dgamma is not None:
            dgamma.copy_((df * xh).sum(dim=0))
            dbeta.copy_(df.sum(dim=0))

    def gelu_fwd(self, x, y):
        xf = x.float()
        y.copy_(0.5 * xf * (1.0 + torch.erf(xf * 0.7071067811865476)))

    def gelu_bwd(self, dy, x, dx):
        xf = x.float()
        cdf = 0.5 * (1.0 + torch.erf(xf * 0.7071067811865476))
        pdf = 0.3989422804014327 * torch.exp(-0.5 * xf * xf)
        dx.copy_(dy.float() * (cdf + xf * pdf))

    # ---- fused optimiser step --------------------------------------------------------------------------
    def sgd_update_multi(self, ps, gs, bufs, lr, momentum, dampening,
                         weight_decay, nesterov, scale, masters=None):
        for i, (p, g) in enumerate(zip(ps, gs)):
            if masters:
                m = masters[i]
                self.sgd_update(m, g.to(m.dtype), bufs[i] if bufs else None,
                                lr, momentum, dampening, weight_decay,
                                nesterov, scale)
                p.copy_(m)
            else:
                self.sgd_update(p, g, bufs[i] if bufs else None, lr, momentum,
                                dampening, weight_decay, nesterov, scale)

    def sgd_update(self, p, g, buf, lr, momentum, dampening, weight_decay,
                   nesterov, scale):
        """One fused SGD step.  lr / momentum / ... are one-element tensors
        (device-resident hyper-parameters); ``scale`` multiplies the incoming
        gradient (1/world_size for data parallel training)."""
        lr = float(lr.reshape(-1)[0])
        mom = float(momentum.reshape(-1)[0])
        damp = float(dampening.reshape(-1)[0])
        wd = float(weight_decay.reshape(-1)[0])
        gg = g * scale if scale != 1.0 else g.clone()
        if wd != 0:
            gg = gg + wd * p
        if buf is not None:
            buf.mul_(mom).add_(gg, alpha=1.0 - damp)
            gg = gg + mom * buf if nesterov else buf
        p.add_(gg, alpha=-lr)

    # ---- recurrent cells ----------------------------------------------------------------------------------
    def lstm_cell_fwd(self, gates, c_prev, c, h, acts):
        """gates (batch, 4*H) pre-activations ordered i, f, o, g."""
        H = c_prev.shape[1]
        i = torch.sigmoid(gates[:, 0:H])
        f = torch.sigmoid(gates[:, H:2 * H])
        o = torch.sigmoid(gates[:, 2 * H:3 * H])
        g = torch.tanh(gates[:, 3 * H:4 * H])
        c.copy_(f * c_prev + i * g)
        h.copy_(o * torch.tanh(c))
        acts.copy_(torch.cat([i, f, o, g], dim=1))

    def lstm_cell_bwd(self, dh, dc_next, acts, c_prev, c, dgates, dc_prev):
        H = c_prev.shape[1]
        i, f, o, g = (acts[:, 0:H], acts[:, H:2 * H], acts[:, 2 * H:3 * H],
                      acts[:, 3 * H:4 * H])
        tc = torch.tanh(c)
        do = dh * tc
        dc = dc_next + dh * o * (1 - tc * tc)
        di = dc * g
        dg = dc * i
        df = dc * c_prev
        dc_prev.copy_(dc * f)
        dgates.copy_(torch.cat([di * i * (1 - i), df * f * (1 - f),
                                do * o * (1 - o), dg * (1 - g * g)], dim=1))

    def gru_cell_fwd(self, gx, gh, brn, h_prev, h, acts, ghn):
        """One GRU step from the (batch, 3H) input and recurrent projections
        (gates r, z, n); ``brn`` is added to gh_n inside the reset-gated
        term."""
        H = h_prev.shape[1]
        r = torch.sigmoid(gx[:, :H] + gh[:, :H])
        z = torch.sigmoid(gx[:, H:2 * H] + gh[:, H:2 * H])
        hn = gh[:, 2 * H:] if brn is None else gh[:, 2 * H:] + brn.view(1, H)
        n = torch.tanh(gx[:, 2 * H:] + r * hn)
        ghn.copy_(hn)
        acts.copy_(torch.cat([r, z, n], dim=1))
        h.copy_(n + z * (h_prev - n))

    def gru_cell_bwd(self, dh, acts, ghn, h_prev, dgx, dgh, dh_prev):
        H = h_prev.shape[1]
        r, z, n = acts[:, :H], acts[:, H:2 * H], acts[:, 2 * H:]
        dpre_n = dh * (1 - z) * (1 - n * n)
        dr = dpre_n * ghn * r * (1 - r)
        dz = dh * (h_prev - n) * z * (1 - z)
        dgx.copy_(torch.cat([dr, dz, dpre_n], dim=1))
        dgh.copy_(torch.cat([dr, dz, dpre_n * r], dim=1))
        dh_prev.copy_(dh * z)

    def rnn_act_bwd(self, mode, dh, h, d):
        """d = dh * act'(h); mode 0 relu, 1 tanh."""
        d.copy_(dh * (1 - h * h) if mode == 1 else dh * (h > 0).to(dh.dtype))

    def lstm_seq_supported(self, batch, hidden, dtype):
        """The single-launch sequence kernels exist on the GPU only."""
        return False
