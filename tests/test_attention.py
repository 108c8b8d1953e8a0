"""Fused attention: the autograd operator against a numpy evaluation and
against the unfused operator chain (host), and the HIP kernels against the
fp32 host backend (forward, backward, masks, ragged lengths, 16-bit)."""
import numpy as np
import pytest
import torch

from singa_amd import autograd, device, tensor
from singa_amd.core import math_cpu


def _ref(q, k, v, mask, scale):
    """q, k, v: (B, S, H, D) float64 -> out (B, Sq, H * D)"""
    qh, kh, vh = [t.transpose(0, 2, 1, 3) for t in (q, k, v)]
    sc = qh @ kh.transpose(0, 1, 3, 2) * scale
    if mask is not None:
        sc = sc + mask
    sc = sc - sc.max(-1, keepdims=True)
    p = np.exp(sc)
    p /= p.sum(-1, keepdims=True)
    o = (p @ vh).transpose(0, 2, 1, 3)
    return o.reshape(o.shape[0], o.shape[1], -1)


def _views(dev, q, k, v):
    """(B, S, H, D) arrays -> the transposed views an exported graph feeds to
    the two MatMuls: q (B,H,S,D), kT (B,H,D,S), v (B,H,S,D)"""
    tq, tk, tv = [tensor.from_numpy(a.astype(np.float32), dev)
                  for a in (q, k, v)]
    for t in (tq, tk, tv):
        t.requires_grad = True
        t.stores_grad = True
    return (tq, tk, tv,
            autograd.transpose(tq, (0, 2, 1, 3)),
            autograd.transpose(tk, (0, 2, 3, 1)),
            autograd.transpose(tv, (0, 2, 1, 3)))


@pytest.mark.parametrize("D", [64, 16])
def test_attention_operator_matches_numpy_and_unfused_chain(D):
    dev = device.get_default_device()
    rs = np.random.RandomState(0)
    B, Sq, Sk, H = 2, 7, 9, 3
    q = rs.randn(B, Sq, H, D)
    k = rs.randn(B, Sk, H, D)
    v = rs.randn(B, Sk, H, D)
    mask = np.where(rs.rand(B, 1, 1, Sk) < 0.3, -10000.0, 0.0)
    scale = 1.0 / np.sqrt(D)
    want = _ref(q, k, v, mask, scale)
    w = rs.randn(*want.shape).astype(np.float32)

    autograd.training = True
    try:
        grads = []
        for fused in (True, False):
            tq, tk, tv, q4, kt, v4 = _views(dev, q, k, v)
            tm = tensor.from_numpy(mask.astype(np.float32), dev)
            tm.requires_grad = False
            if fused:
                y = autograd.attention(q4, kt, v4, tm, scale)
            else:
                sc = autograd.mul(autograd.matmul(q4, kt), tensor.from_numpy(
                    np.array([scale], np.float32), dev))
                p = autograd.softmax(autograd.add(sc, tm), axis=-1)
                y = autograd.reshape(autograd.transpose(
                    autograd.matmul(p, v4), (0, 2, 1, 3)), (B, Sq, H * D))
            np.testing.assert_allclose(tensor.to_numpy(y), want, rtol=1e-4,
                                       atol=1e-5)
            loss = autograd.reduce_sum(autograd.mul(
                y, tensor.from_numpy(w, dev)), None, 0)
            g = {id(p): tensor.to_numpy(dp)
                 for p, dp in autograd.backward(loss)}
            grads.append([g[id(t)] for t in (tq, tk, tv)])
        for a, b in zip(*grads):
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4)
    finally:
        autograd.training = False


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,Sq,Sk,masked", [
    (2, 3, 256, 256, True),    # the BERT shape: two 64-key chunks
    (1, 2, 40, 75, True),      # ragged tiles in both directions
    (2, 1, 130, 33, False),    # no mask, a single partial key tile
    (1, 12, 128, 128, True),
    (2, 2, 96, 96, "causal"),  # a mask that depends on the query
])
def test_attention_kernels_match_host(dt, B, H, Sq, Sk, masked):
    """Forward and backward of the HIP kernels on strided (B, S, H, D) views
    against the host backend in float32 on the same (rounded) inputs; one
    batch entry has most keys hidden, which moves the running maximum late in
    the key loop."""
    dev = device.create_cuda_gpu()
    hm, cm = dev.math, math_cpu.CpuMath(device.get_default_device())
    D = 64
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, Sq, H, D, generator=g).to(dt)
    k = torch.randn(B, Sk, H, D, generator=g).to(dt)
    v = torch.randn(B, Sk, H, D, generator=g).to(dt)
    q[0, Sq // 2] *= 4.0          # a row whose maximum jumps between tiles
    mask = None
    if masked == "causal":
        mask = torch.triu(torch.full((Sq, Sk), -10000.0), 1)[None, None]
        mask = mask.to(dt)
    elif masked:
        mask = torch.zeros(B, 1, 1, Sk)
        mask[0, :, :, : Sk - 3] = -10000.0
        mask[-1, :, :, 5::7] = -10000.0
        mask = mask.to(dt)
    do = torch.randn(B, Sq, H, D, generator=g).to(dt)
    scale = 0.125

    def run(m, tensors, dev_t):
        qq, kk, vv, mm, dd = [None if t is None else t.to(dev_t)
                              for t in tensors]
        q4 = qq.permute(0, 2, 1, 3)
        kt = kk.permute(0, 2, 3, 1)
        v4 = vv.permute(0, 2, 1, 3)
        out = torch.empty(B, Sq, H, D, dtype=qq.dtype, device=dev_t)
        lse = torch.empty(B, H, Sq, dtype=torch.float32, device=dev_t)
        m.attention_fwd(q4, kt, v4, mm, scale, out, lse)
        dq = torch.empty(B, Sq, H, D, dtype=qq.dtype, device=dev_t)
        dk = torch.empty(B, Sk, H, D, dtype=qq.dtype, device=dev_t)
        dv = torch.empty(B, Sk, H, D, dtype=qq.dtype, device=dev_t)
        m.attention_bwd(dd, q4, kt, v4, mm, scale, out, lse, dq, dk, dv)
        return [t.float().cpu().numpy() for t in (out, lse, dq, dk, dv)]

    host = [None if t is None else t.float() for t in (q, k, v, mask, do)]
    want = run(cm, host, torch.device("cpu"))
    got = run(hm, (q, k, v, mask, do), dev.torch_device)
    eps = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dt]
    for name, a, b in zip(("out", "lse", "dq", "dk", "dv"), got, want):
        scale_ref = max(1.0, float(np.abs(b).max()))
        np.testing.assert_allclose(a, b, rtol=eps * 4, atol=eps * scale_ref,
                                   err_msg=name)
