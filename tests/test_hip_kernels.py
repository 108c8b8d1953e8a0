"""Numerics of every HIP kernel against a plain PyTorch float32 reference of
the same op (the host math backend)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def maths(gpu_dev):
    from singa_amd import device
    return gpu_dev.math, device.get_default_device().math, gpu_dev


def g(t, dev):
    return t.to(dev.torch_device)


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    r = np.random.RandomState(seed)
    return torch.from_numpy(r.uniform(lo, hi, shape).astype(np.float32))


UNARY = ["abs", "exp", "log", "sqrt", "square", "sign", "ceil", "floor",
         "round", "rounde", "sigmoid", "tanh", "relu", "erf", "cos", "cosh",
         "acos", "acosh", "sin", "sinh", "asin", "asinh", "tan", "atan",
         "atanh", "softplus", "softsign"]


@pytest.mark.parametrize("op", UNARY)
def test_unary(maths, op):
    hm, cm, dev = maths
    lo, hi = (-0.9, 0.9)
    if op in ("log", "sqrt"):
        lo, hi = 0.1, 3.0
    if op == "acosh":
        lo, hi = 1.1, 3.0
    x = rnd(1000, 7, seed=1, lo=lo, hi=hi)
    ref = torch.empty_like(x)
    cm.unary(op, x, ref)
    out = torch.empty_like(x, device=dev.torch_device)
    hm.unary(op, g(x, dev), out)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5,
                               atol=2e-6)


@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "pow", "lt", "le",
                                "gt", "ge", "eq", "max", "min", "relu_bwd"])
def test_binary_broadcast_and_transpose(maths, op):
    hm, cm, dev = maths
    a = rnd(6, 5, 33, seed=2, lo=0.2, hi=2.0)
    b = rnd(5, 1, seed=3, lo=0.2, hi=2.0)
    bt = b.expand(6, 5, 33)
    ref = torch.empty(6, 5, 33)
    cm.binary(op, a, bt, ref)
    out = torch.empty(6, 5, 33, device=dev.torch_device)
    hm.binary(op, g(a, dev), g(b, dev).expand(6, 5, 33), out)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5)
    # transposed operand
    at = rnd(33, 5, 6, seed=4, lo=0.2, hi=2.0)
    ref2 = torch.empty(6, 5, 33)
    cm.binary(op, at.permute(2, 1, 0), a, ref2)
    out2 = torch.empty(6, 5, 33, device=dev.torch_device)
    hm.binary(op, g(at, dev).permute(2, 1, 0), g(a, dev), out2)
    np.testing.assert_allclose(out2.cpu().numpy(), ref2.numpy(), rtol=2e-5)


@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "pow", "lt", "ge",
                                "eq", "rdiv", "max", "min"])
def test_scalar(maths, op):
    hm, cm, dev = maths
    x = rnd(4099, seed=5, lo=0.5, hi=2.0)
    ref = torch.empty_like(x)
    cm.binary_scalar(op, x, 1.25, ref)
    out = torch.empty_like(x, device=dev.torch_device)
    hm.binary_scalar(op, g(x, dev), 1.25, out)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.int32])
def test_cast_and_strided_copy(maths, dt):
    hm, cm, dev = maths
    x = rnd(17, 9, 5, seed=6, lo=-50, hi=50)
    out = torch.empty(5, 9, 17, dtype=dt, device=dev.torch_device)
    hm.copy(g(x, dev).permute(2, 1, 0), out)
    ref = x.permute(2, 1, 0).to(dt)
    assert torch.equal(out.cpu().float(), ref.float())
    back = torch.empty(5, 9, 17, device=dev.torch_device)
    hm.copy(out, back)
    assert torch.equal(back.cpu(), ref.float())


def test_fill_axpy(maths):
    hm, cm, dev = maths
    y = torch.empty(1000, device=dev.torch_device)
    hm.fill(y, 2.5)
    assert float(y.min()) == 2.5 and float(y.max()) == 2.5
    x = g(rnd(1000, seed=7), dev)
    hm.axpy(0.5, x, y)
    np.testing.assert_allclose(y.cpu().numpy(), 2.5 + 0.5 * x.cpu().numpy(),
                               rtol=1e-6)
    alpha = torch.tensor([-2.0], device=dev.torch_device)
    hm.axpy(alpha, x, y)
    np.testing.assert_allclose(y.cpu().numpy(), 2.5 - 1.5 * x.cpu().numpy(),
                               rtol=1e-5, atol=1e-6)


def test_reductions(maths):
    hm, cm, dev = maths
    x = rnd(300, 70, seed=8)
    xg = g(x, dev)
    out = torch.empty(1, device=dev.torch_device)
    hm.sum_all(xg, out)
    np.testing.assert_allclose(out.cpu().numpy(), [x.sum().item()], rtol=1e-4)
    o0 = torch.empty(70, device=dev.torch_device)
    hm.sum_axis(xg, 0, o0)
    np.testing.assert_allclose(o0.cpu().numpy(), x.sum(0).numpy(), rtol=1e-4,
                               atol=1e-5)
    o1 = torch.empty(300, device=dev.torch_device)
    hm.sum_axis(xg, 1, o1)
    np.testing.assert_allclose(o1.cpu().numpy(), x.sum(1).numpy(), rtol=1e-4,
                               atol=1e-5)
    hm.row_max(xg, o1)
    np.testing.assert_allclose(o1.cpu().numpy(), x.max(1).values.numpy())
    assert abs(hm.asum(xg) - x.abs().sum().item()) < 1e-1
    assert abs(hm.nrm2(xg) - x.norm().item()) < 1e-3
    big = rnd(5000, 130, seed=9)
    ob = torch.empty(130, device=dev.torch_device)
    hm.sum_axis(g(big, dev), 0, ob)
    np.testing.assert_allclose(ob.cpu().numpy(), big.sum(0).numpy(), rtol=1e-4,
                               atol=1e-3)
    x4 = rnd(3, 4, 5, 6, seed=10)
    for axes, op in [([1, 3], "sum"), ([0], "mean"), ([2, 3], "max"),
                     ([0, 1], "sum"), ([3], "min")]:
        ref = torch.empty(360)
        cm.reduce(op, x4, axes, ref)
        outf = torch.empty(360, device=dev.torch_device)
        hm.reduce(op, g(x4, dev), axes, outf)
        n = x4.numel() // int(np.prod([x4.shape[a] for a in axes]))
        np.testing.assert_allclose(outf.cpu().numpy()[:n], ref.numpy()[:n],
                                   rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cols", [10, 1000, 5000])
def test_softmax_and_losses(maths, cols):
    hm, cm, dev = maths
    x = rnd(37, cols, seed=11, lo=-4, hi=4)
    t = torch.from_numpy(np.random.RandomState(1).randint(0, cols, 37)
                         .astype(np.int32))
    ref = torch.empty_like(x)
    cm.softmax(x, ref)
    out = torch.empty_like(x, device=dev.torch_device)
    hm.softmax(g(x, dev), out)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5,
                               atol=1e-7)
    dy = rnd(37, cols, seed=12)
    rdx = torch.empty_like(x)
    cm.softmax_bwd(dy, ref, rdx)
    dx = torch.empty_like(x, device=dev.torch_device)
    hm.softmax_bwd(g(dy, dev), out, dx)
    np.testing.assert_allclose(dx.cpu().numpy(), rdx.numpy(), rtol=1e-4,
                               atol=1e-7)
    rl = torch.empty(37)
    cm.cross_entropy_fwd(ref, t, rl)
    l = torch.empty(37, device=dev.torch_device)
    hm.cross_entropy_fwd(out, g(t, dev), l)
    np.testing.assert_allclose(l.cpu().numpy(), rl.numpy(), rtol=1e-5)
    # fused
    rp, rloss = torch.empty_like(x), torch.empty(1)
    cm.softmax_ce_fused_fwd(x, t, rp, rloss)
    p = torch.empty_like(x, device=dev.torch_device)
    loss = torch.empty(1, device=dev.torch_device)
    hm.softmax_ce_fused_fwd(g(x, dev), g(t, dev), p, loss)
    np.testing.assert_allclose(loss.cpu().numpy(), rloss.numpy(), rtol=1e-5)
    np.testing.assert_allclose(p.cpu().numpy(), rp.numpy(), rtol=1e-5, atol=1e-7)
    rdx2 = torch.empty_like(x)
    cm.softmax_ce_fused_bwd(rp, t, rdx2)
    dx2 = torch.empty_like(x, device=dev.torch_device)
    hm.softmax_ce_fused_bwd(p, g(t, dev), dx2)
    np.testing.assert_allclose(dx2.cpu().numpy(), rdx2.numpy(), rtol=1e-5,
                               atol=1e-8)
    # dense targets
    td = torch.softmax(rnd(37, cols, seed=13), dim=1)
    cm.cross_entropy_fwd(ref, td, rl)
    hm.cross_entropy_fwd(out, g(td, dev), l)
    np.testing.assert_allclose(l.cpu().numpy(), rl.numpy(), rtol=1e-4)


def test_rng_moments(maths):
    hm, cm, dev = maths
    dev.SetRandSeed(7)
    n = 1 << 20
    y = torch.empty(n, device=dev.torch_device)
    hm.uniform(y, -1.0, 3.0)
    assert abs(float(y.mean()) - 1.0) < 0.01 and float(y.min()) >= -1.0
    assert float(y.max()) <= 3.0
    hm.gaussian(y, 0.5, 2.0)
    assert abs(float(y.mean()) - 0.5) < 0.01
    assert abs(float(y.std()) - 2.0) < 0.01
    hm.bernoulli(y, 0.3)
    assert abs(float(y.mean()) - 0.3) < 0.005
    a = torch.empty(1000, device=dev.torch_device)
    b = torch.empty(1000, device=dev.torch_device)
    hm.uniform(a, 0, 1)
    hm.uniform(b, 0, 1)
    assert not torch.equal(a, b)  # the device-side offset advanced
    dev.SetRandSeed(7)
    c = torch.empty(n, device=dev.torch_device)
    hm.uniform(c, -1.0, 3.0)
    dev.SetRandSeed(7)
    d = torch.empty(n, device=dev.torch_device)
    hm.uniform(d, -1.0, 3.0)
    assert torch.equal(c, d)  # reproducible for a fixed seed


GEMM_SHAPES = [(64, 2048, 1600), (33, 17, 65), (400, 3, 2), (256, 768, 768),
               (1, 1000, 2048), (130, 129, 131)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_f32_layouts(maths, M, N, K):
    hm, cm, dev = maths
    a, b = rnd(M, K, seed=20), rnd(K, N, seed=21)
    ref = a.double() @ b.double()
    tol = 2e-6 * K
    c = torch.empty(M, N, device=dev.torch_device)
    hm.gemm(1.0, g(a, dev), g(b, dev), 0.0, c)
    np.testing.assert_allclose(c.cpu().numpy(), ref.numpy(), atol=tol)
    # transposed views of both operands (strides, not copies)
    at = g(a.t().contiguous(), dev).t()
    bt = g(b.t().contiguous(), dev).t()
    hm.gemm(1.0, at, bt, 0.0, c)
    np.testing.assert_allclose(c.cpu().numpy(), ref.numpy(), atol=tol)
    hm.gemm(1.0, at, g(b, dev), 0.0, c)
    np.testing.assert_allclose(c.cpu().numpy(), ref.numpy(), atol=tol)
    # alpha / beta
    c0 = rnd(M, N, seed=22)
    c = g(c0, dev).clone()
    hm.gemm(0.5, g(a, dev), g(b, dev), 2.0, c)
    np.testing.assert_allclose(c.cpu().numpy(),
                               (0.5 * ref + 2.0 * c0.double()).numpy(),
                               atol=tol)


def test_gemm_identity_asymmetric(maths):
    """A = I with an asymmetric B catches swapped row / column maps."""
    hm, cm, dev = maths
    n = 96
    eye = torch.eye(n)
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 17 + \
        torch.arange(n).reshape(n, 1) * 0.5
    c = torch.empty(n, n, device=dev.torch_device)
    hm.gemm(1.0, g(eye, dev), g(b, dev), 0.0, c)
    assert torch.equal(c.cpu(), b)


def test_gemm_batched_and_gemv(maths):
    hm, cm, dev = maths
    a, b = rnd(12, 256, 64, seed=23), rnd(12, 64, 256, seed=24)
    c = torch.empty(12, 256, 256, device=dev.torch_device)
    hm.gemm(1.0, g(a, dev), g(b, dev), 0.0, c)
    np.testing.assert_allclose(c.cpu().numpy(), torch.matmul(a, b).numpy(),
                               atol=2e-4)
    m, v = rnd(70, 50, seed=25), rnd(50, seed=26)
    o = torch.empty(70, device=dev.torch_device)
    hm.gemv(1.0, g(m, dev), g(v, dev), 0.0, o)
    np.testing.assert_allclose(o.cpu().numpy(), (m @ v).numpy(), atol=1e-4)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_half(maths, dt):
    hm, cm, dev = maths
    a, b = rnd(130, 200, seed=27).to(dt), rnd(200, 96, seed=28).to(dt)
    ref = a.float() @ b.float()
    c = torch.empty(130, 96, dtype=dt, device=dev.torch_device)
    hm.gemm(1.0, g(a, dev), g(b, dev), 0.0, c)
    tol = 0.02 if dt == torch.float16 else 0.15
    np.testing.assert_allclose(c.cpu().float().numpy(), ref.numpy(), atol=tol)
    ct = torch.empty(130, 96, dtype=dt, device=dev.torch_device)
    hm.gemm(1.0, g(a.t().contiguous(), dev).t(),
            g(b.t().contiguous(), dev).t(), 0.0, ct)
    np.testing.assert_allclose(ct.cpu().float().numpy(), ref.numpy(), atol=tol)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True),
                                   (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (200, 136, 72),
                                   (128, 384, 1000), (64, 64, 64),
                                   (32, 1000, 256), (8, 24, 64)])
def test_gemm_16bit_tile_path(maths, dt, ta, tb, M, N, K):
    """128 x 128 x 64 kernel: every combination of row- / column-contiguous
    operands (16-byte row reads and transposing LDS reads), ragged M, N and
    reduction tails, against an fp32 product of the same rounded inputs.
    Small integers keep every product and partial sum exact in float."""
    hm, cm, dev = maths
    r = np.random.RandomState(M + N + K)
    a = torch.from_numpy(r.randint(-3, 4, (M, K)).astype(np.float32)).to(dt)
    b = torch.from_numpy(r.randint(-3, 4, (K, N)).astype(np.float32)).to(dt)
    # an identity-like marker so a swapped row / column map cannot cancel out
    a[:, 0] = torch.arange(M).to(dt) % 5
    b[0, :] = torch.arange(N).to(dt) % 7
    ref = (a.float() @ b.float())
    ag = g(a.t().contiguous(), dev).t() if ta else g(a, dev)
    bg = g(b.t().contiguous(), dev).t() if tb else g(b, dev)
    c = torch.empty(M, N, dtype=dt, device=dev.torch_device)
    hm.gemm(1.0, ag, bg, 0.0, c)
    want = ref.to(dt).float()     # only the final rounding differs
    assert torch.equal(c.float().cpu(), want)
    # alpha, beta, bias and the ReLU epilogue
    bias = torch.from_numpy(r.randint(-2, 3, (N,)).astype(np.float32)).to(dt)
    c0 = torch.from_numpy(r.randint(-2, 3, (M, N)).astype(np.float32)).to(dt)
    c = g(c0, dev).clone()
    hm.gemm(0.5, ag, bg, 2.0, c, bias=g(bias, dev), act=1)
    want = torch.relu(0.5 * ref + 2.0 * c0.float() + bias.float()).to(dt)
    assert torch.equal(c.float().cpu(), want.float())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_16bit_batched(maths, dt):
    hm, cm, dev = maths
    a, b = rnd(6, 192, 128, seed=61).to(dt), rnd(6, 128, 256, seed=62).to(dt)
    ref = torch.matmul(a.float(), b.float())
    c = torch.empty(6, 192, 256, dtype=dt, device=dev.torch_device)
    hm.gemm(1.0, g(a, dev), g(b, dev), 0.0, c)
    tol = 0.03 if dt == torch.float16 else 0.25
    np.testing.assert_allclose(c.float().cpu().numpy(), ref.numpy(), atol=tol)
    # attention-style: batch of Q K^T with K given as a transposed view
    kt = g(b.transpose(1, 2).contiguous(), dev).transpose(1, 2)
    hm.gemm(1.0, g(a, dev), kt, 0.0, c)
    np.testing.assert_allclose(c.float().cpu().numpy(), ref.numpy(), atol=tol)


CONV_CASES = [
    # N, C, H, W, K, R, S, stride, pad, groups
    (2, 3, 32, 32, 16, 7, 7, 2, 3, 1),
    (4, 64, 14, 14, 64, 3, 3, 1, 1, 1),
    (3, 32, 15, 15, 48, 3, 3, 2, 1, 1),
    (2, 128, 9, 9, 256, 1, 1, 1, 0, 1),
    (2, 64, 12, 12, 128, 1, 1, 2, 0, 1),
    (2, 16, 10, 11, 32, 3, 2, 1, 1, 4),
    (2, 8, 9, 9, 8, 3, 3, 1, 1, 8),
    (1, 20, 8, 8, 50, 5, 5, 1, 0, 1),
    (5, 130, 7, 7, 200, 3, 3, 1, 1, 1),
    # 4-wide staged 1x1 path, small and 128x128 tiles
    (2, 128, 8, 8, 256, 1, 1, 1, 0, 1),
    (3, 64, 6, 10, 36, 1, 1, 1, 0, 2),
    (16, 128, 36, 36, 256, 1, 1, 1, 0, 1),
    # 128x128 tiles on the gather path
    (32, 128, 36, 36, 128, 3, 3, 1, 1, 1),
    (32, 128, 36, 36, 128, 3, 3, 2, 1, 1),
    # stride 3: nine stride classes for backward-data, some without taps
    (2, 8, 13, 11, 12, 5, 5, 3, 2, 1),
    (2, 16, 9, 9, 8, 2, 2, 3, 0, 1),
    # more than 63 taps: compare-based bounds tests; 16 stride classes
    (1, 4, 12, 12, 8, 9, 9, 1, 4, 1),
    (1, 3, 35, 35, 8, 11, 11, 4, 2, 1),
    # few tiles, long reduction: split across workgroups in every pass
    (2, 512, 7, 7, 512, 3, 3, 1, 1, 1),
    # depthwise (groups == C == K): kernels of their own; the 4x4 filter
    # gradient falls back to the grouped implicit GEMM
    (4, 32, 19, 19, 32, 3, 3, 1, 1, 32),
    (3, 24, 17, 15, 24, 3, 3, 2, 1, 24),
    (2, 16, 14, 14, 16, 5, 5, 1, 2, 16),
    (2, 12, 21, 21, 12, 7, 7, 2, 3, 12),
    (2, 6, 10, 10, 6, 4, 4, 1, 1, 6),
    (2, 1024, 5, 5, 1024, 3, 3, 1, 1, 1024),
    (16, 728, 19, 19, 728, 3, 3, 1, 1, 728),
    # stride-1 3x3 depthwise staged through LDS: row bands of a large plane,
    # no padding, padding wider than "same"
    (1, 4, 150, 150, 4, 3, 3, 1, 1, 4),
    (2, 5, 40, 33, 5, 3, 3, 1, 0, 5),
    (1, 3, 12, 12, 3, 3, 3, 1, 2, 3),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_all_passes(maths, case):
    hm, cm, dev = maths
    N, C, H, W, K, R, S, st, pd, grp = case
    x = rnd(N, C, H, W, seed=30)
    w = rnd(K, C // grp, R, S, seed=31)
    b = rnd(K, seed=32)
    P = (H + 2 * pd - R) // st + 1
    Q = (W + 2 * pd - S) // st + 1
    ry = torch.empty(N, K, P, Q)
    cm.conv_fwd(x, w, b, ry, (st, st), (pd, pd), grp)
    y = torch.empty(N, K, P, Q, device=dev.torch_device)
    hm.conv_fwd(g(x, dev), g(w, dev), g(b, dev), y, (st, st), (pd, pd), grp)
    tol = 3e-6 * (C // grp) * R * S
    np.testing.assert_allclose(y.cpu().numpy(), ry.numpy(), atol=tol, rtol=1e-5)
    dy = rnd(N, K, P, Q, seed=33)
    rdx = torch.empty_like(x)
    cm.conv_bwd_data(dy, w, rdx, (st, st), (pd, pd), grp)
    dx = torch.empty_like(x, device=dev.torch_device)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (st, st), (pd, pd), grp)
    np.testing.assert_allclose(dx.cpu().numpy(), rdx.numpy(),
                               atol=3e-6 * (K // grp) * R * S, rtol=1e-5)
    rdw = torch.empty_like(w)
    cm.conv_bwd_weight(dy, x, rdw, (st, st), (pd, pd), grp)
    dw = torch.empty_like(w, device=dev.torch_device)
    hm.conv_bwd_weight(g(dy, dev), g(x, dev), dw, (st, st), (pd, pd), grp)
    np.testing.assert_allclose(dw.cpu().numpy(), rdw.numpy(),
                               atol=3e-6 * N * P * Q, rtol=1e-4)
    rdb = torch.empty(K)
    cm.conv_bwd_bias(dy, rdb)
    db = torch.empty(K, device=dev.torch_device)
    hm.conv_bwd_bias(g(dy, dev), db)
    np.testing.assert_allclose(db.cpu().numpy(), rdb.numpy(), atol=1e-4,
                               rtol=1e-4)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_all_passes_16bit(maths, case, dt):
    """16-bit inputs run on the 32x32x16 MFMA path (float accumulation, split
    reductions staged through a float scratch); the reference is the fp32
    result on the same rounded inputs, so only the final rounding differs."""
    hm, cm, dev = maths
    N, C, H, W, K, R, S, st, pd, grp = case
    x = rnd(N, C, H, W, seed=30).to(dt)
    w = rnd(K, C // grp, R, S, seed=31).to(dt)
    b = rnd(K, seed=32).to(dt)
    P = (H + 2 * pd - R) // st + 1
    Q = (W + 2 * pd - S) // st + 1
    eps = 2 ** -10 if dt == torch.float16 else 2 ** -7

    def close(got, ref):
        ref = ref.numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got.float().cpu().numpy(), ref,
                                   atol=eps * scale, rtol=2 * eps)

    ry = torch.empty(N, K, P, Q)
    cm.conv_fwd(x.float(), w.float(), b.float(), ry, (st, st), (pd, pd), grp)
    y = torch.empty(N, K, P, Q, device=dev.torch_device, dtype=dt)
    hm.conv_fwd(g(x, dev), g(w, dev), g(b, dev), y, (st, st), (pd, pd), grp)
    close(y, ry)
    dy = rnd(N, K, P, Q, seed=33).to(dt)
    add = rnd(N, C, H, W, seed=34).to(dt)
    rdx = torch.empty(N, C, H, W)
    cm.conv_bwd_data(dy.float(), w.float(), rdx, (st, st), (pd, pd), grp)
    dx = torch.empty_like(x, device=dev.torch_device)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (st, st), (pd, pd), grp)
    close(dx, rdx)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (st, st), (pd, pd), grp,
                     g(add, dev))
    close(dx, rdx + add.float())
    rdw = torch.empty(K, C // grp, R, S)
    cm.conv_bwd_weight(dy.float(), x.float(), rdw, (st, st), (pd, pd), grp)
    dw = torch.empty_like(w, device=dev.torch_device)
    hm.conv_bwd_weight(g(dy, dev), g(x, dev), dw, (st, st), (pd, pd), grp)
    close(dw, rdw)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W,K", [
    (2, 64, 8, 8, 256),      # backward-data with 64 rows: 64 x 128 tiles
    (3, 256, 14, 14, 64),    # forward with 64 rows; rows 4-byte aligned only
    (2, 128, 12, 20, 128),   # 128 x 128 tiles, ragged pixel tile
    (2, 64, 7, 7, 64),       # odd row length: implicit-GEMM kernel instead
])
def test_conv1x1_16bit_gemm_path(maths, dt, N, C, H, W, K):
    """16-bit 1x1 / stride 1 convolutions without bias run as one tiled GEMM
    per image (forward and backward-data, with and without the fused
    addend); reference: fp32 convolution on the same rounded inputs."""
    hm, cm, dev = maths
    x = rnd(N, C, H, W, seed=40).to(dt)
    w = rnd(K, C, 1, 1, seed=41).to(dt)
    eps = 2 ** -10 if dt == torch.float16 else 2 ** -7

    def close(got, ref):
        ref = ref.numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got.float().cpu().numpy(), ref,
                                   atol=eps * scale, rtol=2 * eps)

    ry = torch.empty(N, K, H, W)
    cm.conv_fwd(x.float(), w.float(), None, ry, (1, 1), (0, 0), 1)
    y = torch.empty(N, K, H, W, device=dev.torch_device, dtype=dt)
    hm.conv_fwd(g(x, dev), g(w, dev), None, y, (1, 1), (0, 0), 1)
    close(y, ry)
    dy = rnd(N, K, H, W, seed=42).to(dt)
    add = rnd(N, C, H, W, seed=43).to(dt)
    rdx = torch.empty(N, C, H, W)
    cm.conv_bwd_data(dy.float(), w.float(), rdx, (1, 1), (0, 0), 1)
    dx = torch.empty_like(x, device=dev.torch_device)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (1, 1), (0, 0), 1)
    close(dx, rdx)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (1, 1), (0, 0), 1,
                     g(add, dev))
    close(dx, rdx + add.float())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,C,K,H,W", [(2, 64, 64, 56, 56), (2, 128, 96, 28, 28),
                                       (3, 64, 256, 14, 14), (1, 192, 40, 9, 12),
                                       (2, 64, 64, 7, 8)])
def test_conv3x3_16bit_tap_gemm_path(maths, dt, N, C, K, H, W):
    """16-bit 3x3 / stride 1 / pad 1 convolutions without bias run as nine
    shifted GEMMs per image on the tiled 16-bit kernel (forward and
    backward-data, with and without the fused addend); reference: fp32
    convolution on the same rounded inputs."""
    hm, cm, dev = maths
    x = rnd(N, C, H, W, seed=60).to(dt)
    w = rnd(K, C, 3, 3, seed=61, lo=-0.3, hi=0.3).to(dt)
    eps = 2 ** -10 if dt == torch.float16 else 2 ** -7

    def close(got, ref):
        ref = ref.numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got.float().cpu().numpy(), ref,
                                   atol=eps * scale, rtol=2 * eps)

    ry = torch.empty(N, K, H, W)
    cm.conv_fwd(x.float(), w.float(), None, ry, (1, 1), (1, 1), 1)
    y = torch.empty(N, K, H, W, device=dev.torch_device, dtype=dt)
    assert hm._conv3x3_h16(0, g(x, dev), g(w, dev), None, y)
    close(y, ry)
    y.zero_()
    hm.conv_fwd(g(x, dev), g(w, dev), None, y, (1, 1), (1, 1), 1)
    close(y, ry)
    dy = rnd(N, K, H, W, seed=62).to(dt)
    add = rnd(N, C, H, W, seed=63).to(dt)
    rdx = torch.empty(N, C, H, W)
    cm.conv_bwd_data(dy.float(), w.float(), rdx, (1, 1), (1, 1), 1)
    dx = torch.empty_like(x, device=dev.torch_device)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (1, 1), (1, 1), 1)
    close(dx, rdx)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (1, 1), (1, 1), 1,
                     g(add, dev))
    close(dx, rdx + add.float())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,C,K,H,W,R", [
    (5, 64, 96, 14, 14, 3),     # 196 positions: half-filled last piece
    (6, 128, 64, 7, 7, 3),      # odd image size: per-piece parity
    (3, 64, 200, 7, 7, 1),      # 1x1 on an odd image size
    (2, 192, 72, 5, 9, 3),      # odd row length with an even-sized... image
    (40, 64, 64, 14, 14, 1),    # 1x1 over many small images, 64-row tiles
    (33, 128, 256, 14, 14, 3),  # enough columns for the 128 x 128 tile
    (2, 64, 40, 28, 28, 3),     # larger image through the same kernel
])
def test_conv_16bit_flattened_positions_path(maths, dt, N, C, K, H, W, R,
                                             monkeypatch):
    """16-bit 1x1 and 3x3 / pad 1 stride-1 convolutions over the flattened
    positions of all images (forward, backward-data with and without the fused
    addend), including odd image sizes; reference: fp32 convolution on the
    same rounded inputs."""
    from singa_amd.core import math_hip
    # the routing threshold keeps tiny 3x3 images off this kernel for speed
    monkeypatch.setattr(math_hip, "_CONV_FLAT_MIN_HW_3X3", 0)
    hm, cm, dev = maths
    pad = (R // 2, R // 2)
    x = rnd(N, C, H, W, seed=80).to(dt)
    w = rnd(K, C, R, R, seed=81, lo=-0.3, hi=0.3).to(dt)
    eps = 2 ** -10 if dt == torch.float16 else 2 ** -7

    def close(got, ref):
        ref = ref.numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got.float().cpu().numpy(), ref,
                                   atol=eps * scale, rtol=2 * eps)

    ry = torch.empty(N, K, H, W)
    cm.conv_fwd(x.float(), w.float(), None, ry, (1, 1), pad, 1)
    y = torch.zeros(N, K, H, W, device=dev.torch_device, dtype=dt)
    assert hm._flat16(0, g(x, dev), g(w, dev), None, y, (1, 1), pad, 1)
    close(y, ry)
    dy = rnd(N, K, H, W, seed=82).to(dt)
    add = rnd(N, C, H, W, seed=83).to(dt)
    rdx = torch.empty(N, C, H, W)
    cm.conv_bwd_data(dy.float(), w.float(), rdx, (1, 1), pad, 1)
    dx = torch.zeros(N, C, H, W, device=dev.torch_device, dtype=dt)
    if K % 64 == 0:
        assert hm._flat16(1, g(dy, dev), g(w, dev), None, dx, (1, 1), pad, 1)
        close(dx, rdx)
        assert hm._flat16(1, g(dy, dev), g(w, dev), g(add, dev), dx, (1, 1),
                          pad, 1)
        close(dx, rdx + add.float())
    else:
        assert not hm._flat16(1, g(dy, dev), g(w, dev), None, dx, (1, 1), pad,
                              1)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (1, 1), pad, 1, g(add, dev))
    close(dx, rdx + add.float())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,C,K,H,W", [(3, 64, 128, 28, 28), (2, 128, 64, 56, 56),
                                       (4, 64, 72, 14, 14), (2, 64, 64, 12, 20)])
def test_conv1x1_stride2_16bit_gathered(maths, dt, N, C, K, H, W):
    """16-bit 1x1 / stride 2 convolutions run as stride-1 problems on the
    gathered even positions (forward, weight gradient) and scatter the data
    gradient back, fused with the addend; reference: fp32 convolution."""
    hm, cm, dev = maths
    x = rnd(N, C, H, W, seed=110).to(dt)
    w = rnd(K, C, 1, 1, seed=111, lo=-0.3, hi=0.3).to(dt)
    assert hm._strided_pointwise16(g(x, dev), g(w, dev), (2, 2), (0, 0), 1)
    eps = 2 ** -10 if dt == torch.float16 else 2 ** -7

    def close(got, ref):
        ref = ref.numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got.float().cpu().numpy(), ref,
                                   atol=eps * scale, rtol=2 * eps)

    P, Q = H // 2, W // 2
    ry = torch.empty(N, K, P, Q)
    cm.conv_fwd(x.float(), w.float(), None, ry, (2, 2), (0, 0), 1)
    y = torch.zeros(N, K, P, Q, device=dev.torch_device, dtype=dt)
    hm.conv_fwd(g(x, dev), g(w, dev), None, y, (2, 2), (0, 0), 1)
    close(y, ry)
    dy = rnd(N, K, P, Q, seed=112).to(dt)
    add = rnd(N, C, H, W, seed=113).to(dt)
    rdx = torch.empty(N, C, H, W)
    cm.conv_bwd_data(dy.float(), w.float(), rdx, (2, 2), (0, 0), 1)
    dx = torch.full((N, C, H, W), 7.0, device=dev.torch_device, dtype=dt)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (2, 2), (0, 0), 1)
    close(dx, rdx)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (2, 2), (0, 0), 1,
                     g(add, dev))
    close(dx, rdx + add.float())
    rdw = torch.empty(K, C, 1, 1)
    cm.conv_bwd_weight(dy.float(), x.float(), rdw, (2, 2), (0, 0), 1)
    dw = torch.zeros(K, C, 1, 1, device=dev.torch_device, dtype=dt)
    hm.conv_bwd_weight(g(dy, dev), g(x, dev), dw, (2, 2), (0, 0), 1)
    close(dw, rdw)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,C,K,H,W,R", [
    (2, 64, 64, 56, 56, 3),     # 64 x 64 tiles, whole pieces
    (3, 128, 96, 28, 28, 3),    # 128 x 128 tiles with a ragged row tile
    (5, 72, 256, 14, 14, 3),    # 196 positions: half-filled last piece
    (2, 40, 64, 10, 12, 3),     # pieces spanning two image rows, odd counts
    (3, 64, 256, 14, 14, 1),    # 1x1: 128 x 64 tiles
    (2, 256, 64, 28, 28, 1),    # 1x1: 64 x 128 tiles
    (4, 136, 200, 9, 8, 1),     # 1x1: ragged tiles in both directions
    (6, 64, 128, 7, 7, 3),      # odd image size: rows at either parity
    (5, 72, 64, 7, 7, 1),       # 1x1 on an odd image size
    (2, 64, 64, 9, 7, 3),       # 63 positions, 7 columns
    (2, 40, 64, 6, 9, 3),       # odd row length, even image size
])
def test_conv_weight_gradient_16bit_gemm_path(maths, dt, N, C, K, H, W, R):
    """16-bit weight gradients of 1x1 and 3x3 / pad 1 stride-1 convolutions
    run on the tiled 16-bit kernel with the reduction over positions split
    across workgroups; reference: fp32 weight gradient on the same rounded
    inputs."""
    hm, cm, dev = maths
    pad = (R // 2, R // 2)
    x = rnd(N, C, H, W, seed=70).to(dt)
    dy = rnd(N, K, H, W, seed=71).to(dt)
    rdw = torch.empty(K, C, R, R)
    cm.conv_bwd_weight(dy.float(), x.float(), rdw, (1, 1), pad, 1)
    dw = torch.zeros(K, C, R, R, device=dev.torch_device, dtype=dt)
    assert hm._wgrad16(g(dy, dev), g(x, dev), dw, (1, 1), pad, 1)
    eps = 2 ** -10 if dt == torch.float16 else 2 ** -7
    ref = rdw.numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(dw.float().cpu().numpy(), ref,
                               atol=eps * scale, rtol=2 * eps)
    dw.zero_()
    hm.conv_bwd_weight(g(dy, dev), g(x, dev), dw, (1, 1), pad, 1)
    np.testing.assert_allclose(dw.float().cpu().numpy(), ref,
                               atol=eps * scale, rtol=2 * eps)


@pytest.mark.parametrize("N,C,K,HW", [(2, 16, 64, 56), (3, 24, 96, 28),
                                      (2, 64, 40, 14), (1, 8, 128, 56),
                                      (2, 72, 64, 28)])
def test_conv3x3_float32_halo_path(maths, N, C, K, HW):
    """float32 3x3 / stride 1 / pad 1 convolutions on 56 / 28 / 14 pixel rows
    run on the direct halo-tile kernel (forward with bias, backward-data with
    and without the fused addend); channel counts exercise several 8-channel
    chunks and a partial 64-channel output tile."""
    hm, cm, dev = maths
    x = rnd(N, C, HW, HW, seed=50)
    w = rnd(K, C, 3, 3, seed=51, lo=-0.3, hi=0.3)
    b = rnd(K, seed=52)

    def close(got, ref):
        ref = ref.numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got.cpu().numpy(), ref, atol=2e-5 * scale,
                                   rtol=1e-4)

    ry = torch.empty(N, K, HW, HW)
    cm.conv_fwd(x, w, b, ry, (1, 1), (1, 1), 1)
    y = torch.empty(N, K, HW, HW, device=dev.torch_device)
    hm.conv_fwd(g(x, dev), g(w, dev), g(b, dev), y, (1, 1), (1, 1), 1)
    close(y, ry)
    dy = rnd(N, K, HW, HW, seed=53)
    add = rnd(N, C, HW, HW, seed=54)
    rdx = torch.empty(N, C, HW, HW)
    cm.conv_bwd_data(dy, w, rdx, (1, 1), (1, 1), 1)
    dx = torch.empty(N, C, HW, HW, device=dev.torch_device)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (1, 1), (1, 1), 1)
    close(dx, rdx)
    hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (1, 1), (1, 1), 1, g(add, dev))
    close(dx, rdx + add)


@pytest.mark.parametrize("shape", [(8, 16, 14, 14), (32, 64, 7, 7), (16, 10),
                                   (4, 3, 33, 17), (32, 8, 28, 28),
                                   (40, 6, 56, 56), (9, 5, 61, 67)])
def test_batchnorm(maths, shape):
    """Channels that fit one workgroup's registers run the one-kernel path
    (up to 7168 pieces: the first five shapes), larger ones the two split
    passes ((40, 6, 56, 56) four values per access, (9, 5, 61, 67) scalar)."""
    hm, cm, dev = maths
    C = shape[1]
    x = rnd(*shape, seed=40, lo=-2, hi=5)
    scale, bias = rnd(C, seed=41, lo=0.5, hi=1.5), rnd(C, seed=42)
    rm, rv = rnd(C, seed=43), rnd(C, seed=44, lo=0.5, hi=2.0)
    rm_r, rv_r = rm.clone(), rv.clone()
    ry, rmean, ristd = torch.empty_like(x), torch.empty(C), torch.empty(C)
    cm.bn_fwd_train(x, scale, bias, rm_r, rv_r, 0.1, 1e-5, ry, rmean, ristd)
    d = dev.torch_device
    rm_g, rv_g = g(rm, dev), g(rv, dev)
    y = torch.empty_like(x, device=d)
    mean, istd = torch.empty(C, device=d), torch.empty(C, device=d)
    hm.bn_fwd_train(g(x, dev), g(scale, dev), g(bias, dev), rm_g, rv_g, 0.1,
                    1e-5, y, mean, istd)
    np.testing.assert_allclose(y.cpu().numpy(), ry.numpy(), atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(mean.cpu().numpy(), rmean.numpy(), atol=1e-5)
    np.testing.assert_allclose(istd.cpu().numpy(), ristd.numpy(), rtol=1e-4)
    np.testing.assert_allclose(rm_g.cpu().numpy(), rm_r.numpy(), atol=1e-5)
    np.testing.assert_allclose(rv_g.cpu().numpy(), rv_r.numpy(), rtol=1e-4)
    dy = rnd(*shape, seed=45)
    rdx, rds, rdb = torch.empty_like(x), torch.empty(C), torch.empty(C)
    cm.bn_bwd(dy, x, scale, rmean, ristd, rdx, rds, rdb)
    dx = torch.empty_like(x, device=d)
    ds, db = torch.empty(C, device=d), torch.empty(C, device=d)
    hm.bn_bwd(g(dy, dev), g(x, dev), g(scale, dev), mean, istd, dx, ds, db)
    np.testing.assert_allclose(dx.cpu().numpy(), rdx.numpy(), atol=5e-5,
                               rtol=1e-3)
    np.testing.assert_allclose(ds.cpu().numpy(), rds.numpy(), atol=1e-3,
                               rtol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), rdb.numpy(), atol=1e-3,
                               rtol=1e-4)
    riy = torch.empty_like(x)
    cm.bn_fwd_infer(x, scale, bias, rm, rv, 1e-5, riy)
    iy = torch.empty_like(x, device=d)
    hm.bn_fwd_infer(g(x, dev), g(scale, dev), g(bias, dev), g(rm, dev),
                    g(rv, dev), 1e-5, iy)
    np.testing.assert_allclose(iy.cpu().numpy(), riy.numpy(), atol=2e-5,
                               rtol=1e-4)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_batchnorm_two_sweeps_per_channel(maths, act):
    """256 channels of 40 x 28 x 28 values each: too many for a workgroup's
    registers, enough channels for one workgroup per CU -- the kernels that
    sweep a channel twice (statistics, then apply), with every activation."""
    hm, cm, dev = maths
    shape, C = (40, 256, 28, 28), 256
    x = rnd(*shape, seed=50, lo=-2, hi=5)
    res = rnd(*shape, seed=56) if act == 2 else None
    scale, bias = rnd(C, seed=51, lo=0.5, hi=1.5), rnd(C, seed=52)
    rm, rv = rnd(C, seed=53), rnd(C, seed=54, lo=0.5, hi=2.0)
    rm_r, rv_r = rm.clone(), rv.clone()
    ry, rmean, ristd = torch.empty_like(x), torch.empty(C), torch.empty(C)
    cm.bn_fwd_train(x, scale, bias, rm_r, rv_r, 0.1, 1e-5, ry, rmean, ristd,
                    act, res)
    d = dev.torch_device
    rm_g, rv_g = g(rm, dev), g(rv, dev)
    y = torch.empty_like(x, device=d)
    mean, istd = torch.empty(C, device=d), torch.empty(C, device=d)
    hm.bn_fwd_train(g(x, dev), g(scale, dev), g(bias, dev), rm_g, rv_g, 0.1,
                    1e-5, y, mean, istd, act,
                    g(res, dev) if res is not None else None)
    np.testing.assert_allclose(y.cpu().numpy(), ry.numpy(), atol=5e-5,
                               rtol=1e-4)
    np.testing.assert_allclose(mean.cpu().numpy(), rmean.numpy(), atol=1e-5)
    np.testing.assert_allclose(istd.cpu().numpy(), ristd.numpy(), rtol=1e-4)
    np.testing.assert_allclose(rm_g.cpu().numpy(), rm_r.numpy(), atol=1e-5)
    np.testing.assert_allclose(rv_g.cpu().numpy(), rv_r.numpy(), rtol=1e-4)
    dy = rnd(*shape, seed=55)
    rdx, rds, rdb = torch.empty_like(x), torch.empty(C), torch.empty(C)
    rdres = torch.empty_like(x) if act == 2 else None
    cm.bn_bwd(dy, x, scale, rmean, ristd, rdx, rds, rdb, act, bias, ry, rdres)
    dx = torch.empty_like(x, device=d)
    dres = torch.empty_like(x, device=d) if act == 2 else None
    ds, db = torch.empty(C, device=d), torch.empty(C, device=d)
    hm.bn_bwd(g(dy, dev), g(x, dev), g(scale, dev), mean, istd, dx, ds, db,
              act, g(bias, dev), y, dres)
    # an element whose activation input is within rounding of zero may fall
    # on the other side of the ReLU than on the host: compare all but a few
    bad = ~np.isclose(dx.cpu().numpy(), rdx.numpy(), atol=1e-4, rtol=1e-3)
    assert bad.mean() < 1e-4, bad.sum()
    np.testing.assert_allclose(ds.cpu().numpy(), rds.numpy(), atol=5e-2,
                               rtol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), rdb.numpy(), atol=5e-2,
                               rtol=1e-3)
    if act == 2:
        bad = ~np.isclose(dres.cpu().numpy(), rdres.numpy(), atol=1e-5)
        assert bad.mean() < 1e-4, bad.sum()


def test_conv_backward_pair_while_plans_are_measured(maths, monkeypatch):
    """The data-gradient launch clears the weight gradient for the launch
    that follows.  When the weight-gradient launch plan of a shape is still to
    be chosen, timing launches run in between and leave a full gradient in
    the buffer: the final launch must then start from zero again.  First call
    = plans measured, second = plans known; both must match the host."""
    from singa_amd.core import math_hip, nn_ops, tensor_core
    hm, cm, dev = maths
    monkeypatch.setattr(math_hip, "_CONV_AUTOTUNE", True)
    monkeypatch.setattr(type(hm), "_plan_file", ("", {}))
    saved = dict(hm._conv_plans)
    hm._conv_plans.clear()
    try:
        N, C, K, H = 4, 24, 40, 13            # a shape no other test uses
        x = rnd(N, C, H, H, seed=301)
        w = rnd(K, C, 3, 3, seed=302)
        dy = rnd(N, K, H, H, seed=303)
        rdx, rdw = torch.empty_like(x), torch.empty_like(w)
        cm.conv_bwd_data(dy, w, rdx, (1, 1), (1, 1), 1)
        cm.conv_bwd_weight(dy, x, rdw, (1, 1), (1, 1), 1)
        for attempt in ("measuring", "planned"):
            dx = torch.empty_like(x, device=dev.torch_device)
            dw = torch.full_like(w, 7.0, device=dev.torch_device)
            cleared = hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (1, 1),
                                       (1, 1), 1, side_zero=dw)
            hm.conv_bwd_weight(g(dy, dev), g(x, dev), dw, (1, 1), (1, 1), 1,
                               zeroed=bool(cleared))
            np.testing.assert_allclose(dx.cpu().numpy(), rdx.numpy(),
                                       atol=2e-3, rtol=1e-3, err_msg=attempt)
            np.testing.assert_allclose(dw.cpu().numpy(), rdw.numpy(),
                                       atol=5e-3, rtol=1e-3, err_msg=attempt)
    finally:
        hm._conv_plans.clear()
        hm._conv_plans.update(saved)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("shape", [(3, 16, 56, 56), (2, 24, 14, 14),
                                   (4, 32, 7, 7), (2, 5, 6, 6)])
def test_batchnorm_inference_vector_paths(maths, shape, act, dt):
    """Running-statistics normalisation with the fused ReLU / residual:
    planes of 3136 elements move 16 bytes per access, 196 move four elements,
    49 and 36 (odd / unaligned in 16 bits) take the scalar kernel."""
    hm, cm, dev = maths
    C = shape[1]
    d = dev.torch_device
    x = rnd(*shape, seed=140, lo=-2, hi=3)
    res = rnd(*shape, seed=141)
    scale, bias = rnd(C, seed=142, lo=0.5, hi=1.5), rnd(C, seed=143)
    rm, rv = rnd(C, seed=144), rnd(C, seed=145, lo=0.5, hi=2.0)
    xq, rq = x.to(dt), res.to(dt)
    want = (xq.float() - rm[None, :, None, None]) * (
        scale / torch.sqrt(rv + 1e-5))[None, :, None, None] + \
        bias[None, :, None, None]
    if act == 2:
        want = want + rq.float()
    if act >= 1:
        want = want.clamp_min(0)
    y = torch.empty(shape, dtype=dt, device=d)
    hm.bn_fwd_infer(xq.to(d), g(scale, dev), g(bias, dev), g(rm, dev),
                    g(rv, dev), 1e-5, y, act=act,
                    residual=rq.to(d) if act == 2 else None)
    tol = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}[dt]
    np.testing.assert_allclose(y.float().cpu().numpy(), want.numpy(),
                               atol=tol, rtol=tol)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 16, 14, 14), (4, 3, 33, 17),
                                   (40, 6, 56, 56), (9, 5, 61, 67)])
def test_batchnorm_16bit(maths, shape, dt):
    """16-bit activations, float parameters and statistics; (8, 16, 14, 14)
    takes the four-elements-per-access path, (4, 3, 33, 17) the scalar one."""
    hm, cm, dev = maths
    C = shape[1]
    d = dev.torch_device
    eps = 2 ** -9 if dt == torch.float16 else 2 ** -6
    x = rnd(*shape, seed=40, lo=-2, hi=5).to(dt)
    scale, bias = rnd(C, seed=41, lo=0.5, hi=1.5), rnd(C, seed=42)
    rm, rv = rnd(C, seed=43), rnd(C, seed=44, lo=0.5, hi=2.0)
    ry, rmean, ristd = torch.empty(shape), torch.empty(C), torch.empty(C)
    cm.bn_fwd_train(x.float(), scale, bias, rm.clone(), rv.clone(), 0.1, 1e-5,
                    ry, rmean, ristd)
    y = torch.empty(shape, device=d, dtype=dt)
    mean, istd = torch.empty(C, device=d), torch.empty(C, device=d)
    hm.bn_fwd_train(g(x, dev), g(scale, dev), g(bias, dev), g(rm, dev),
                    g(rv, dev), 0.1, 1e-5, y, mean, istd)
    np.testing.assert_allclose(y.float().cpu().numpy(), ry.numpy(),
                               atol=4 * eps, rtol=2 * eps)
    np.testing.assert_allclose(mean.cpu().numpy(), rmean.numpy(), atol=1e-4)
    dy = rnd(*shape, seed=45).to(dt)
    rdx, rds, rdb = torch.empty(shape), torch.empty(C), torch.empty(C)
    cm.bn_bwd(dy.float(), x.float(), scale, rmean, ristd, rdx, rds, rdb)
    dx = torch.empty(shape, device=d, dtype=dt)
    ds, db = torch.empty(C, device=d), torch.empty(C, device=d)
    hm.bn_bwd(g(dy, dev), g(x, dev), g(scale, dev), mean, istd, dx, ds, db)
    np.testing.assert_allclose(dx.float().cpu().numpy(), rdx.numpy(),
                               atol=2 * eps, rtol=2 * eps)
    np.testing.assert_allclose(ds.cpu().numpy(), rds.numpy(), atol=2e-2,
                               rtol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), rdb.numpy(), atol=2e-2,
                               rtol=1e-3)


@pytest.mark.parametrize("k,s,p,is_max", [(3, 2, 1, True), (2, 2, 0, True),
                                          (7, 1, 0, False), (3, 2, 1, False),
                                          (3, 1, 1, True)])
def test_pooling(maths, k, s, p, is_max):
    hm, cm, dev = maths
    x = rnd(3, 5, 15, 14, seed=50)
    P = (15 + 2 * p - k) // s + 1
    Q = (14 + 2 * p - k) // s + 1
    ry = torch.empty(3, 5, P, Q)
    cm.pool_fwd(x, ry, (k, k), (s, s), (p, p), is_max)
    y = torch.empty(3, 5, P, Q, device=dev.torch_device)
    hm.pool_fwd(g(x, dev), y, (k, k), (s, s), (p, p), is_max)
    np.testing.assert_allclose(y.cpu().numpy(), ry.numpy(), rtol=1e-6,
                               atol=1e-6)
    dy = rnd(3, 5, P, Q, seed=51)
    rdx = torch.empty_like(x)
    cm.pool_bwd(dy, x, ry, rdx, (k, k), (s, s), (p, p), is_max)
    dx = torch.empty_like(x, device=dev.torch_device)
    hm.pool_bwd(g(dy, dev), g(x, dev), y, dx, (k, k), (s, s), (p, p), is_max)
    np.testing.assert_allclose(dx.cpu().numpy(), rdx.numpy(), rtol=1e-5,
                               atol=1e-6)


@pytest.mark.parametrize("nesterov,mom", [(False, 0.0), (False, 0.9),
                                          (True, 0.9)])
def test_sgd_fused(maths, nesterov, mom):
    hm, cm, dev = maths
    p, gr, buf = rnd(1003, seed=60), rnd(1003, seed=61), rnd(1003, seed=62)
    hp = [torch.tensor([v]) for v in (0.1, mom, 0.0, 1e-3)]
    pr, br = p.clone(), buf.clone()
    cm.sgd_update(pr, gr, br if mom else None, hp[0], hp[1], hp[2], hp[3],
                  nesterov, 0.5)
    pg, bg = g(p, dev), g(buf, dev)
    hm.sgd_update(pg, g(gr, dev), bg if mom else None,
                  *[g(h, dev) for h in hp], nesterov, 0.5)
    np.testing.assert_allclose(pg.cpu().numpy(), pr.numpy(), rtol=1e-5,
                               atol=1e-7)
    if mom:
        np.testing.assert_allclose(bg.cpu().numpy(), br.numpy(), rtol=1e-5,
                                   atol=1e-7)


def test_index_ops(maths):
    hm, cm, dev = maths
    x = rnd(4, 9, 6, seed=70)
    idx = torch.tensor([3, 0, 8, 3, -1], dtype=torch.int32)
    ref = torch.index_select(x, 1, torch.tensor([3, 0, 8, 3, 8]))
    out = torch.empty(4, 5, 6, device=dev.torch_device)
    hm.index_select(g(x, dev), 1, g(idx, dev), out)
    assert torch.equal(out.cpu(), ref)
    dy = rnd(4, 5, 6, seed=71)
    rdx = torch.zeros(4, 9, 6).index_add_(1, torch.tensor([3, 0, 8, 3, 8]), dy)
    dx = torch.empty(4, 9, 6, device=dev.torch_device)
    hm.index_add(g(dy, dev), 1, g(idx, dev), dx)
    np.testing.assert_allclose(dx.cpu().numpy(), rdx.numpy(), rtol=1e-6,
                               atol=1e-6)


def test_lstm_cell(maths):
    hm, cm, dev = maths
    B, H = 7, 33
    gates, cp = rnd(B, 4 * H, seed=80, lo=-2, hi=2), rnd(B, H, seed=81)
    rc, rh, ra = torch.empty(B, H), torch.empty(B, H), torch.empty(B, 4 * H)
    cm.lstm_cell_fwd(gates, cp, rc, rh, ra)
    d = dev.torch_device
    c, h, a = (torch.empty(B, H, device=d), torch.empty(B, H, device=d),
               torch.empty(B, 4 * H, device=d))
    hm.lstm_cell_fwd(g(gates, dev), g(cp, dev), c, h, a)
    np.testing.assert_allclose(h.cpu().numpy(), rh.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c.cpu().numpy(), rc.numpy(), rtol=1e-5, atol=1e-6)
    dh, dc = rnd(B, H, seed=82), rnd(B, H, seed=83)
    rdg, rdcp = torch.empty(B, 4 * H), torch.empty(B, H)
    cm.lstm_cell_bwd(dh, dc, ra, cp, rc, rdg, rdcp)
    dg, dcp = torch.empty(B, 4 * H, device=d), torch.empty(B, H, device=d)
    hm.lstm_cell_bwd(g(dh, dev), g(dc, dev), a, g(cp, dev), c, dg, dcp)
    np.testing.assert_allclose(dg.cpu().numpy(), rdg.numpy(), rtol=1e-4,
                               atol=1e-6)
    np.testing.assert_allclose(dcp.cpu().numpy(), rdcp.numpy(), rtol=1e-4,
                               atol=1e-6)


@pytest.mark.parametrize("n,k", [(1000, 1), (1000, 37), (65537, 4096),
                                 (1 << 20, 10000), (300, 300)])
def test_topk_abs_radix_select(maths, n, k):
    """Device radix select against a sort on the host: the multiset of
    selected magnitudes must match, each value must sit at its index, and no
    index may repeat.  A block of equal magnitudes straddles the cut."""
    hm, cm, dev = maths
    x = rnd(n, seed=90, lo=-4, hi=4)
    x[::7] = 0.5          # many ties
    x[3] = -1e30
    x[5] = float("inf") if n > 5 else x[5]
    xg = g(x, dev)
    vals, idx = hm.topk_abs(xg, k)
    torch.cuda.synchronize()
    vals, idx = vals.cpu(), idx.cpu().long()
    assert idx.unique().numel() == k
    assert torch.equal(x[idx], vals)
    ref = torch.sort(x.abs(), descending=True).values[:k]
    got = torch.sort(vals.abs(), descending=True).values
    assert torch.equal(ref, got)


@pytest.mark.parametrize("M,N,K", [(32, 1000, 2048), (32, 2048, 1000),
                                   (8, 8, 4096), (100, 70, 1536)])
def test_gemm_skinny_split_reduction(maths, M, N, K):
    """Few output tiles and a long reduction: the product is computed as a
    batch of partial products over slices of K and summed by the epilogue
    kernel (alpha, beta, bias, ReLU applied there)."""
    hm, cm, dev = maths
    assert hm._gemm_splits(M, N, K) > 1
    a, b = rnd(M, K, seed=70), rnd(K, N, seed=71)
    bias, c0 = rnd(N, seed=72), rnd(M, N, seed=73)
    ref = a.double() @ b.double()
    c = torch.empty(M, N, device=dev.torch_device)
    hm.gemm(1.0, g(a, dev), g(b, dev), 0.0, c)
    np.testing.assert_allclose(c.cpu().numpy(), ref.numpy(), atol=2e-4)
    c = g(c0, dev).clone()
    hm.gemm(0.5, g(a.t().contiguous(), dev).t(), g(b, dev), 2.0, c,
            bias=g(bias, dev), act=1)
    want = torch.relu(0.5 * ref + 2.0 * c0.double() + bias.double())
    np.testing.assert_allclose(c.cpu().numpy(), want.numpy(), atol=3e-4)


@pytest.mark.parametrize("wire", [torch.float32, torch.float16])
def test_pack_unpack_bucket(maths, wire):
    """70 tensors (more than one launch's worth of entries) into one flat
    bucket and back, with conversion to the wire type and scaling on return."""
    hm, cm, dev = maths
    r = np.random.RandomState(5)
    sizes = [int(v) for v in r.randint(1, 5000, 70)]
    host = [rnd(n, seed=100 + i) for i, n in enumerate(sizes)]
    views = [g(t, dev) for t in host]
    total = sum(sizes)
    bucket = torch.full((total + 16,), 7.0, dtype=wire, device=dev.torch_device)
    plan = hm.pack_plan(views)
    assert plan[2] == total
    hm.pack(plan, views, bucket[:total])
    flat = torch.cat(host).to(wire)
    assert torch.equal(bucket[:total].cpu(), flat)
    assert torch.all(bucket[total:].cpu() == 7.0)
    outs = [torch.zeros_like(v) for v in views]
    hm.unpack(hm.pack_plan(outs), bucket[:total], outs, 0.5)
    for o, t in zip(outs, host):
        np.testing.assert_allclose(o.cpu().numpy(),
                                   (t.to(wire).float() * 0.5).numpy(),
                                   rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(32, 1000, 2048), (1024, 1024, 4096),
                                   (256, 384, 1024)])
def test_gemm_16bit_split_reduction(maths, dt, M, N, K):
    """Few output tiles, long reduction, 16-bit operands: float partial
    products per slice of K, one epilogue.  Integer-valued inputs keep the
    float sums exact, so only the final rounding to 16 bits differs."""
    hm, cm, dev = maths
    assert hm._gemm_splits(M, N, K) > 1
    r = np.random.RandomState(K)
    a = torch.from_numpy(r.randint(-2, 3, (M, K)).astype(np.float32)).to(dt)
    b = torch.from_numpy(r.randint(-2, 3, (K, N)).astype(np.float32)).to(dt)
    bias = torch.from_numpy(r.randint(-2, 3, (N,)).astype(np.float32)).to(dt)
    ref = a.float() @ b.float()
    c = torch.empty(M, N, dtype=dt, device=dev.torch_device)
    hm.gemm(1.0, g(a, dev), g(b, dev), 0.0, c)
    assert torch.equal(c.float().cpu(), ref.to(dt).float())
    hm.gemm(0.25, g(a, dev), g(b.t().contiguous(), dev).t(), 0.0, c,
            bias=g(bias, dev), act=1)
    want = torch.relu(0.25 * ref + bias.float()).to(dt).float()
    assert torch.equal(c.float().cpu(), want)


def test_byte_copy_of_unsupported_dtypes(maths):
    """uint8 / int64 tensors have no arithmetic kernels; copies go through the
    byte-copy kernel (odd sizes exercise the tail)."""
    hm, cm, dev = maths
    for dt, n in ((torch.uint8, 1003), (torch.int64, 77), (torch.uint8, 4096)):
        src = torch.randint(0, 200, (n,), dtype=dt).to(dev.torch_device)
        dst = torch.zeros_like(src)
        hm.copy(src, dst)
        assert torch.equal(src.cpu(), dst.cpu())


def test_conv_plans_persist_and_reload(maths, tmp_path, monkeypatch):
    """A measured launch plan is written to the plan file and a fresh cache
    (a new process) takes it from there instead of measuring again."""
    hm, cm, dev = maths
    from singa_amd.core import math_hip
    path = str(tmp_path / "plans.json")
    monkeypatch.setenv("SINGA_CONV_PLAN_CACHE", path)
    monkeypatch.setattr(math_hip.HipMath, "_plan_file", None)
    monkeypatch.setattr(math_hip.HipMath, "_conv_plans", {})
    x = g(rnd(2, 16, 12, 12, seed=70), dev)
    w = g(rnd(32, 16, 3, 3, seed=71), dev)
    y = torch.empty(2, 32, 10, 10, device=dev.torch_device)
    hm.conv_fwd(x, w, None, y, (1, 1), (0, 0), 1)
    import json
    blob = json.load(open(path))
    assert blob["build"] == math_hip.HipMath._build_id()
    assert len(blob["plans"]) == 1
    (name, plan), = blob["plans"].items()
    # plant a recognisable plan: the next "process" must launch with it
    blob["plans"][name] = [64, 2048]
    json.dump(blob, open(path, "w"))
    monkeypatch.setattr(math_hip.HipMath, "_plan_file", None)
    monkeypatch.setattr(math_hip.HipMath, "_conv_plans", {})
    calls = []
    monkeypatch.setattr(math_hip.HipMath, "_measure_plan",
                        lambda self, g_, launch: calls.append(1) or (0, 0))
    y2 = torch.empty_like(y)
    hm.conv_fwd(x, w, None, y2, (1, 1), (0, 0), 1)
    assert not calls
    assert list(math_hip.HipMath._conv_plans.values()) == [(64, 2048)]
    np.testing.assert_allclose(y2.cpu().numpy(), y.cpu().numpy(), rtol=1e-5,
                               atol=1e-5)
    # a plan file written by another kernel build is ignored
    blob["build"] = "other"
    json.dump(blob, open(path, "w"))
    monkeypatch.setattr(math_hip.HipMath, "_plan_file", None)
    assert math_hip.HipMath._plan_store()[1] == {}


def test_extreme_and_nonfinite_inputs(maths):
    """The kernels are compiled with fast-math flags; the reductions that take
    extreme or non-finite values in practice are pinned here: soft-max with
    -inf mask entries and huge logits, cross-entropy with saturated logits,
    batch / layer normalisation of constant inputs (zero variance), relu and
    clamp on infinities."""
    hm, cm, dev = maths
    d = dev.torch_device
    ninf = float("-inf")
    x = torch.tensor([[0.0, ninf, 1.0, ninf], [1e30, 1.0, -1e30, 0.0],
                      [88.0, 89.0, 90.0, 91.0], [-200.0, -201.0, -199.0, -500.0]])
    out = torch.empty_like(x, device=d)
    hm.softmax(x.to(d), out)
    want = torch.softmax(x, 1)
    got = out.cpu()
    assert torch.isfinite(got).all()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-7)
    # saturated logits: the loss is the log-sum-exp gap, never inf / nan
    logits = torch.tensor([[1e4, -1e4, 0.0], [-1e4, 1e4, 0.0], [50.0, 60.0, 70.0]])
    t = torch.tensor([0, 0, 2], dtype=torch.int32)
    p = torch.empty(3, 3, device=d)
    loss = torch.empty(1, device=d)
    hm.softmax_ce_fused_fwd(logits.to(d), t.to(d), p, loss)
    # reference semantics (src/core/tensor/math_kernel.cu ComputeCrossEntropy):
    # batch mean of -log(max(p, FLT_MIN)), so a zero probability costs 87.3
    probs = torch.softmax(logits, 1)[torch.arange(3), t.long()]
    ref_v = float((-torch.log(torch.clamp(
        probs, min=float(np.finfo(np.float32).tiny)))).mean())
    got = float(loss.cpu()[0])
    assert np.isfinite(got)
    assert got == pytest.approx(ref_v, rel=1e-4)
    assert torch.isfinite(p.cpu()).all()
    # constant input: variance exactly zero
    xc = torch.full((4, 3, 5, 5), 2.5)
    C = 3
    y = torch.empty_like(xc, device=d)
    mean, istd = torch.empty(C, device=d), torch.empty(C, device=d)
    rm, rv = torch.zeros(C, device=d), torch.ones(C, device=d)
    hm.bn_fwd_train(xc.to(d), torch.ones(C, device=d), torch.zeros(C, device=d),
                    rm, rv, 0.1, 1e-5, y, mean, istd)
    assert torch.isfinite(y.cpu()).all()
    np.testing.assert_allclose(y.cpu().numpy(), 0.0, atol=1e-3)
    np.testing.assert_allclose(mean.cpu().numpy(), 2.5, rtol=1e-6)
    xr = torch.full((6, 64), -3.0)
    yr = torch.empty_like(xr, device=d)
    mu, rstd = torch.empty(6, device=d), torch.empty(6, device=d)
    hm.layernorm_fwd(xr.to(d), torch.ones(64, device=d),
                     torch.zeros(64, device=d), yr, mu, rstd, 1e-5)
    assert torch.isfinite(yr.cpu()).all()
    np.testing.assert_allclose(yr.cpu().numpy(), 0.0, atol=1e-3)
    # element-wise ops keep infinities where the maths says so
    v = torch.tensor([ninf, -1.0, 0.0, 2.0, float("inf")])
    o = torch.empty_like(v, device=d)
    hm.unary("relu", v.to(d), o)
    np.testing.assert_array_equal(o.cpu().numpy(),
                                  np.array([0, 0, 0, 2, np.inf], np.float32))
    hm.clamp(v.to(d), -1.5, 1.5, o)
    np.testing.assert_array_equal(o.cpu().numpy(),
                                  np.array([-1.5, -1, 0, 1.5, 1.5], np.float32))


@pytest.mark.parametrize("dt,N,C,K,H,W,R,st,pd,bias", [
    (torch.float32, 3, 16, 72, 28, 28, 3, 1, 1, True),    # halo kernel
    (torch.float32, 2, 24, 40, 14, 14, 3, 1, 1, False),   # halo, partial tile
    (torch.float32, 4, 32, 96, 12, 12, 1, 1, 0, True),    # implicit GEMM
    (torch.float32, 5, 20, 130, 9, 9, 3, 2, 1, False),    # strided, ragged
    (torch.float32, 2, 512, 64, 4, 4, 3, 1, 1, False),    # split reduction
    (torch.float32, 3, 12, 12, 10, 10, 3, 1, 1, False),   # depthwise (groups)
    (torch.float16, 3, 64, 96, 14, 14, 3, 1, 1, False),   # 16-bit tap GEMM
    (torch.bfloat16, 2, 64, 128, 28, 28, 1, 1, 0, False),  # 16-bit 1x1 GEMM
    (torch.float16, 2, 24, 40, 10, 10, 3, 1, 1, True),    # 16-bit implicit GEMM
])
def test_conv_forward_emits_batchnorm_statistics(maths, dt, N, C, K, H, W, R,
                                                 st, pd, bias):
    """Forward convolutions add the per-channel sums of (y - shift) and
    (y - shift)^2 to the statistics rows (fused epilogue where the kernel has
    one, a pass over y otherwise); reference: the sums of the stored output."""
    hm, cm, dev = maths
    grp = C if (C == K == 12) else 1
    x = rnd(N, C, H, W, seed=90).to(dt)
    w = rnd(K, C // grp, R, R, seed=91, lo=-0.4, hi=0.4).to(dt)
    b = (rnd(K, seed=92) + 1.5).to(dt) if bias else None
    P = (H + 2 * pd - R) // st + 1
    shift = rnd(K, seed=93, lo=-0.5, hi=0.5)
    copies = hm.STAT_COPIES
    y = torch.empty(N, K, P, P, device=dev.torch_device, dtype=dt)
    stats = torch.zeros((2 * copies + 1) * K, device=dev.torch_device)
    for _ in range(2):      # sums accumulate over launches
        hm.conv_fwd(g(x, dev), g(w, dev), g(b, dev) if bias else None, y,
                    (st, st), (pd, pd), grp, stats, g(shift, dev))
    got = stats[:2 * copies * K].cpu().reshape(copies, 2, K).sum(0).numpy()
    assert float(stats[2 * copies * K:].abs().max()) == 0.0
    d = y.float().cpu() - shift.reshape(1, K, 1, 1)
    ref1 = 2 * d.sum(dim=(0, 2, 3)).numpy()
    ref2 = 2 * (d * d).sum(dim=(0, 2, 3)).numpy()
    n = N * P * P
    np.testing.assert_allclose(got[0], ref1, rtol=1e-4,
                               atol=2e-5 * n * float(d.abs().max()))
    np.testing.assert_allclose(got[1], ref2, rtol=1e-4, atol=1e-4 * n)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("N,C,HW,act", [(6, 40, 49, 0), (4, 64, 196, 1),
                                        (3, 24, 784, 2), (33, 8, 16, 1)])
def test_batchnorm_forward_from_producer_sums(maths, dt, N, C, HW, act):
    """Training forward from statistics handed over by the producer equals
    the two-pass kernel (output, saved and running statistics), and leaves the
    statistics rows and block counters cleared."""
    hm, cm, dev = maths
    side = int(round(HW ** 0.5))
    x = (rnd(N, C, side, side, seed=95) * 2.0 + 1.0).to(dt)
    res = rnd(N, C, side, side, seed=96).to(dt) if act == 2 else None
    scale, bias = rnd(C, seed=97, lo=0.5, hi=1.5), rnd(C, seed=98)
    rm0, rv0 = rnd(C, seed=99, lo=0.5, hi=1.5), rnd(C, seed=100, lo=0.5, hi=2.0)
    copies = hm.STAT_COPIES

    def run(fused):
        rm, rv = g(rm0.clone(), dev), g(rv0.clone(), dev)
        y = torch.empty_like(x, device=dev.torch_device)
        mean = torch.empty(C, device=dev.torch_device)
        inv = torch.empty(C, device=dev.torch_device)
        kw = {}
        if fused:
            stats = torch.zeros((2 * copies + 1) * C, device=dev.torch_device)
            xd = g(x, dev)
            d = xd.float() - rm.reshape(1, C, 1, 1)
            # spread over the copies as kernels on several XCDs would
            s1, s2 = d.sum(dim=(0, 2, 3)), (d * d).sum(dim=(0, 2, 3))
            view = stats[:2 * copies * C].reshape(copies, 2, C)
            view[0, 0], view[0, 1] = 0.25 * s1, 0.25 * s2
            view[copies - 1, 0], view[copies - 1, 1] = 0.75 * s1, 0.75 * s2
            kw["stats"] = stats
        hm.bn_fwd_train(g(x, dev), g(scale, dev), g(bias, dev), rm, rv, 0.1,
                        1e-5, y, mean, inv, act,
                        g(res, dev) if res is not None else None, **kw)
        if fused:
            assert float(kw["stats"].abs().max()) == 0.0
        return [t.float().cpu().numpy() for t in (y, mean, inv, rm, rv)]

    for a, b in zip(run(True), run(False)):
        np.testing.assert_allclose(a, b, rtol=2e-3 if dt != torch.float32
                                   else 1e-4, atol=2e-3 if dt != torch.float32
                                   else 1e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,D,float_out", [
    (4099, 768, False), (2048, 100, True), (1500, 3072, False),
    (700, 33000, False)])    # the last one exceeds the scratch: float path
def test_layernorm_parameter_gradients_one_launch(maths, dt, rows, D,
                                                  float_out):
    """Row slices of the layer-norm parameter gradients meet in a scratch that
    holds zeros between launches; the last slice of a column block writes the
    totals in the outputs' type.  Reference: float64 sums on the same rounded
    inputs; run twice (a scratch left dirty would show in the second run)."""
    hm, cm, dev = maths
    d = dev.torch_device
    gen = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=gen) * 1.5 - 0.2).to(dt)
    dy = (torch.randn(rows, D, generator=gen) * 0.1).to(dt)
    gam = (torch.rand(D, generator=gen) + 0.5).to(dt)
    y = torch.empty(rows, D, dtype=dt, device=d)
    mean = torch.empty(rows, device=d)
    rstd = torch.empty(rows, device=d)
    xd, dyd, gd = x.to(d), dy.to(d), gam.to(d)
    hm.layernorm_fwd(xd, gd, torch.zeros(D, dtype=dt, device=d), y, mean,
                     rstd, 1e-5)
    odt = torch.float32 if float_out else dt
    x64, d64 = x.double(), dy.double()
    xhat = (x64 - mean.cpu().double()[:, None]) * rstd.cpu().double()[:, None]
    want_g = (d64 * xhat).sum(0).numpy()
    want_b = d64.sum(0).numpy()
    eps = {torch.float32: 1e-5, torch.float16: 1e-3, torch.bfloat16: 8e-3}[
        odt if not float_out else torch.float32]
    for _ in range(2):
        dx = torch.empty(rows, D, dtype=dt, device=d)
        dg = torch.full((D,), 7.0, dtype=odt, device=d)
        db = torch.full((D,), 7.0, dtype=odt, device=d)
        hm.layernorm_bwd(dyd, xd, gd, mean, rstd, dx, dg, db)
        for got, want in ((dg, want_g), (db, want_b)):
            np.testing.assert_allclose(
                got.double().cpu().numpy(), want, rtol=4 * eps,
                atol=4 * eps * max(1.0, float(np.abs(want).max())))
    sums, tickets = hm._zero_scratch()
    assert float(sums.abs().max()) == 0.0 and int(tickets.abs().max()) == 0


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,D", [(9, 768), (5, 100), (3, 2048), (7, 1026),
                                    (4, 516), (2, 4096), (130, 64)])
def test_layernorm_and_gelu_kernels_all_types(maths, dt, rows, D):
    """Layer normalisation (rows held by one wave, rows held by a workgroup,
    rows re-read) and GELU (vector and element kernels) in every float type
    against float64 PyTorch on the same rounded inputs."""
    hm, cm, dev = maths
    d = dev.torch_device
    gen = torch.Generator().manual_seed(rows * 131 + D)
    x = (torch.randn(rows, D, generator=gen) * 2 + 0.3).to(dt)
    gam = (torch.rand(D, generator=gen) + 0.5).to(dt)
    bet = torch.randn(D, generator=gen).to(dt)
    dy = torch.randn(rows, D, generator=gen).to(dt)
    y = torch.empty(rows, D, dtype=dt, device=d)
    mean = torch.empty(rows, device=d)
    rstd = torch.empty(rows, device=d)
    hm.layernorm_fwd(x.to(d), gam.to(d), bet.to(d), y, mean, rstd, 1e-5)
    dx = torch.empty(rows, D, dtype=dt, device=d)
    dg = torch.empty(D, dtype=dt, device=d)
    db = torch.empty(D, dtype=dt, device=d)
    hm.layernorm_bwd(dy.to(d), x.to(d), gam.to(d), mean, rstd, dx, dg, db)
    x64 = x.double().requires_grad_(True)
    g64 = gam.double().requires_grad_(True)
    b64 = bet.double().requires_grad_(True)
    y64 = torch.nn.functional.layer_norm(x64, (D,), g64, b64, 1e-5)
    (y64 * dy.double()).sum().backward()
    eps = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dt]

    def close(got, want, name, extra=1.0):
        want = want.detach().numpy()
        np.testing.assert_allclose(
            got.double().cpu().numpy(), want, rtol=eps * 4,
            atol=eps * extra * max(1.0, float(np.abs(want).max())),
            err_msg=name)
    close(y, y64, "y")
    close(mean, x.double().mean(1), "mean")
    close(dx, x64.grad, "dx", 4.0)
    close(dg, g64.grad, "dgamma", 4.0)
    close(db, b64.grad, "dbeta", 4.0)
    # GELU on the same data; an odd length takes the element kernel
    for n in (rows * D, rows * D - 3):
        xv = x.reshape(-1)[:n].contiguous()
        dv = dy.reshape(-1)[:n].contiguous()
        out = torch.empty(n, dtype=dt, device=d)
        hm.gelu_fwd(xv.to(d), out)
        xg = xv.double().requires_grad_(True)
        ref = torch.nn.functional.gelu(xg)
        close(out, ref, "gelu")
        gx = torch.empty(n, dtype=dt, device=d)
        hm.gelu_bwd(dv.to(d), xv.to(d), gx)
        (ref * dv.double()).sum().backward()
        close(gx, xg.grad, "gelu grad", 2.0)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("k,s,p,H,W", [(3, 2, 1, 28, 28), (2, 2, 0, 10, 16),
                                       (3, 1, 1, 9, 12), (3, 2, 1, 112, 112)])
def test_max_pooling_gradient_by_planes(maths, dt, k, s, p, H, W):
    """Max-pooling gradient on images whose width is a multiple of four (the
    plane kernel with four columns per thread).  Inputs come after a ReLU and
    are quantised, so equal maxima inside a window occur: every position equal
    to the window maximum receives the window's gradient (the rule of both
    GPU kernels, stated in pooling.hip; the host backend picks one position,
    so the expectation is computed here)."""
    hm, cm, dev = maths
    x = torch.relu(rnd(2, 3, H, W, seed=70) * 4).round().div(4).to(dt)
    P, Q = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ry = torch.empty(2, 3, P, Q)
    cm.pool_fwd(x.float(), ry, (k, k), (s, s), (p, p), True)
    dy = rnd(2, 3, P, Q, seed=71).to(dt)
    xf, dyf = x.float().numpy(), dy.float().numpy()
    want = np.zeros_like(xf)
    yv = ry.numpy()
    for ph in range(P):
        for qw in range(Q):
            h0, w0 = ph * s - p, qw * s - p
            hs, ws = max(h0, 0), max(w0, 0)
            he, we = min(h0 + k, H), min(w0 + k, W)
            win = xf[:, :, hs:he, ws:we]
            hit = win == yv[:, :, ph, qw][:, :, None, None]
            want[:, :, hs:he, ws:we] += hit * dyf[:, :, ph, qw][:, :, None,
                                                                 None]
    y = torch.empty(2, 3, P, Q, dtype=dt, device=dev.torch_device)
    hm.pool_fwd(g(x, dev), y, (k, k), (s, s), (p, p), True)
    np.testing.assert_array_equal(y.float().cpu().numpy(), yv)
    dx = torch.empty(2, 3, H, W, dtype=dt, device=dev.torch_device)
    hm.pool_bwd(g(dy, dev), g(x, dev), y, dx, (k, k), (s, s), (p, p), True)
    eps = {torch.float32: 1e-6, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dt]
    np.testing.assert_allclose(dx.float().cpu().numpy(), want, rtol=eps,
                               atol=eps * 8)


@pytest.mark.parametrize("M,N,K", [(64, 2048, 1595), (64, 2048, 1605),
                                   (32, 1000, 2047), (100, 70, 1537),
                                   (64, 2048, 1598)])
def test_gemm_unequal_reduction_slices(maths, M, N, K):
    """A long reduction whose length has no usable divisor (odd lengths of the
    reference's product micro-benchmark, 64 x {1595, 1605} x 2048): slices of
    a multiple of four indices, the last one shorter; row-major and transposed
    operands, scaling, previous C, bias and ReLU in the summing epilogue."""
    hm, cm, dev = maths
    assert hm._gemm_splits(M, N, K) == 1
    n_sl, kc = hm._gemm_uneven_slices(M, N, K)
    assert n_sl > 1 and kc % 4 == 0 and (n_sl - 1) * kc < K <= n_sl * kc
    a, b = rnd(M, K, seed=80), rnd(K, N, seed=81)
    bias, c0 = rnd(N, seed=82), rnd(M, N, seed=83)
    ref = a.double() @ b.double()
    c = torch.empty(M, N, device=dev.torch_device)
    hm.gemm(1.0, g(a, dev), g(b, dev), 0.0, c)
    np.testing.assert_allclose(c.cpu().numpy(), ref.numpy(), atol=2e-4)
    c = g(c0, dev).clone()
    hm.gemm(0.5, g(a.t().contiguous(), dev).t(),
            g(b.t().contiguous(), dev).t(), 2.0, c, bias=g(bias, dev), act=1)
    want = torch.relu(0.5 * ref + 2.0 * c0.double() + bias.double())
    np.testing.assert_allclose(c.cpu().numpy(), want.numpy(), atol=3e-4)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(64, 2048, 1600), (32, 1000, 2040),
                                   (128, 256, 1096)])
def test_gemm_16bit_unequal_reduction_slices(maths, dt, M, N, K):
    """16-bit products whose reduction length is a multiple of 8 but not of
    the 64-deep tiles times a slice count: slices of a multiple of 64, the
    last one shorter, summed from float partial products."""
    hm, cm, dev = maths
    n_sl, kc = hm._gemm_uneven_slices(M, N, K, 64)
    assert n_sl > 1 and kc % 64 == 0 and (n_sl - 1) * kc < K
    a, b = rnd(M, K, seed=90).to(dt), rnd(K, N, seed=91).to(dt)
    bias = rnd(N, seed=92).to(dt)
    ref = a.double() @ b.double()
    c = torch.empty(M, N, dtype=dt, device=dev.torch_device)
    hm.gemm(1.0, g(a, dev), g(b, dev), 0.0, c)
    eps = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dt]
    scale = float(ref.abs().max())
    np.testing.assert_allclose(c.double().cpu().numpy(), ref.numpy(),
                               atol=eps * scale)
    c2 = torch.empty(M, N, dtype=dt, device=dev.torch_device)
    hm.gemm(0.5, g(a.t().contiguous(), dev).t(), g(b, dev), 0.0, c2,
            bias=g(bias, dev), act=1)
    want = torch.relu(0.5 * ref + bias.double())
    np.testing.assert_allclose(c2.double().cpu().numpy(), want.numpy(),
                               atol=eps * scale)


@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("N,C,K,H,W", [(2, 16, 24, 14, 14), (3, 8, 8, 7, 7),
                                       (1, 5, 3, 9, 13), (2, 32, 16, 28, 28),
                                       (1, 3, 4, 2, 3)])
def test_winograd_float32_routes(maths, m, N, C, K, H, W):
    """F(2x2, 3x3) and F(4x4, 3x3): forward with bias and data gradient with
    an addend, image sizes that are and are not multiples of the tile, against
    the host convolution in float32."""
    hm, cm, dev = maths
    x = rnd(N, C, H, W, seed=1)
    w = rnd(K, C, 3, 3, seed=2, lo=-0.5, hi=0.5)
    b = rnd(K, seed=3)
    dy = rnd(N, K, H, W, seed=4)
    ad = rnd(N, C, H, W, seed=5)
    y_ref = torch.nn.functional.conv2d(x, w, b, stride=1, padding=1)
    dx_ref = torch.nn.functional.conv_transpose2d(dy, w, stride=1,
                                                  padding=1) + ad
    y = torch.empty(N, K, H, W, device=dev.torch_device)
    assert hm._winograd(m, 0, g(x, dev), g(w, dev), g(b, dev), None, y)
    dx = torch.empty(N, C, H, W, device=dev.torch_device)
    assert hm._winograd(m, 1, g(dy, dev), g(w, dev), None, g(ad, dev), dx)
    tol = 2e-5 if m == 2 else 1e-4
    scale = float(y_ref.abs().max())
    assert float((y.cpu() - y_ref).abs().max()) <= tol * scale
    scale = float(dx_ref.abs().max())
    assert float((dx.cpu() - dx_ref).abs().max()) <= tol * scale
    # weight gradient, float64 reference
    xd = x.double().requires_grad_(False)
    wd = w.double().requires_grad_(True)
    torch.nn.functional.conv2d(xd, wd, None, stride=1, padding=1).backward(
        dy.double())
    dw = torch.full((K, C, 3, 3), 7.0, device=dev.torch_device)
    assert hm._winograd_wgrad(m, g(dy, dev), g(x, dev), dw)
    scale = float(wd.grad.abs().max())
    assert float((dw.cpu().double() - wd.grad).abs().max()) <= tol * scale


@pytest.mark.parametrize("m", [2, 4])
def test_winograd_weight_gradient_sliced_tile_sum(maths, m):
    """Few output tiles, many image tiles: the tile sum is cut into slices
    (further batch entries of the product) that the filter transform adds."""
    hm, cm, dev = maths
    N, C, K, H = 8, 16, 16, 56
    P, T = (m + 2) ** 2, N * (-(-H // m)) ** 2
    assert hm._winograd_slices(P, K, C, T) > 1
    x, dy = rnd(N, C, H, H, seed=11), rnd(N, K, H, H, seed=12)
    wd = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double(), wd, None, stride=1,
                               padding=1).backward(dy.double())
    dw = torch.empty(K, C, 3, 3, device=dev.torch_device)
    assert hm._winograd_wgrad(m, g(dy, dev), g(x, dev), dw)
    scale = float(wd.grad.abs().max())
    assert float((dw.cpu().double() - wd.grad).abs().max()) <= \
        (2e-5 if m == 2 else 1e-4) * scale


def test_winograd_route_is_chosen_by_measurement(maths, monkeypatch):
    """A float32 3x3 layer met for the first time is timed on the direct
    kernels and both transforms; whichever wins, forward and data gradient
    agree with the host convolution, and the choice is kept as a plan."""
    import singa_amd.core.math_hip as mh
    hm, cm, dev = maths
    monkeypatch.setenv("SINGA_CONV_PLAN_CACHE", "0")
    monkeypatch.setattr(type(hm), "_plan_file", None)
    N, C, K, H = 8, 64, 64, 14
    x, w = rnd(N, C, H, H, seed=6), rnd(K, C, 3, 3, seed=7, lo=-.2, hi=.2)
    dy = rnd(N, K, H, H, seed=8)
    y = torch.empty(N, K, H, H, device=dev.torch_device)
    dx = torch.empty(N, C, H, H, device=dev.torch_device)
    dw = torch.empty(K, C, 3, 3, device=dev.torch_device)
    wd = w.double().requires_grad_(True)
    torch.nn.functional.conv2d(x.double(), wd, None, stride=1,
                               padding=1).backward(dy.double())
    for _ in range(2):      # measured, then from the plan
        hm.conv_fwd(g(x, dev), g(w, dev), None, y, (1, 1), (1, 1), 1)
        hm.conv_bwd_data(g(dy, dev), g(w, dev), dx, (1, 1), (1, 1), 1)
        hm.conv_bwd_weight(g(dy, dev), g(x, dev), dw, (1, 1), (1, 1), 1)
        assert float((dw.cpu().double() - wd.grad).abs().max()) <= \
            1e-4 * float(wd.grad.abs().max())
        y_ref = torch.nn.functional.conv2d(x, w, None, stride=1, padding=1)
        dx_ref = torch.nn.functional.conv_transpose2d(dy, w, stride=1,
                                                      padding=1)
        assert float((y.cpu() - y_ref).abs().max()) <= \
            1e-4 * float(y_ref.abs().max())
        assert float((dx.cpu() - dx_ref).abs().max()) <= \
            1e-4 * float(dx_ref.abs().max())
    keys = [k for k in type(hm)._conv_plans if k[0] == "w32"]
    assert any(k[2] == (N, C, H, H) for k in keys), keys
    monkeypatch.setattr(type(hm), "_plan_file", None)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,C,K,H,W", [(2, 64, 32, 14, 14), (3, 16, 16, 7, 7),
                                       (1, 8, 8, 9, 13)])
def test_winograd_16bit_forward_and_data_gradient(maths, dt, N, C, K, H, W):
    """F(2x2, 3x3) on 16-bit tensors (float32 transform arithmetic, operands
    of the product rounded to 16 bits) against the float32 host convolution
    of the same 16-bit inputs."""
    hm, cm, dev = maths
    x = rnd(N, C, H, W, seed=1).to(dt)
    w = rnd(K, C, 3, 3, seed=2, lo=-0.3, hi=0.3).to(dt)
    b = rnd(K, seed=3).to(dt)
    dy = rnd(N, K, H, W, seed=4).to(dt)
    ad = rnd(N, C, H, W, seed=5).to(dt)
    y_ref = torch.nn.functional.conv2d(x.float(), w.float(), b.float(),
                                       stride=1, padding=1)
    dx_ref = torch.nn.functional.conv_transpose2d(
        dy.float(), w.float(), stride=1, padding=1) + ad.float()
    y = torch.empty(N, K, H, W, dtype=dt, device=dev.torch_device)
    assert hm._winograd(2, 0, g(x, dev), g(w, dev), g(b, dev), None, y)
    dx = torch.empty(N, C, H, W, dtype=dt, device=dev.torch_device)
    assert hm._winograd(2, 1, g(dy, dev), g(w, dev), None, g(ad, dev), dx)
    tol = 4e-3 if dt == torch.float16 else 3e-2
    assert float((y.cpu().float() - y_ref).abs().max()) <= \
        tol * float(y_ref.abs().max())
    assert float((dx.cpu().float() - dx_ref).abs().max()) <= \
        tol * float(dx_ref.abs().max())
    # weight gradient: float products of the 16-bit transformed operands
    wd = w.double().requires_grad_(True)
    torch.nn.functional.conv2d(x.double(), wd, None, stride=1,
                               padding=1).backward(dy.double())
    dw = torch.empty(K, C, 3, 3, dtype=dt, device=dev.torch_device)
    if hm._winograd_wgrad(2, g(dy, dev), g(x, dev), dw):
        assert float((dw.cpu().double() - wd.grad).abs().max()) <= \
            tol * float(wd.grad.abs().max())
    else:       # tile counts the 16-bit product kernel does not take
        assert (N * (-(-H // 2)) * (-(-W // 2))) % 64 != 0
