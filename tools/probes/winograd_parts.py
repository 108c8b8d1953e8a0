"""Time of each piece of the F4 route per ResNet-50 3x3 layer (fp32, batch 32)
next to the time its bytes would take at 5.5 TB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from singa_amd import device
dev = device.create_cuda_gpu_on(0)
hm = dev.math; k = hm.k
td = dev.torch_device

def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(8): fn()
    g.replay()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 32 * 1e3   # us

m = 4; P = 36
for (C, HW) in ((64, 56), (128, 28), (256, 14), (512, 7)):
    N = 32; K = C
    T = N * (-(-HW // m)) ** 2
    x = torch.randn(N, C, HW, HW, device=td); w = torch.randn(K, C, 3, 3, device=td)
    U = torch.empty(P, K, C, device=td); V = torch.empty(P, C, T, device=td)
    M = torch.empty(P, K, T, device=td); y = torch.empty(N, K, HW, HW, device=td)
    s = lambda: hm._s()
    t_prep = timed(lambda: k.wino_prepare(0, m, x.data_ptr(), V.data_ptr(), N, C, HW, HW, w.data_ptr(), U.data_ptr(), K, C, False, s()))
    t_in = timed(lambda: k.wino_input(0, m, x.data_ptr(), V.data_ptr(), N, C, HW, HW, 1, s()))
    t_f = timed(lambda: k.wino_filter(0, m, w.data_ptr(), U.data_ptr(), K, C, False, s()))
    t_g = timed(lambda: hm.gemm(1.0, U, V, 0.0, M))
    t_out = timed(lambda: k.wino_output(0, m, M.data_ptr(), 0, 0, y.data_ptr(), N, K, HW, HW, s()))
    mb = lambda *ts: sum(t.numel() * 4 for t in ts) / 1e6
    print("C=%d HW=%d: prepare %.1f us (input alone %.1f, filter alone %.1f; bytes %.0f MB = %.1f us) | gemm %.1f us (%.0f MB = %.1f us) | output %.1f us (%.0f MB = %.1f us)" % (
        C, HW, t_prep, t_in, t_f, mb(x, V), mb(x, V) / 5.5, t_g, mb(U, V, M), mb(U, V, M) / 5.5, t_out, mb(M, y), mb(M, y) / 5.5), flush=True)
