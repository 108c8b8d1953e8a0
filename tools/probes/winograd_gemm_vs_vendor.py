"""The batched float32 products of the Winograd routes: own GEMM against the
vendor library (torch.bmm), ResNet-50 3x3 layer shapes at batch 32."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from singa_amd import device
dev = device.create_cuda_gpu_on(0)
hm = dev.math
td = dev.torch_device
torch.backends.cuda.matmul.allow_tf32 = False

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
    return e0.elapsed_time(e1) / 32

for m in (4, 2):
    P = (m + 2) ** 2
    for (C, HW) in ((64, 56), (128, 28), (256, 14), (512, 7)):
        T = 32 * (-(-HW // m)) ** 2
        U = torch.randn(P, C, C, device=td); V = torch.randn(P, C, T, device=td)
        M = torch.empty(P, C, T, device=td)
        t_own = timed(lambda: hm.gemm(1.0, U, V, 0.0, M))
        t_lib = timed(lambda: torch.bmm(U, V, out=M))
        gf = 2.0 * P * C * C * T / 1e9
        print("F%d C=%d HW=%d: [%dx%d]x[%dx%d] x%d  own %.3f ms (%.0f TF/s)  vendor %.3f ms (%.0f TF/s)" % (
            m, C, HW, C, C, C, T, P, t_own, gf / t_own, t_lib, gf / t_lib), flush=True)
