"""1x1 stride-1 float32 convolutions: the implicit-GEMM convolution kernel
against one batched GEMM (a GEMM per image) on the tile family."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from singa_amd import device as sdevice
from bench_conv import timeit
sdev = sdevice.create_cuda_gpu(); dev = sdev.torch_device; m = sdev.math
N = 32
for (C, K, H) in [(64, 64, 56), (64, 256, 56), (256, 64, 56), (256, 128, 56), (128, 512, 28), (512, 128, 28), (512, 256, 28),
                  (256, 1024, 14), (1024, 256, 14), (1024, 512, 14), (512, 2048, 7), (2048, 512, 7)]:
    x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, 1, 1, device=dev) * 0.05
    y = torch.empty(N, K, H, H, device=dev); y2 = torch.empty_like(y)
    dy = torch.randn(N, K, H, H, device=dev); dx = torch.empty_like(x); dx2 = torch.empty_like(x)
    t_c = timeit(lambda: m.conv_fwd(x, w, None, y, (1, 1), (0, 0), 1), 10)
    t_d = timeit(lambda: m.conv_bwd_data(dy, w, dx, (1, 1), (0, 0), 1), 10)
    w2 = w.view(K, C)
    best = {}
    for tile in (-2, 0, 1, 2, 3, 7):
        m.k.set_gemm_f32_tile(tile)
        import singa_amd.core.math_hip as mh
        mh._GEMM_AUTOTUNE = False
        tf = timeit(lambda: m.gemm(1.0, w2.unsqueeze(0).expand(N, K, C), x.view(N, C, H * H), 0.0, y2.view(N, K, H * H)), 10)
        td = timeit(lambda: m.gemm(1.0, w2.t().unsqueeze(0).expand(N, C, K), dy.view(N, K, H * H), 0.0, dx2.view(N, C, H * H)), 10)
        best[tile] = (tf, td)
    m.k.set_gemm_f32_tile(-1)
    ef = (y - y2).abs().max().item(); ed = (dx - dx2).abs().max().item()
    bf = min(best.items(), key=lambda kv: kv[1][0]); bd = min(best.items(), key=lambda kv: kv[1][1])
    print("%4d->%4d @%3d  fwd conv %.3f  gemm %.3f (tile %d)   dgrad conv %.3f gemm %.3f (tile %d)  err %.1e %.1e" % (
        C, K, H, t_c, bf[1][0], bf[0], t_d, bd[1][1], bd[0], ef, ed), flush=True)
