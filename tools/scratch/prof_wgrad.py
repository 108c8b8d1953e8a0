"""Weight-gradient launches of a few ResNet-50 layers, for a kernel trace."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                "..", ".."))
from singa_amd import device  # noqa: E402

dev = device.create_cuda_gpu_on(0)
m = dev.math
dt = torch.float16
layers = [(64, 64, 1, 56), (256, 64, 1, 56), (64, 256, 1, 56), (64, 64, 3, 56),
          (512, 128, 1, 28), (128, 128, 3, 28), (1024, 256, 1, 14),
          (256, 256, 3, 14), (256, 1024, 1, 14)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for C, K, R, H in layers:
    x = torch.randn(32, C, H, H, device="cuda", dtype=dt)
    dy = torch.randn(32, K, H, H, device="cuda", dtype=dt)
    dw = torch.empty(K, C, R, R, device="cuda", dtype=dt)
    for _ in range(reps):
        m.conv_bwd_weight(dy, x, dw, (1, 1), (R // 2, R // 2), 1)
    torch.cuda.synchronize()
print("done")
