import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from singa_amd import device
from singa_amd.core.math_cpu import CpuMath
dev = device.create_cuda_gpu_on(0)
hm = dev.math; cm = device.get_default_device().math
torch.manual_seed(0)
for (N,C,K,H,W) in [(1,64,64,14,14),(1,64,64,8,8),(1,64,64,56,56)]:
    for mode in (0,1):
        x = (torch.rand(N,C,H,W)*2-1).half(); w=(torch.rand(K,C,3,3)*0.6-0.3).half(); dy=(torch.rand(N,K,H,W)*2-1).half()
        if mode==0:
            ref=torch.empty(N,K,H,W); cm.conv_fwd(x.float(),w.float(),None,ref,(1,1),(1,1),1)
            out=torch.empty(N,K,H,W,device='cuda',dtype=torch.half)
            ok=hm._conv3x3_h16(0,x.cuda(),w.cuda(),None,out)
        else:
            ref=torch.empty(N,C,H,W); cm.conv_bwd_data(dy.float(),w.float(),ref,(1,1),(1,1),1)
            out=torch.empty(N,C,H,W,device='cuda',dtype=torch.half)
            ok=hm._conv3x3_h16(1,dy.cuda(),w.cuda(),None,out)
        torch.cuda.synchronize()
        err=(out.float().cpu()-ref).abs()
        bad=(err>0.05)
        print("shape",(N,C,K,H,W),"mode",mode,"ok",ok,"max err",float(err.max()),"bad",int(bad.sum()))
        if bad.any():
            pos=bad.any(dim=1)[0]  # HxW
            print(" bad positions (h,w):", [(int(a),int(b)) for a,b in pos.nonzero()][:40])
            ch=bad.any(dim=3).any(dim=2)[0]
            print(" bad channels:", [int(c) for c in ch.nonzero().flatten()][:20], "of", ch.numel())
