import os, sys
import numpy as np
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29548")
os.environ["SINGA_DIST_FORCE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CORR = os.environ.get("DBG_CORR", "1") == "1"
THR = int(os.environ.get("DBG_THR", "2097152"))
TOPK = os.environ.get("DBG_TOPK", "1") == "1"

def train(dist, graph, steps=5):
    from singa_amd import device, layer, model, opt, tensor
    dev = device.create_cuda_gpu_on(0)
    dev.SetRandSeed(0)
    rs = np.random.RandomState(0)
    x = rs.randn(16, 64).astype(np.float32)
    y = rs.randint(0, 10, 16).astype(np.int32)
    tx = tensor.Tensor(device=dev, data=x); ty = tensor.Tensor(device=dev, data=y)
    class MLP(model.Model):
        def __init__(self):
            super().__init__()
            self.f1 = layer.Linear(128); self.r = layer.ReLU(); self.f2 = layer.Linear(10)
            self.loss = layer.SoftMaxCrossEntropy()
        def forward(self, x):
            return self.f2(self.r(self.f1(x)))
        def train_one_batch(self, x, y):
            out = self.forward(x); l = self.loss(out, y)
            if isinstance(self.optimizer, opt.DistOpt):
                self.optimizer.backward_and_sparse_update(l, threshold=THR, spars=1.0 if TOPK else 0.0, topK=TOPK, corr=CORR)
            else:
                self.optimizer(l)
            return out, l
    sgd = opt.SGD(lr=0.05, momentum=0.9)
    if dist:
        sgd = opt.DistOpt(sgd)
    np.random.seed(7)
    m = MLP(); m.set_optimizer(sgd)
    m.compile([tx], is_train=True, use_graph=graph, sequential=dist)
    rs2 = np.random.RandomState(3)
    for k, v in sorted(m.get_params().items()):
        v.copy_from_numpy((rs2.randn(*v.shape) * 0.1).astype(np.float32))
    out = []
    for i in range(steps):
        _, loss = m(tx, ty)
        out.append(float(tensor.to_numpy(loss).reshape(-1)[0]))
    if dist:
        print("acc sums", [float(np.abs(tensor.to_numpy(a)).sum()) for a in getattr(sgd, "gradAccumulation", [])])
    norms = {k: float(np.abs(tensor.to_numpy(v)).sum()) for k, v in sorted(m.get_params().items())}
    info = dev.hipgraph_segments()
    dev.EnableGraph(False); dev.ResetGraph()
    return out, norms, info

got = train(True, True)
print("segments ", got)
ref = train(False, False)
print("eager    ", ref)
