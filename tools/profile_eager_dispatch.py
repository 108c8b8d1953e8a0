"""cProfile of the eager (operator by operator) BERT-base inference path."""
import cProfile, pstats, sys, os, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples", "onnx"))
import text_models as tm
from singa_amd import device, tensor
dev = device.create_cuda_gpu()
model, _ = tm.bert_encoder(batch=1)
rs = np.random.RandomState(0)
xs = [tensor.from_numpy(rs.randint(0, 1000, (1, 256)).astype(np.int32), dev),
      tensor.from_numpy(np.ones((1, 256), np.int32), dev),
      tensor.from_numpy(np.zeros((1, 256), np.int32), dev)]
m = tm.TextModel(model)
m.compile(xs, is_train=False, use_graph=False, sequential=True)
m.eval()
for _ in range(5):
    m(*xs)
dev.SyncNow()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    m(*xs)
dev.SyncNow()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
