"""ArcFace LResNet100E-IR (ONNX model zoo ``resnet100``): face embedding.
Improved-residual units (BN - conv - BN - PReLU - strided conv - BN), input
normalisation inside the graph, 512-d embedding head; two faces are compared
by the angle between their embeddings.

    python arcface.py
    python arcface.py --model resnet100.onnx --image face1.jpg --image2 face2.jpg

Workload: reference examples/onnx/arcface.py.
"""
import logging
import sys

import numpy as np

import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("resnet100.tar.gz", "resnet100/resnet100.onnx")
INPUT = (3, 112, 112)
preprocess = zoo.rgb_raw_preprocess(112)


def postprocess(y, labels=None):
    v = np.squeeze(y).astype(np.float64)
    v = v / max(np.linalg.norm(v), 1e-12)
    logging.info("embedding dim=%d, first values %s", v.size, v[:4])
    return v


def similarity(e1, e2):
    """(cosine similarity, squared distance) of two normalised embeddings."""
    return float(np.dot(e1, e2)), float(np.sum((e1 - e2) ** 2))


def prelu(g, x, c):
    return g.op("PRelu", [x, g.const(np.full((c, 1, 1), 0.25, np.float32),
                                     "slope")])


def ir_unit(g, x, cin, cout, stride):
    y = g.conv(g.bn(x, cin, eps=2e-5), cin, cout, 3, bias=False)
    y = prelu(g, g.bn(y, cout, eps=2e-5), cout)
    y = g.bn(g.conv(y, cout, cout, 3, stride, bias=False), cout, eps=2e-5)
    if stride != 1 or cin != cout:
        x = g.bn(g.conv(x, cin, cout, 1, stride, pad=0, bias=False), cout,
                 eps=2e-5)
    return g.op("Add", [y, x])


def build(width=1.0, size=112, embedding=512, units=(3, 13, 30, 3), seed=0):
    g = GraphBuilder("arcface_resnet100", seed)
    x = g.input("data", [1, 3, size, size])
    x = g.op("Mul", [g.op("Sub", [x, g.const(np.float32(127.5), "mean")]),
                     g.const(np.float32(0.0078125), "scale")])
    c = scaled(64, width)
    x = prelu(g, g.bn(g.conv(x, 3, c, 3, bias=False), c, eps=2e-5), c)
    for stage, n in enumerate(units):
        cout = scaled(64 << stage, width)
        for i in range(n):
            x = ir_unit(g, x, c, cout, 2 if i == 0 else 1)
            c = cout
    x = g.op("Dropout", [g.bn(x, c, eps=2e-5)], ratio=0.4)
    feat = c * (size // 16) ** 2
    x = g.dense(g.flatten(x), feat, embedding)
    # the embedding's batch norm runs on a (N, C, 1, 1) view
    x = g.bn(g.reshape(x, [1, embedding, 1, 1]), embedding, eps=2e-5)
    g.output(g.flatten(x), [1, embedding], "fc1")
    return g.model()


if __name__ == "__main__":
    zoo.main(["arcface"] + sys.argv[1:])
