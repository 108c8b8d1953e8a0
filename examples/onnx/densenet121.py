"""DenseNet-121 (ONNX model zoo ``densenet121``): dense blocks whose layers
append ``growth`` channels to the running concatenation, halved by transition
layers.

    python densenet121.py
    python densenet121.py --model model.onnx --image cat.jpg

Workload: reference examples/onnx/densenet121.py.
"""
import sys

import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("densenet121.tar.gz", "densenet121/model.onnx")
INPUT = (3, 224, 224)
preprocess = zoo.imagenet_preprocess
postprocess = zoo.top5


def dense_layer(g, x, cin, growth):
    y = g.conv(g.relu(g.bn(x, cin)), cin, 4 * growth, 1, pad=0, bias=False)
    y = g.conv(g.relu(g.bn(y, 4 * growth)), 4 * growth, growth, 3, bias=False)
    return g.op("Concat", [x, y], axis=1)


def build(width=1.0, size=224, classes=1000, blocks=(6, 12, 24, 16), seed=0):
    g = GraphBuilder("densenet121", seed)
    growth = scaled(32, width)
    x = g.input("data_0", [1, 3, size, size])
    c = 2 * growth
    x = g.maxpool(g.conv_bn_relu(x, 3, c, 7, 2, pad=3), 3, 2, pad=1)
    for b, n in enumerate(blocks):
        for _ in range(n):
            x = dense_layer(g, x, c, growth)
            c += growth
        if b + 1 < len(blocks):
            x = g.conv(g.relu(g.bn(x, c)), c, c // 2, 1, pad=0, bias=False)
            x = g.avgpool(x, 2, 2)
            c //= 2
    x = g.global_pool(g.relu(g.bn(x, c)))
    x = g.conv(x, c, classes, 1, pad=0)           # the zoo file's 1x1 head
    g.output(x, [1, classes, 1, 1], "fc6_1")
    return g.model()


if __name__ == "__main__":
    zoo.main(["densenet121"] + sys.argv[1:])
