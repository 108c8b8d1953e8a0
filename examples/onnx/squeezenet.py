"""SqueezeNet 1.1 (ONNX model zoo ``squeezenet1.1``): fire modules squeeze to
few channels with a 1x1 convolution and expand with parallel 1x1 and 3x3
convolutions whose outputs are concatenated.

    python squeezenet.py
    python squeezenet.py --model squeezenet1.1.onnx --image cat.jpg

Workload: reference examples/onnx/squeezenet.py.
"""
import sys

import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("squeezenet1.1.tar.gz", "squeezenet1.1/squeezenet1.1.onnx")
INPUT = (3, 224, 224)
preprocess = zoo.imagenet_preprocess
postprocess = zoo.top5


def fire(g, x, cin, squeeze, expand):
    s = g.relu(g.conv(x, cin, squeeze, 1, pad=0))
    return g.op("Concat", [g.relu(g.conv(s, squeeze, expand, 1, pad=0)),
                           g.relu(g.conv(s, squeeze, expand, 3))], axis=1)


def build(width=1.0, size=224, classes=1000, seed=0):
    g = GraphBuilder("squeezenet1.1", seed)
    x = g.input("data", [1, 3, size, size])
    c = scaled(64, width)
    x = g.maxpool(g.relu(g.conv(x, 3, c, 3, 2, pad=0)), 3, 2)
    for i, (sq, ex) in enumerate(((16, 64), (16, 64), (32, 128), (32, 128),
                                  (48, 192), (48, 192), (64, 256), (64, 256))):
        sq, ex = scaled(sq, width), scaled(ex, width)
        x = fire(g, x, c, sq, ex)
        c = 2 * ex
        if i in (1, 3):
            x = g.maxpool(x, 3, 2)
    x = g.op("Dropout", [x], ratio=0.5)
    x = g.global_pool(g.relu(g.conv(x, c, classes, 1, pad=0)))
    g.output(g.flatten(x), [1, classes], "squeezenet0_flatten0_reshape0")
    return g.model()


if __name__ == "__main__":
    zoo.main(["squeezenet"] + sys.argv[1:])
