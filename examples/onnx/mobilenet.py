"""MobileNet v2 1.0 (ONNX model zoo ``mobilenetv2-1.0``): inverted residual
blocks of 1x1 expansion, 3x3 depthwise and linear 1x1 projection.

    python mobilenet.py
    python mobilenet.py --model mobilenetv2-1.0.onnx --image cat.jpg

Workload: reference examples/onnx/mobilenet.py.
"""
import sys

import numpy as np

import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("mobilenetv2-1.0.tar.gz", "mobilenetv2-1.0/mobilenetv2-1.0.onnx")
INPUT = (3, 224, 224)
preprocess = zoo.imagenet_preprocess
postprocess = zoo.top5

# expansion, output channels, repeats, stride of the first repeat
SETTINGS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
            (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def relu6(g, x):
    return g.op("Clip", [x, g.const(np.float32(0), "lo"),
                         g.const(np.float32(6), "hi")])


def inverted_residual(g, x, cin, cout, stride, expand):
    mid = cin * expand
    y = x
    if expand != 1:
        y = relu6(g, g.conv_bn_relu(y, cin, mid, 1, pad=0, act=False))
    y = relu6(g, g.conv_bn_relu(y, mid, mid, 3, stride, group=mid, act=False))
    y = g.conv_bn_relu(y, mid, cout, 1, pad=0, act=False)
    return g.op("Add", [x, y]) if stride == 1 and cin == cout else y


def build(width=1.0, size=224, classes=1000, seed=0):
    g = GraphBuilder("mobilenetv2", seed)
    x = g.input("data", [1, 3, size, size])
    c = scaled(32, width)
    x = relu6(g, g.conv_bn_relu(x, 3, c, 3, 2, act=False))
    for t, ch, n, s in SETTINGS:
        cout = scaled(ch, width)
        for i in range(n):
            x = inverted_residual(g, x, c, cout, s if i == 0 else 1, t)
            c = cout
    last = scaled(1280, width)
    x = relu6(g, g.conv_bn_relu(x, c, last, 1, pad=0, act=False))
    x = g.conv(g.global_pool(x), last, classes, 1, pad=0, bias=False)
    g.output(g.reshape(x, [1, classes]), [1, classes],
             "mobilenetv20_output_flatten0_reshape0")
    return g.model()


if __name__ == "__main__":
    zoo.main(["mobilenet"] + sys.argv[1:])
