"""ShuffleNet v2 1.0x (ONNX model zoo ``shufflenet-v2-10``): units split the
channels in two, transform one half (1x1, depthwise 3x3, 1x1), concatenate and
shuffle; down-sampling units transform both halves of the full input.

    python shufflenetv2.py
    python shufflenetv2.py --model model.onnx --image cat.jpg

Workload: reference examples/onnx/shufflenetv2.py.
"""
import sys

import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("shufflenet-v2-10.tar.gz", "model/test_shufflenetv2/model.onnx")
INPUT = (3, 224, 224)
preprocess = zoo.imagenet_preprocess
postprocess = zoo.top5


def right_branch(g, x, cin, cout, stride):
    y = g.conv_bn_relu(x, cin, cout, 1, pad=0)
    y = g.conv_bn_relu(y, cout, cout, 3, stride, group=cout, act=False)
    return g.conv_bn_relu(y, cout, cout, 1, pad=0)


def unit(g, x, cin, cout, stride, hw):
    half = cout // 2
    if stride == 1:
        left, right = g.op("Split", [x], n_out=2, axis=1, split=[half, half])
        right = right_branch(g, right, half, half, 1)
    else:
        left = g.conv_bn_relu(x, cin, cin, 3, 2, group=cin, act=False)
        left = g.conv_bn_relu(left, cin, half, 1, pad=0)
        right = right_branch(g, x, cin, half, 2)
    return g.channel_shuffle(g.op("Concat", [left, right], axis=1),
                             cout, 2, hw)


def build(width=1.0, size=224, classes=1000, repeats=(4, 8, 4), seed=0):
    g = GraphBuilder("shufflenetv2", seed)
    x = g.input("input", [1, 3, size, size])
    c = 24
    x = g.maxpool(g.conv_bn_relu(x, 3, c, 3, 2), 3, 2, pad=1)
    hw = size // 4
    for stage, n in enumerate(repeats):
        cout = scaled(116 << stage, width, divisor=4)
        hw //= 2
        for i in range(n):
            x = unit(g, x, c, cout, 2 if i == 0 else 1, hw)
            c = cout
    last = scaled(1024, width)
    x = g.conv_bn_relu(x, c, last, 1, pad=0)
    g.output(g.dense(g.flatten(g.global_pool(x)), last, classes),
             [1, classes], "output")
    return g.model()


if __name__ == "__main__":
    zoo.main(["shufflenetv2"] + sys.argv[1:])
