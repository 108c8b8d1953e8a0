"""ResNet-18 v1 (ONNX model zoo ``resnet18v1``): ImageNet classification.

    python resnet18.py                       # synthetic weights, random image
    python resnet18.py --model resnet18v1.onnx --image cat.jpg --labels synset.txt

Workload: reference examples/onnx/resnet18.py.
"""
import sys

import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("resnet18v1.tar.gz", "resnet18v1/resnet18v1.onnx")
INPUT = (3, 224, 224)
preprocess = zoo.imagenet_preprocess
postprocess = zoo.top5


def basic_block(g, x, cin, cout, stride):
    y = g.conv_bn_relu(x, cin, cout, 3, stride)
    y = g.conv_bn_relu(y, cout, cout, 3, act=False)
    if stride != 1 or cin != cout:
        x = g.conv_bn_relu(x, cin, cout, 1, stride, pad=0, act=False)
    return g.relu(g.op("Add", [y, x]))


def build(width=1.0, size=224, classes=1000, blocks=(2, 2, 2, 2), seed=0):
    g = GraphBuilder("resnet18v1", seed)
    x = g.input("data", [1, 3, size, size])
    c = scaled(64, width)
    x = g.conv_bn_relu(x, 3, c, 7, 2, pad=3)
    x = g.maxpool(x, 3, 2, pad=1)
    for stage, n in enumerate(blocks):
        cout = scaled(64 << stage, width)
        for i in range(n):
            x = basic_block(g, x, c, cout, 2 if stage and i == 0 else 1)
            c = cout
    x = g.dense(g.flatten(g.global_pool(x)), c, classes)
    g.output(x, [1, classes], "resnetv15_dense0_fwd")
    return g.model()


if __name__ == "__main__":
    zoo.main(["resnet18"] + sys.argv[1:])
