"""ShuffleNet v1, 3 groups (ONNX model zoo ``shufflenet``): grouped 1x1
convolutions with a channel shuffle between them; stride-2 units concatenate
an average-pooled shortcut, stride-1 units add the input.

    python shufflenetv1.py
    python shufflenetv1.py --model model.onnx --image cat.jpg

Workload: reference examples/onnx/shufflenetv1.py.
"""
import sys

import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("shufflenet.tar.gz", "shufflenet/model.onnx")
INPUT = (3, 224, 224)
preprocess = zoo.imagenet_preprocess
postprocess = zoo.top5
GROUPS = 3


def unit(g, x, cin, cout, stride, hw, first_groups):
    """``hw``: spatial size of the unit's output."""
    branch = cout - cin if stride == 2 else cout
    mid = max(GROUPS, cout // 4 // GROUPS * GROUPS)
    y = g.conv_bn_relu(x, cin, mid, 1, pad=0, group=first_groups)
    y = g.channel_shuffle(y, mid, GROUPS, hw * stride)
    y = g.conv_bn_relu(y, mid, mid, 3, stride, group=mid, act=False)
    y = g.conv_bn_relu(y, mid, branch, 1, pad=0, group=GROUPS, act=False)
    if stride == 2:
        return g.relu(g.op("Concat", [g.avgpool(x, 3, 2, pad=1), y], axis=1))
    return g.relu(g.op("Sum", [x, y]))


def build(width=1.0, size=224, classes=1000, repeats=(4, 8, 4), seed=0):
    g = GraphBuilder("shufflenet", seed)
    x = g.input("gpu_0/data_0", [1, 3, size, size])
    c = 24
    x = g.maxpool(g.conv_bn_relu(x, 3, c, 3, 2), 3, 2, pad=1)
    hw = size // 4
    for stage, n in enumerate(repeats):
        cout = scaled(240 << stage, width, divisor=4 * GROUPS)
        hw //= 2
        for i in range(n):
            x = unit(g, x, c, cout, 2 if i == 0 else 1, hw,
                     1 if stage == 0 and i == 0 else GROUPS)
            c = cout
    x = g.dense(g.flatten(g.global_pool(x)), c, classes)
    g.output(g.op("Softmax", [x], axis=1), [1, classes], "gpu_0/softmax_1")
    return g.model()


if __name__ == "__main__":
    zoo.main(["shufflenetv1"] + sys.argv[1:])
