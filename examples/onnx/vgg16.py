"""VGG-16 (ONNX model zoo ``vgg16``): ImageNet classification; ``vgg()`` is
shared with vgg19.py.

    python vgg16.py                          # synthetic weights, random image
    python vgg16.py --model vgg16.onnx --image cat.jpg

Workload: reference examples/onnx/vgg16.py.
"""
import sys

import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("vgg16.tar.gz", "vgg16/vgg16.onnx")
INPUT = (3, 224, 224)
preprocess = zoo.imagenet_preprocess
postprocess = zoo.top5


def vgg(name, convs_per_stage, width, size, classes, seed):
    """Five stages of 3x3 convolutions + ReLU closed by 2x2 max pooling, then
    two hidden dense layers with dropout and the classifier."""
    assert size % 32 == 0, "five poolings need a multiple of 32"
    g = GraphBuilder(name, seed)
    x = g.input("data", [1, 3, size, size])
    c = 3
    for stage, n in enumerate(convs_per_stage):
        cout = scaled(min(64 << stage, 512), width)
        for _ in range(n):
            x = g.relu(g.conv(x, c, cout, 3))
            c = cout
        x = g.maxpool(x, 2, 2)
    feat = c * (size // 32) ** 2
    hidden = scaled(4096, width)
    x = g.flatten(x)
    for cin in (feat, hidden):
        x = g.op("Dropout", [g.relu(g.dense(x, cin, hidden))], ratio=0.5)
    g.output(g.dense(x, hidden, classes), [1, classes], name + "_dense2_fwd")
    return g.model()


def build(width=1.0, size=224, classes=1000, seed=0):
    return vgg("vgg16", (2, 2, 3, 3, 3), width, size, classes, seed)


if __name__ == "__main__":
    zoo.main(["vgg16"] + sys.argv[1:])
