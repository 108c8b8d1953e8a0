"""VGG-19 (ONNX model zoo ``vgg19``): the 16-layer network of vgg16.py with
one more convolution in each of the last three stages.

    python vgg19.py
    python vgg19.py --model vgg19.onnx --image cat.jpg

Workload: reference examples/onnx/vgg19.py.
"""
import sys

import zoo
from vgg16 import vgg

ARCHIVE = ("vgg19.tar.gz", "vgg19/vgg19.onnx")
INPUT = (3, 224, 224)
preprocess = zoo.imagenet_preprocess
postprocess = zoo.top5


def build(width=1.0, size=224, classes=1000, seed=0):
    return vgg("vgg19", (2, 2, 4, 4, 4), width, size, classes, seed)


if __name__ == "__main__":
    zoo.main(["vgg19"] + sys.argv[1:])
