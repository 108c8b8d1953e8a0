"""Sub-pixel super-resolution (ONNX model zoo ``super_resolution``): three
feature convolutions on the luma channel, a convolution to r x r channels and
a pixel shuffle (Reshape - Transpose - Reshape) to the r-times larger image.

    python superresolution.py
    python superresolution.py --model super_resolution.onnx --image cat.jpg

Workload: reference examples/onnx/superresolution.py.
"""
import logging
import sys

import numpy as np

import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("super_resolution.onnx", "super_resolution.onnx")
INPUT = (1, 224, 224)
UPSCALE = 3


def preprocess(img):
    y = img.convert("YCbCr").resize((224, 224)).split()[0]
    return (np.asarray(y, np.float32) / 255.0)[None, None]


def postprocess(y, labels=None):
    img = np.clip(np.squeeze(y) * 255.0, 0, 255).astype(np.uint8)
    logging.info("super-resolved luma %s", img.shape)
    return img


def merge_colour(luma, img):
    """Super-resolved luma + bicubic chroma of the PIL image -> RGB image."""
    from PIL import Image
    _, cb, cr = img.convert("YCbCr").resize((224, 224)).split()
    size = luma.shape[::-1]
    return Image.merge("YCbCr", [Image.fromarray(luma, "L"),
                                 cb.resize(size, Image.BICUBIC),
                                 cr.resize(size, Image.BICUBIC)]).convert("RGB")


def build(width=1.0, size=224, upscale=UPSCALE, seed=0):
    g = GraphBuilder("super_resolution", seed)
    x = g.input("input", [1, 1, size, size])
    c1, c2 = scaled(64, width), scaled(32, width)
    x = g.relu(g.conv(x, 1, c1, 5))
    x = g.relu(g.conv(x, c1, c1, 3))
    x = g.relu(g.conv(x, c1, c2, 3))
    x = g.conv(x, c2, upscale * upscale, 3)
    x = g.reshape(x, [1, 1, upscale, upscale, size, size])
    x = g.op("Transpose", [x], perm=[0, 1, 4, 2, 5, 3])
    g.output(g.reshape(x, [1, 1, size * upscale, size * upscale]),
             [1, 1, size * upscale, size * upscale], "output")
    return g.model()


if __name__ == "__main__":
    zoo.main(["superresolution"] + sys.argv[1:])
