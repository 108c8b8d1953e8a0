"""Tiny YOLO v2 (ONNX model zoo ``tiny_yolov2``, Pascal VOC): nine
convolutions with leaky ReLU, six max poolings (the last with stride 1), a
13x13 grid of 5 anchors x (4 box + 1 objectness + 20 classes).

    python tiny_yolov2.py
    python tiny_yolov2.py --model Model.onnx --image street.jpg

Workload: reference examples/onnx/tiny_yolov2.py.
"""
import logging
import sys

import numpy as np

import utils
import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("tiny_yolov2.tar.gz", "tiny_yolov2/Model.onnx")
INPUT = (3, 416, 416)
ANCHORS = np.array([[1.08, 1.19], [3.42, 4.41], [6.63, 11.38], [9.42, 5.11],
                    [16.62, 10.52]], np.float32)
CLASSES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car",
           "cat", "chair", "cow", "diningtable", "dog", "horse", "motorbike",
           "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")
preprocess = zoo.rgb_raw_preprocess(416)


def decode(y, thresh=0.3):
    """Grid output (1, 125, S, S) -> rows [x0, y0, x1, y1, score, class] in
    pixels of the 32 S input."""
    out = np.squeeze(y)
    S = out.shape[-1]
    out = out.reshape(5, 25, S, S)
    gy, gx = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    sig = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
    cx = (sig(out[:, 0]) + gx) * 32
    cy = (sig(out[:, 1]) + gy) * 32
    w = np.exp(np.minimum(out[:, 2], 10)) * ANCHORS[:, 0, None, None] * 32
    h = np.exp(np.minimum(out[:, 3], 10)) * ANCHORS[:, 1, None, None] * 32
    conf = sig(out[:, 4])[:, None] * utils.softmax(out[:, 5:], axis=1)
    cls, score = conf.argmax(1), conf.max(1)
    keep = score > thresh
    return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, score,
                     cls], -1)[keep]


def nms(boxes, iou_thresh=0.45):
    """Greedy non-maximum suppression per class."""
    kept = []
    for b in boxes[np.argsort(-boxes[:, 4])]:
        ok = True
        for k in kept:
            if k[5] != b[5]:
                continue
            iw = min(b[2], k[2]) - max(b[0], k[0])
            ih = min(b[3], k[3]) - max(b[1], k[1])
            inter = max(iw, 0) * max(ih, 0)
            union = (b[2] - b[0]) * (b[3] - b[1]) + \
                (k[2] - k[0]) * (k[3] - k[1]) - inter
            if inter > iou_thresh * union:
                ok = False
                break
        if ok:
            kept.append(b)
    return np.array(kept).reshape(-1, 6)


def postprocess(y, labels=None):
    boxes = nms(decode(y))
    logging.info("%d boxes", len(boxes))
    for b in boxes[:5]:
        logging.info("  %s %.2f at (%.0f, %.0f)-(%.0f, %.0f)",
                     CLASSES[int(b[5])], b[4], *b[:4])
    return boxes


def build(width=1.0, size=416, seed=0):
    assert size % 32 == 0
    g = GraphBuilder("tiny_yolov2", seed)
    x = g.input("image", [1, 3, size, size])
    x = g.op("Add", [g.op("Mul", [x, g.const(np.float32(1 / 255.0), "scl")]),
                     g.const(np.zeros((3, 1, 1), np.float32), "bias")])
    c = 3
    for i in range(6):
        cout = scaled(16 << i, width)
        x = g.op("LeakyRelu", [g.conv_bn_relu(x, c, cout, 3, act=False)],
                 alpha=0.1)
        c = cout
        if i < 5:
            x = g.maxpool(x, 2, 2)
        else:   # stride 1, padded after the last row / column only
            x = g.op("MaxPool", [x], kernel_shape=[2, 2], strides=[1, 1],
                     pads=[0, 0, 1, 1])
    for _ in range(2):
        cout = scaled(1024, width)
        x = g.op("LeakyRelu", [g.conv_bn_relu(x, c, cout, 3, act=False)],
                 alpha=0.1)
        c = cout
    g.output(g.conv(x, c, 125, 1, pad=0), [1, 125, size // 32, size // 32],
             "grid")
    return g.model()


if __name__ == "__main__":
    zoo.main(["tiny_yolov2"] + sys.argv[1:])
