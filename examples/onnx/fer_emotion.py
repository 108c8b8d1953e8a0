"""FER+ emotion recognition (ONNX model zoo ``emotion_ferplus``): a VGG-13
style network on 64x64 grey faces, eight emotion classes; the input scaling
is part of the graph.

    python fer_emotion.py
    python fer_emotion.py --model model.onnx --image face.png

Workload: reference examples/onnx/fer_emotion.py.
"""
import logging
import sys

import numpy as np

import utils
import zoo
from vision_graphs import GraphBuilder, scaled

ARCHIVE = ("emotion_ferplus.tar.gz", "emotion_ferplus/model.onnx")
INPUT = (1, 64, 64)
EMOTIONS = ("neutral", "happiness", "surprise", "sadness", "anger", "disgust",
            "fear", "contempt")
preprocess = zoo.gray_preprocess(64)


def postprocess(y, labels=None):
    p = utils.softmax(np.squeeze(y))
    for i in np.argsort(p)[::-1][:3]:
        logging.info("emotion=%s ; probability=%f", EMOTIONS[i], p[i])
    return p


def build(width=1.0, size=64, classes=8, seed=0):
    g = GraphBuilder("emotion_ferplus", seed)
    x = g.input("Input3", [1, 1, size, size])
    x = g.op("Div", [g.op("Sub", [x, g.const(np.float32(127.5), "mean")]),
                     g.const(np.float32(255.0), "range")])
    c = 1
    for ch, n in ((64, 2), (128, 2), (256, 3), (256, 3)):
        cout = scaled(ch, width)
        for _ in range(n):
            x = g.relu(g.conv(x, c, cout, 3))
            c = cout
        x = g.op("Dropout", [g.maxpool(x, 2, 2)], ratio=0.25)
    feat = c * (size // 16) ** 2
    hidden = scaled(1024, width)
    x = g.reshape(x, [1, feat])
    for cin in (feat, hidden):
        x = g.op("Dropout", [g.relu(g.dense(x, cin, hidden))], ratio=0.5)
    g.output(g.dense(x, hidden, classes), [1, classes], "Plus692_Output_0")
    return g.model()


if __name__ == "__main__":
    zoo.main(["fer_emotion"] + sys.argv[1:])
