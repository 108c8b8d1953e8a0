"""Xception (Chollet, arXiv:1610.02357) for 299x299 inputs, the layer plan of
reference examples/cnn/model/xceptionnet.py:25-281: entry flow (two 3x3
convolutions, three strided blocks 128 / 256 / 728), middle flow (eight
728-wide blocks of three separable convolutions), exit flow (728 -> 1024
strided block, separable 1536 and 2048), 10x10 max pooling, classifier."""
from .. import layer
from .common import Classifier


class Block(layer.Layer):
    """``reps`` x (ReLU, separable 3x3, BN) with a residual connection; a
    strided block ends in a 3x3 max pooling and projects the shortcut with a
    strided 1x1 convolution + BN."""

    def __init__(self, in_filters, out_filters, reps, strides=1, padding=0,
                 start_with_relu=True, grow_first=True):
        super(Block, self).__init__()
        if out_filters != in_filters or strides != 1:
            self.skip = layer.Conv2d(out_filters, 1, stride=strides,
                                     padding=padding, bias=False)
            self.skipbn = layer.BatchNorm2d(out_filters)
        else:
            self.skip = None
        widths = []
        if grow_first:
            widths.append(out_filters)
            widths += [out_filters] * (reps - 1)
        else:
            widths += [in_filters] * (reps - 1)
            widths.append(out_filters)
        seq = []
        for w in widths:
            seq += [layer.ReLU(),
                    layer.SeparableConv2d(w, 3, stride=1, padding=1,
                                          bias=False),
                    layer.BatchNorm2d(w)]
        if not start_with_relu:
            seq = seq[1:]
        if strides != 1:
            seq.append(layer.MaxPool2d(3, strides, padding + 1))
        self.layers = seq
        self.register_layers(*seq)
        self.add = layer.Add()

    def forward(self, x):
        y = x
        for l in self.layers:
            y = l(y)
        skip = self.skipbn(self.skip(x)) if self.skip is not None else x
        return self.add(y, skip)


class Xception(Classifier):

    def __init__(self, num_classes=10, num_channels=3):
        super(Xception, self).__init__(num_classes, 299)
        self.conv1 = layer.Conv2d(32, 3, stride=2, padding=0, bias=False)
        self.bn1 = layer.BatchNorm2d(32)
        self.relu1 = layer.ReLU()
        self.conv2 = layer.Conv2d(64, 3, stride=1, padding=1, bias=False)
        self.bn2 = layer.BatchNorm2d(64)
        self.relu2 = layer.ReLU()
        plan = [(64, 128, 2, 2, False, True), (128, 256, 2, 2, True, True),
                (256, 728, 2, 2, True, True)]
        plan += [(728, 728, 3, 1, True, True)] * 8
        plan += [(728, 1024, 2, 2, True, False)]
        self.blocks = []
        for i, (fi, fo, reps, st, relu_first, grow) in enumerate(plan, 1):
            blk = Block(fi, fo, reps, st, padding=0,
                        start_with_relu=relu_first, grow_first=grow)
            setattr(self, "block%d" % i, blk)
            self.blocks.append(blk)
        self.conv3 = layer.SeparableConv2d(1536, 3, stride=1, padding=1)
        self.bn3 = layer.BatchNorm2d(1536)
        self.relu3 = layer.ReLU()
        self.conv4 = layer.SeparableConv2d(2048, 3, stride=1, padding=1)
        self.bn4 = layer.BatchNorm2d(2048)
        self.relu4 = layer.ReLU()
        self.globalpooling = layer.MaxPool2d(10, 1)
        self.flatten = layer.Flatten()
        self.fc = layer.Linear(num_classes)

    def features(self, x):
        x = self.relu1(self.bn1(self.conv1(x)))
        x = self.relu2(self.bn2(self.conv2(x)))
        for blk in self.blocks:
            x = blk(x)
        x = self.relu3(self.bn3(self.conv3(x)))
        return self.bn4(self.conv4(x))

    def logits(self, features):
        x = self.globalpooling(self.relu4(features))
        return self.fc(self.flatten(x))

    def forward(self, x):
        return self.logits(self.features(x))


def create_model(pretrained=False, **kwargs):
    return Xception(**kwargs)


__all__ = ['Xception', 'create_model']
