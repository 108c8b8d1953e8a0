"""ResNet-18/34/50/101/152 (He et al.) on the Layer / Model API.

Architecture parity: reference examples/cnn/model/resnet.py:27-295 (Bottleneck
with the stride on the 3x3 convolution, no convolution bias, BatchNorm after
every convolution, 7x7 average pool, Linear head; ``train_one_batch`` accepts
the ``dist_option`` / ``spars`` switches of the training scripts).
"""
import os

from .. import layer
from .. import model

# The ReLU after a BatchNorm -- and the residual addition in front of the last
# ReLU of a block -- run inside the BatchNorm kernels (no separate trips
# through HBM).  SINGA_RESNET_FUSE=0 builds the blocks from separate
# BatchNorm / Add / ReLU operators instead (same parameters, same numerics),
# e.g. to export a graph of primitive ONNX nodes.
FUSE_BN_ACT = os.environ.get("SINGA_RESNET_FUSE", "1") != "0"


class _BnRelu(layer.Layer):
    """relu(bn(x)) or relu(bn(x) + residual); parameters live in ``bn``."""

    def __init__(self, planes):
        super().__init__()
        self.fused = FUSE_BN_ACT
        self.bn = layer.BatchNorm2d(
            planes, activation="RELU" if self.fused else None)
        if not self.fused:
            self.add = layer.Add()
            self.relu = layer.ReLU()

    def forward(self, x, residual=None):
        if self.fused:
            return self.bn(x) if residual is None else self.bn(x, residual)
        out = self.bn(x)
        if residual is not None:
            out = self.add(out, residual)
        return self.relu(out)


def conv3x3(planes, stride=1):
    return layer.Conv2d(planes, 3, stride=stride, padding=1, bias=False)


class _Downsample(layer.Layer):
    """1x1 strided projection + BN on the shortcut branch."""

    def __init__(self, planes, stride):
        super().__init__()
        self.conv = layer.Conv2d(planes, 1, stride=stride, bias=False)
        self.bn = layer.BatchNorm2d(planes)

    def forward(self, x):
        return self.bn(self.conv(x))


class BasicBlock(layer.Layer):
    expansion = 1

    def __init__(self, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(planes, stride)
        self.bn1 = _BnRelu(planes)
        self.conv2 = conv3x3(planes)
        self.bn2 = _BnRelu(planes)
        if downsample is not None:
            self.downsample = downsample
        self.has_downsample = downsample is not None

    def forward(self, x):
        residual = x
        out = self.bn1(self.conv1(x))
        if self.has_downsample:
            residual = self.downsample(x)
        return self.bn2(self.conv2(out), residual)


class Bottleneck(layer.Layer):
    expansion = 4

    def __init__(self, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = layer.Conv2d(planes, 1, bias=False)
        self.bn1 = _BnRelu(planes)
        self.conv2 = layer.Conv2d(planes, 3, stride=stride, padding=1,
                                  bias=False)
        self.bn2 = _BnRelu(planes)
        self.conv3 = layer.Conv2d(planes * self.expansion, 1, bias=False)
        self.bn3 = _BnRelu(planes * self.expansion)
        if downsample is not None:
            self.downsample = downsample
        self.has_downsample = downsample is not None

    def forward(self, x):
        residual = x
        out = self.bn1(self.conv1(x))
        out = self.bn2(self.conv2(out))
        if self.has_downsample:
            residual = self.downsample(x)
        return self.bn3(self.conv3(out), residual)


class ResNet(model.Model):

    def __init__(self, block, layers, num_classes=10, num_channels=3):
        super().__init__()
        self.inplanes = 64
        self.num_classes = num_classes
        self.input_size = 224
        self.dimension = 4
        self.conv1 = layer.Conv2d(64, 7, stride=2, padding=3, bias=False)
        self.bn1 = _BnRelu(64)
        self.maxpool = layer.MaxPool2d(kernel_size=3, stride=2, padding=1)
        stages = []
        for i, (planes, n, stride) in enumerate(
                zip((64, 128, 256, 512), layers, (1, 2, 2, 2))):
            stages.append(self._make_layer(block, planes, n, stride))
        self.stages = stages
        blocks = [b for st in stages for b in st]
        self.register_layers(*blocks)
        self.avgpool = layer.AvgPool2d(7, stride=1)
        self.flatten = layer.Flatten()
        self.fc = layer.Linear(num_classes)
        self.softmax_cross_entropy = layer.SoftMaxCrossEntropy()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = _Downsample(planes * block.expansion, stride)
        out = [block(planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            out.append(block(planes))
        return out

    def forward(self, x):
        x = self.maxpool(self.bn1(self.conv1(x)))
        for stage in self.stages:
            for blk in stage:
                x = blk(x)
        x = self.flatten(self.avgpool(x))
        return self.fc(x)

    def train_one_batch(self, x, y, dist_option='plain', spars=None):
        out = self.forward(x)
        loss = self.softmax_cross_entropy(out, y)
        if dist_option in ('plain', 'fp32'):
            self.optimizer(loss)
        elif dist_option == 'half':
            self.optimizer.backward_and_update_half(loss)
        elif dist_option == 'partialUpdate':
            self.optimizer.backward_and_partial_update(loss)
        elif dist_option == 'sparseTopK':
            self.optimizer.backward_and_sparse_update(loss, topK=True,
                                                      spars=spars)
        elif dist_option == 'sparseThreshold':
            self.optimizer.backward_and_sparse_update(loss, topK=False,
                                                      spars=spars)
        return out, loss


def resnet18(pretrained=False, **kwargs):
    return ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet34(pretrained=False, **kwargs):
    return ResNet(BasicBlock, [3, 4, 6, 3], **kwargs)


def resnet50(pretrained=False, **kwargs):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)


def resnet101(pretrained=False, **kwargs):
    return ResNet(Bottleneck, [3, 4, 23, 3], **kwargs)


def resnet152(pretrained=False, **kwargs):
    return ResNet(Bottleneck, [3, 8, 36, 3], **kwargs)


__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101',
           'resnet152']
