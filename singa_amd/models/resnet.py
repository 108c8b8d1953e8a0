"""ResNet-18/34/50/101/152 (He et al.) on the Layer / Model API.

Architecture parity: reference examples/cnn/model/resnet.py:27-295 (Bottleneck
with the stride on the 3x3 convolution, no convolution bias, BatchNorm after
every convolution, 7x7 average pool, Linear head; ``train_one_batch`` accepts
the ``dist_option`` / ``spars`` switches of the training scripts).
"""
from .. import layer
from .. import model


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
        self.bn1 = layer.BatchNorm2d(planes)
        self.relu1 = layer.ReLU()
        self.conv2 = conv3x3(planes)
        self.bn2 = layer.BatchNorm2d(planes)
        self.add = layer.Add()
        self.relu2 = layer.ReLU()
        if downsample is not None:
            self.downsample = downsample
        self.has_downsample = downsample is not None

    def forward(self, x):
        residual = x
        out = self.relu1(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.has_downsample:
            residual = self.downsample(x)
        return self.relu2(self.add(out, residual))


class Bottleneck(layer.Layer):
    expansion = 4

    def __init__(self, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = layer.Conv2d(planes, 1, bias=False)
        self.bn1 = layer.BatchNorm2d(planes)
        self.relu1 = layer.ReLU()
        self.conv2 = layer.Conv2d(planes, 3, stride=stride, padding=1,
                                  bias=False)
        self.bn2 = layer.BatchNorm2d(planes)
        self.relu2 = layer.ReLU()
        self.conv3 = layer.Conv2d(planes * self.expansion, 1, bias=False)
        self.bn3 = layer.BatchNorm2d(planes * self.expansion)
        self.add = layer.Add()
        self.relu3 = layer.ReLU()
        if downsample is not None:
            self.downsample = downsample
        self.has_downsample = downsample is not None

    def forward(self, x):
        residual = x
        out = self.relu1(self.bn1(self.conv1(x)))
        out = self.relu2(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.has_downsample:
            residual = self.downsample(x)
        return self.relu3(self.add(out, residual))


class ResNet(model.Model):

    def __init__(self, block, layers, num_classes=10, num_channels=3):
        super().__init__()
        self.inplanes = 64
        self.num_classes = num_classes
        self.input_size = 224
        self.dimension = 4
        self.conv1 = layer.Conv2d(64, 7, stride=2, padding=3, bias=False)
        self.bn1 = layer.BatchNorm2d(64)
        self.relu = layer.ReLU()
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
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
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
