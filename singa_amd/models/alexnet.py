"""AlexNet for 224x224 inputs (reference examples/cnn/model/alexnet.py:24-101):
five convolutions (11/4, 5, 3, 3, 3) with 2x2 max pooling after the 1st, 2nd
and 5th, a 3x3/2 average pooling, then two dropout + fc4096 blocks and the
classifier."""
from .. import layer
from .common import Classifier

# (filters, kernel, stride, padding, max-pool after)
_CONVS = [(64, 11, 4, 2, True), (192, 5, 1, 2, True), (384, 3, 1, 1, False),
          (256, 3, 1, 1, False), (256, 3, 1, 1, True)]


class AlexNet(Classifier):

    def __init__(self, num_classes=10, num_channels=1):
        super(AlexNet, self).__init__(num_classes, 224)
        self.features = []
        for i, (f, k, s, p, pool) in enumerate(_CONVS, 1):
            conv = layer.Conv2d(f, k, stride=s, padding=p)
            relu = layer.ReLU()
            setattr(self, "conv%d" % i, conv)
            setattr(self, "relu%d" % i, relu)
            self.features += [conv, relu]
            if pool:
                mp = layer.MaxPool2d(2, 2, padding=0)
                setattr(self, "pooling%d" % (len([c for c in _CONVS[:i]
                                                   if c[4]])), mp)
                self.features.append(mp)
        self.avg_pooling1 = layer.AvgPool2d(3, 2, padding=0)
        self.flatten = layer.Flatten()
        self.dropout1 = layer.Dropout()
        self.linear1 = layer.Linear(4096)
        self.relu6 = layer.ReLU()
        self.dropout2 = layer.Dropout()
        self.linear2 = layer.Linear(4096)
        self.relu7 = layer.ReLU()
        self.linear3 = layer.Linear(num_classes)

    def forward(self, x):
        y = x
        for f in self.features:
            y = f(y)
        y = self.flatten(self.avg_pooling1(y))
        y = self.relu6(self.linear1(self.dropout1(y)))
        y = self.relu7(self.linear2(self.dropout2(y)))
        return self.linear3(y)


def create_model(pretrained=False, **kwargs):
    return AlexNet(**kwargs)


__all__ = ['AlexNet', 'create_model']
