"""LeNet-style CNN for 28x28 inputs (reference examples/cnn/model/cnn.py:24-72):
conv5-20 + ReLU, pool, conv5-50 + ReLU, pool, fc500 + ReLU, fc."""
from .. import layer
from .common import Classifier


class CNN(Classifier):

    def __init__(self, num_classes=10, num_channels=1):
        super(CNN, self).__init__(num_classes, 28)
        self.conv1 = layer.Conv2d(20, 5, padding=0, activation="RELU")
        self.conv2 = layer.Conv2d(50, 5, padding=0, activation="RELU")
        self.pooling1 = layer.MaxPool2d(2, 2, padding=0)
        self.pooling2 = layer.MaxPool2d(2, 2, padding=0)
        self.flatten = layer.Flatten()
        self.linear1 = layer.Linear(500)
        self.relu = layer.ReLU()
        self.linear2 = layer.Linear(num_classes)

    def forward(self, x):
        y = self.pooling1(self.conv1(x))
        y = self.pooling2(self.conv2(y))
        y = self.relu(self.linear1(self.flatten(y)))
        return self.linear2(y)


def create_model(pretrained=False, **kwargs):
    return CNN(**kwargs)


__all__ = ['CNN', 'create_model']
