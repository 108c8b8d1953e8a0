"""Two-layer perceptron (reference examples/mlp/model.py:33-73)."""
from .. import layer
from .common import Classifier


class MLP(Classifier):

    def __init__(self, data_size=10, perceptron_size=100, num_classes=10):
        super(MLP, self).__init__(num_classes, data_size, dimension=2)
        self.linear1 = layer.Linear(perceptron_size)
        self.relu = layer.ReLU()
        self.linear2 = layer.Linear(num_classes)

    def forward(self, inputs):
        return self.linear2(self.relu(self.linear1(inputs)))


def create_model(pretrained=False, **kwargs):
    return MLP(**kwargs)


__all__ = ['MLP', 'create_model']
