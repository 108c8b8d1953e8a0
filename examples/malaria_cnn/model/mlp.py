"""Two-layer perceptron baseline on flattened thumbnails (reference
examples/malaria_cnn/model/mlp.py:27-60): fc(perceptron_size) + ReLU,
fc(num_classes)."""
from singa_amd import layer
from singa_amd.models.common import Classifier


class MLP(Classifier):

    def __init__(self, data_size=3 * 128 * 128, perceptron_size=100,
                 num_classes=2):
        super(MLP, self).__init__(num_classes, data_size, dimension=2)
        self.linear1 = layer.Linear(perceptron_size)
        self.relu = layer.ReLU()
        self.linear2 = layer.Linear(num_classes)

    def forward(self, x):
        return self.linear2(self.relu(self.linear1(x)))


def create_model(pretrained=False, **kwargs):
    return MLP(**kwargs)


__all__ = ["MLP", "create_model"]
