"""The model every client trains: a two-layer perceptron with the
``train_one_batch`` ladder of distributed options (counterpart of the
reference's examples/hfl/src/mlp.py:33-97)."""
import os
import sys

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)),
                             "..", "..", "..")]

from singa_amd import layer, model  # noqa: E402


class MLP(model.Model):

    def __init__(self, data_size=10, perceptron_size=100, num_classes=10):
        super(MLP, self).__init__()
        self.num_classes = num_classes
        self.dimension = 2          # (batch, features) input
        self.relu = layer.ReLU()
        self.linear1 = layer.Linear(perceptron_size)
        self.linear2 = layer.Linear(num_classes)
        self.softmax_cross_entropy = layer.SoftMaxCrossEntropy()

    def forward(self, inputs):
        y = self.linear1(inputs)
        y = self.relu(y)
        return self.linear2(y)

    def train_one_batch(self, x, y, dist_option="plain", spars=None):
        out = self.forward(x)
        loss = self.softmax_cross_entropy(out, y)
        if dist_option == "plain":
            self.optimizer(loss)
        elif dist_option == "half":
            self.optimizer.backward_and_update_half(loss)
        elif dist_option == "partialUpdate":
            self.optimizer.backward_and_partial_update(loss)
        elif dist_option == "sparseTopK":
            self.optimizer.backward_and_sparse_update(loss, topK=True,
                                                      spars=spars)
        elif dist_option == "sparseThreshold":
            self.optimizer.backward_and_sparse_update(loss, topK=False,
                                                      spars=spars)
        else:
            raise ValueError("unknown dist_option %r" % (dist_option,))
        return out, loss

    def set_optimizer(self, optimizer):
        self.optimizer = optimizer


def create_model(pretrained=False, **kwargs):
    return MLP(**kwargs)


__all__ = ["MLP", "create_model"]
