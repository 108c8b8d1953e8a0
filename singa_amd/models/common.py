"""Shared training step of the image classifiers: soft-max cross-entropy and
the optimizer call selected by ``dist_option`` (reference examples/cnn/model/
*.py repeat this block in every model)."""
from .. import layer, model


def apply_update(optimizer, loss, dist_option='plain', spars=None):
    """Back-propagate ``loss`` and update through ``optimizer`` using one of
    DistOpt's gradient exchange schemes."""
    if dist_option in ('plain', 'fp32'):
        optimizer(loss)
    elif dist_option == 'half':
        optimizer.backward_and_update_half(loss)
    elif dist_option == 'partialUpdate':
        optimizer.backward_and_partial_update(loss)
    elif dist_option == 'sparseTopK':
        optimizer.backward_and_sparse_update(loss, topK=True, spars=spars)
    elif dist_option == 'sparseThreshold':
        optimizer.backward_and_sparse_update(loss, topK=False, spars=spars)
    else:
        raise ValueError("unknown dist_option %r" % (dist_option,))


class Classifier(model.Model):
    """Base of the zoo's classifiers; subclasses define ``forward``."""

    def __init__(self, num_classes, input_size, dimension=4):
        super(Classifier, self).__init__()
        self.num_classes = num_classes
        self.input_size = input_size
        self.dimension = dimension
        self.softmax_cross_entropy = layer.SoftMaxCrossEntropy()

    def train_one_batch(self, x, y, dist_option='plain', spars=None):
        out = self.forward(x)
        loss = self.softmax_cross_entropy(out, y)
        apply_update(self.optimizer, loss, dist_option, spars)
        return out, loss

    def set_optimizer(self, optimizer):
        self.optimizer = optimizer
