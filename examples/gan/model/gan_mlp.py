"""MLP generator + discriminator pair.  ``loss='bce'`` gives the vanilla GAN
(sigmoid discriminator, binary cross entropy; reference
examples/gan/model/gan_mlp.py:25-90), ``loss='mse'`` the least-squares GAN
(linear discriminator, squared error; reference model/lsgan_mlp.py)."""
import os
import sys

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)),
                             "..", "..", "..")]

from singa_amd import autograd, layer, model  # noqa: E402


class GAN_MLP(model.Model):

    def __init__(self, noise_size=100, feature_size=784, hidden_size=128,
                 loss="bce"):
        super(GAN_MLP, self).__init__()
        self.noise_size = noise_size
        self.feature_size = feature_size
        self.hidden_size = hidden_size
        self.loss_kind = loss
        # generator: parameter names carry "gen_net", which selects them for
        # the generator update
        self.gen_net_fc_0 = layer.Linear(hidden_size)
        self.gen_net_relu_0 = layer.ReLU()
        self.gen_net_fc_1 = layer.Linear(feature_size)
        self.gen_net_sigmoid_1 = layer.Sigmoid()
        # discriminator
        self.dis_net_fc_0 = layer.Linear(hidden_size)
        self.dis_net_relu_0 = layer.ReLU()
        self.dis_net_fc_1 = layer.Linear(1)
        if loss == "bce":
            self.dis_net_sigmoid_1 = layer.Sigmoid()
            self.criterion = layer.BinaryCrossEntropy()
        else:
            self.criterion = layer.MeanSquareError()

    def forward(self, x):
        return self.forward_dis(self.forward_gen(x))

    def forward_dis(self, x):
        y = self.dis_net_fc_1(self.dis_net_relu_0(self.dis_net_fc_0(x)))
        if self.loss_kind == "bce":
            y = self.dis_net_sigmoid_1(y)
        return y

    def forward_gen(self, x):
        y = self.gen_net_relu_0(self.gen_net_fc_0(x))
        return self.gen_net_sigmoid_1(self.gen_net_fc_1(y))

    def _update(self, loss, which):
        for p, g in autograd.backward(loss):
            if which in p.name:
                self.optimizer.apply(p.name, p, g)
        self.optimizer.step()

    def train_one_batch(self, x, y):
        """Generator step: push D(G(noise)) towards ``y`` (real labels)."""
        out = self.forward(x)
        loss = self.criterion(out, y)
        self._update(loss, "gen_net")
        return out, loss

    def train_one_batch_dis(self, x, y):
        """Discriminator step on a batch of real or generated images."""
        out = self.forward_dis(x)
        loss = self.criterion(out, y)
        self._update(loss, "dis_net")
        return out, loss

    def set_optimizer(self, optimizer):
        self.optimizer = optimizer


def create_model(pretrained=False, **kwargs):
    return GAN_MLP(**kwargs)


__all__ = ["GAN_MLP", "create_model"]
