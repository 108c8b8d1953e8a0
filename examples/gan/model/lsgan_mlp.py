"""Least-squares GAN: the MLP pair of gan_mlp with the squared-error loss."""
from .gan_mlp import GAN_MLP


class LSGAN_MLP(GAN_MLP):

    def __init__(self, noise_size=100, feature_size=784, hidden_size=128):
        super(LSGAN_MLP, self).__init__(noise_size, feature_size, hidden_size,
                                        loss="mse")


def create_model(pretrained=False, **kwargs):
    return LSGAN_MLP(**kwargs)


__all__ = ["LSGAN_MLP", "create_model"]
