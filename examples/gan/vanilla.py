"""Train a GAN on MNIST-shaped data (vanilla by default, ``--lsgan`` for the
least-squares variant; counterpart of the reference's examples/gan/vanilla.py
and lsgan.py, which differ only in model and loss).

    python vanilla.py --use_gpu --iterations 20000
"""
import argparse
import os

import numpy as np

import utils  # sets sys.path
from utils import load_data, print_log, save_grid
from model import gan_mlp, lsgan_mlp
from singa_amd import device, opt, tensor


class GANTrainer(object):

    def __init__(self, dev, rows=28, cols=28, channels=1, noise_size=100,
                 hidden_size=128, batch=128, interval=1000,
                 learning_rate=0.001, iterations=1000000,
                 dataset_filepath=None, file_dir="vanilla_images/",
                 lsgan=False):
        self.dev = dev
        self.rows, self.cols, self.channels = rows, cols, channels
        self.feature_size = rows * cols * channels
        self.noise_size = noise_size
        self.batch_size = batch // 2
        self.interval = interval
        self.learning_rate = learning_rate
        self.iterations = iterations
        self.dataset_filepath = dataset_filepath
        self.file_dir = file_dir
        maker = lsgan_mlp if lsgan else gan_mlp
        self.model = maker.create_model(noise_size=noise_size,
                                        feature_size=self.feature_size,
                                        hidden_size=hidden_size)

    def train(self):
        train_data = load_data(self.dataset_filepath)
        dev = self.dev
        dev.SetRandSeed(0)
        np.random.seed(0)
        bs = self.batch_size
        noise = tensor.Tensor((bs, self.noise_size), dev, tensor.float32)
        real_images = tensor.Tensor((bs, self.feature_size), dev,
                                    tensor.float32)
        real_labels = tensor.Tensor((bs, 1), dev, tensor.float32)
        fake_labels = tensor.Tensor((bs, 1), dev, tensor.float32)
        real_labels.set_value(1.0)
        fake_labels.set_value(0.0)
        noise.uniform(-1, 1)

        self.model.set_optimizer(opt.Adam(lr=self.learning_rate))
        self.model.compile([noise], is_train=True, use_graph=False,
                           sequential=True)
        history = []
        for it in range(self.iterations):
            idx = np.random.randint(0, train_data.shape[0], bs)
            real_images.copy_from_numpy(train_data[idx])
            self.model.train()
            _, d_real = self.model.train_one_batch_dis(real_images,
                                                       real_labels)
            noise.uniform(-1, 1)
            fake_images = self.model.forward_gen(noise)
            _, d_fake = self.model.train_one_batch_dis(fake_images,
                                                       fake_labels)
            noise.uniform(-1, 1)
            _, g_loss_t = self.model.train_one_batch(noise, real_labels)
            if it % self.interval == 0 or it == self.iterations - 1:
                d_loss = float(tensor.to_numpy(d_real).reshape(-1)[0] +
                               tensor.to_numpy(d_fake).reshape(-1)[0])
                g_loss = float(tensor.to_numpy(g_loss_t).reshape(-1)[0])
                history.append((it, g_loss, d_loss))
                self.model.eval()
                self.save_image(it)
                print_log(" The {} iteration, G_LOSS: {}, D_LOSS: {}".format(
                    it, g_loss, d_loss))
        return history

    def save_image(self, iteration, demo_row=5, demo_col=5):
        if not hasattr(self, "demo_noise"):
            self.demo_noise = tensor.Tensor(
                (demo_col * demo_row, self.noise_size), self.dev,
                tensor.float32)
        self.demo_noise.uniform(-1, 1)
        imgs = tensor.to_numpy(self.model.forward_gen(self.demo_noise))
        os.makedirs(self.file_dir, exist_ok=True)
        save_grid(imgs.reshape(-1, self.rows, self.cols), demo_row, demo_col,
                  os.path.join(self.file_dir, "%d.png" % iteration))


VANILLA = GANTrainer


def main(argv=None, lsgan=False):
    parser = argparse.ArgumentParser(description="Train GAN over MNIST")
    parser.add_argument("filepath", type=str, nargs="?", default=None,
                        help="optional .npz dataset (array x)")
    parser.add_argument("--use_gpu", action="store_true")
    parser.add_argument("--lsgan", action="store_true", default=lsgan)
    parser.add_argument("--iterations", type=int, default=1000000)
    parser.add_argument("--interval", type=int, default=1000)
    parser.add_argument("--out", type=str, default=None)
    args = parser.parse_args(argv)
    if args.use_gpu and device.get_num_gpus():
        print("Using GPU")
        dev = device.create_cuda_gpu()
    else:
        print("Using CPU")
        dev = device.get_default_device()
    out = args.out or ("lsgan_images/" if args.lsgan else "vanilla_images/")
    t = GANTrainer(dev, learning_rate=0.0005, iterations=args.iterations,
                   interval=args.interval, dataset_filepath=args.filepath,
                   file_dir=out, lsgan=args.lsgan)
    return t.train()


if __name__ == "__main__":
    main()
