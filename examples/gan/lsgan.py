"""Least-squares GAN on MNIST-shaped data; see vanilla.py."""
import vanilla

LSGAN = vanilla.GANTrainer

if __name__ == "__main__":
    vanilla.main(lsgan=True)
