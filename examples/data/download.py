"""Fetch the datasets the example scripts read into ``$SINGA_DATA`` (default
/tmp), in the layout the loaders of this directory expect (the reference's
examples/cnn/data/download_mnist.py, download_cifar10.py, download_cifar100.py).

    python download.py mnist | cifar10 | cifar100 | all

Needs network access; without the files every script trains on synthetic data
of the same shape.
"""
import os
import sys
import tarfile
import urllib.request

ROOT = os.environ.get("SINGA_DATA", "/tmp")

MNIST_FILES = ["train-images-idx3-ubyte.gz", "train-labels-idx1-ubyte.gz",
               "t10k-images-idx3-ubyte.gz", "t10k-labels-idx1-ubyte.gz"]
MNIST_URL = "https://ossci-datasets.s3.amazonaws.com/mnist/"
CIFAR_URL = "https://www.cs.toronto.edu/~kriz/"


def fetch(url, path):
    if os.path.exists(path):
        print("have", path)
        return
    print("fetching", url)
    tmp = path + ".part"
    urllib.request.urlretrieve(url, tmp)
    os.replace(tmp, path)


def _members_below(tar, root):
    """Members whose path stays under ``root`` (archives are not trusted to
    be free of absolute or parent-relative names)."""
    root = os.path.realpath(root)
    for m in tar.getmembers():
        target = os.path.realpath(os.path.join(root, m.name))
        if (target == root or target.startswith(root + os.sep)) and \
                (m.isfile() or m.isdir()):
            yield m


def mnist():
    d = os.path.join(ROOT, "mnist")
    os.makedirs(d, exist_ok=True)
    for name in MNIST_FILES:
        fetch(MNIST_URL + name, os.path.join(d, name))


def cifar(archive):
    path = os.path.join(ROOT, archive)
    fetch(CIFAR_URL + archive, path)
    with tarfile.open(path, "r:gz") as tar:
        tar.extractall(ROOT, members=_members_below(tar, ROOT))
    print("extracted under", ROOT)


def main(argv):
    what = argv[1] if len(argv) > 1 else "all"
    if what in ("mnist", "all"):
        mnist()
    if what in ("cifar10", "all"):
        cifar("cifar-10-python.tar.gz")      # -> cifar-10-batches-py/
    if what in ("cifar100", "all"):
        cifar("cifar-100-python.tar.gz")     # -> cifar-100-python/
    if what not in ("mnist", "cifar10", "cifar100", "all"):
        sys.exit(__doc__)


if __name__ == "__main__":
    main(sys.argv)
