"""Malaria cell-image classification (parasitized / uninfected): a folder
dataset trained with the prefetching pipeline of
../largedataset_cnn/train_largedata.py.  Point it at the extracted
``cell_images`` directory of the NIH dataset; without a path a synthetic
two-class folder is generated.  Counterpart of the reference's
examples/malaria_cnn.

    python train_cnn.py cnn /data/cell_images -m 10
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "..", "largedataset_cnn"))

import train_largedata  # noqa: E402

if __name__ == "__main__":
    train_largedata.main(sys.argv[1:])
