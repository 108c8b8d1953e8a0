"""Malaria cell images (NIH / Kaggle layout): ``<root>/training_set`` and
``<root>/testing_set``, each with a ``Parasitized`` and an ``Uninfected``
folder of .png / .jpg thumbnails of varying size.

Counterpart of the reference's examples/malaria_cnn/data/malaria.py:27-122
(same folder layout, 128x128 RGB, same per-channel statistics).  Decoding and
resizing of the ~27k files runs on a thread pool (PIL releases the
interpreter lock while decoding) and the decoded uint8 array is cached next to
the dataset, so that a second run starts training within a second instead of
re-reading every file.  ``write_synthetic`` draws a small stand-in dataset of
stained-cell-like blobs for machines without the download.
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from PIL import Image, ImageDraw

CLASSES = ("Uninfected", "Parasitized")      # label 0, label 1
MEAN = np.array([0.5339, 0.4180, 0.4460], np.float32)
STD = np.array([0.3329, 0.2637, 0.2761], np.float32)


def list_images(folder):
    return sorted(f for f in os.listdir(folder)
                  if f.lower().endswith((".png", ".jpg", ".jpeg")))


def _decode(job):
    path, size = job
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB").resize(size, Image.BILINEAR),
                          dtype=np.uint8).transpose(2, 0, 1)


def load_split(split_dir, resize_size=(128, 128), workers=8, cache=True):
    """(uint8 images [n, 3, h, w], int32 labels) of one split folder."""
    tag = os.path.join(split_dir, ".decoded_%dx%d.npz" % tuple(resize_size))
    jobs, labels = [], []
    for label, name in enumerate(CLASSES):
        folder = os.path.join(split_dir, name)
        files = list_images(folder)
        jobs += [(os.path.join(folder, f), tuple(resize_size)) for f in files]
        labels += [label] * len(files)
    labels = np.asarray(labels, np.int32)
    if cache and os.path.exists(tag):
        with np.load(tag) as z:
            if z["labels"].shape == labels.shape:
                return z["images"], z["labels"]
    with ThreadPoolExecutor(max(1, workers)) as pool:
        decoded = list(pool.map(_decode, jobs, chunksize=64))
    images = np.stack(decoded) if decoded else np.zeros(
        (0, 3) + tuple(resize_size), np.uint8)
    if cache:
        try:
            np.savez(tag, images=images, labels=labels)
        except OSError:
            pass          # read-only dataset folder: decode again next time
    return images, labels


def check_dataset_exist(dirpath):
    if not os.path.isdir(os.path.join(dirpath, "training_set")):
        print("No malaria dataset under %s: expected training_set/ and "
              "testing_set/ with Parasitized/ and Uninfected/ folders" %
              dirpath)
        sys.exit(1)
    return dirpath


def load_train_data(dir_path="/tmp/malaria", resize_size=(128, 128)):
    return load_split(os.path.join(check_dataset_exist(dir_path),
                                   "training_set"), resize_size)


def load_test_data(dir_path="/tmp/malaria", resize_size=(128, 128)):
    return load_split(os.path.join(check_dataset_exist(dir_path),
                                   "testing_set"), resize_size)


def normalize(*arrays):
    """uint8 / float images in [0, 255] -> per-channel standardised float32
    (all three channels)."""
    out = []
    for x in arrays:
        x = x.astype(np.float32) / 255.0
        x -= MEAN[None, :, None, None]
        x /= STD[None, :, None, None]
        out.append(x)
    return out[0] if len(out) == 1 else tuple(out)


def load(dir_path="/tmp/malaria", resize_size=(128, 128)):
    """train_x, train_y, val_x, val_y; images float32 NCHW, standardised."""
    train_x, train_y = load_train_data(dir_path, resize_size)
    val_x, val_y = load_test_data(dir_path, resize_size)
    train_x, val_x = normalize(train_x, val_x)
    return train_x, train_y, val_x, val_y


def _draw_cell(rng, infected, side):
    """One thumbnail: a pink disc on black; infected cells carry one to three
    dark purple dots (the stained parasite)."""
    im = Image.new("RGB", (side, side), (0, 0, 0))
    d = ImageDraw.Draw(im)
    r = int(side * rng.uniform(0.38, 0.47))
    cx, cy = side // 2 + rng.randint(-3, 4), side // 2 + rng.randint(-3, 4)
    tone = (int(rng.uniform(190, 230)), int(rng.uniform(140, 180)),
            int(rng.uniform(150, 190)))
    d.ellipse((cx - r, cy - r, cx + r, cy + r), fill=tone)
    if infected:
        for _ in range(rng.randint(1, 4)):
            a, rr = rng.uniform(0, 2 * np.pi), rng.uniform(0, 0.6) * r
            px, py = cx + rr * np.cos(a), cy + rr * np.sin(a)
            pr = rng.uniform(0.08, 0.16) * r + 2
            d.ellipse((px - pr, py - pr, px + pr, py + pr),
                      fill=(int(rng.uniform(90, 130)), 30,
                            int(rng.uniform(110, 150))))
    arr = np.asarray(im, np.float32) + rng.normal(0, 6, (side, side, 3))
    return Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8))


def write_synthetic(dir_path, num_train=256, num_test=64, seed=0):
    """Write a stand-in dataset in the real folder layout (thumbnails of
    varying size, as in the real data) and return ``dir_path``."""
    rng = np.random.RandomState(seed)
    for split, count in (("training_set", num_train),
                         ("testing_set", num_test)):
        for label, name in enumerate(CLASSES):
            folder = os.path.join(dir_path, split, name)
            os.makedirs(folder, exist_ok=True)
            for i in range(count // 2):
                side = int(rng.randint(96, 160))
                _draw_cell(rng, bool(label), side).save(
                    os.path.join(folder, "cell_%04d.png" % i))
    return dir_path
