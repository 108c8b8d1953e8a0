"""Index an image-folder dataset (``root/<class name>/*.{png,jpg,...}``) into
the ``path label`` list files that ``singa_amd.data.ImageBatchIter`` reads.
``make_synthetic`` writes a small folder dataset of coloured shapes so the
training script can be tried without any download.  Counterpart of the
reference's examples/largedataset_cnn/process_data.py."""
import os
import random

import numpy as np

EXT = (".png", ".jpg", ".jpeg", ".bmp", ".ppm")


def index_folder(root, out_prefix, val_ratio=0.2, seed=0):
    classes = sorted(d for d in os.listdir(root)
                     if os.path.isdir(os.path.join(root, d)))
    items = []
    for label, c in enumerate(classes):
        for fn in sorted(os.listdir(os.path.join(root, c))):
            if fn.lower().endswith(EXT):
                items.append((os.path.join(c, fn), label))
    random.Random(seed).shuffle(items)
    nval = int(len(items) * val_ratio)
    splits = {"val": items[:nval], "train": items[nval:]}
    for name, rows in splits.items():
        with open("%s_%s.txt" % (out_prefix, name), "w") as f:
            for path, label in rows:
                f.write("%s %d\n" % (path, label))
    return classes, {k: len(v) for k, v in splits.items()}


def make_synthetic(root, classes=4, per_class=40, size=48, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    colours = rng.randint(40, 215, (classes, 3))
    for c in range(classes):
        d = os.path.join(root, "class%d" % c)
        os.makedirs(d, exist_ok=True)
        for i in range(per_class):
            img = rng.randint(0, 60, (size, size, 3)).astype(np.int32)
            x0, y0 = rng.randint(0, size // 2, 2)
            img[y0:y0 + size // 2, x0:x0 + size // 2] += colours[c]
            Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB").save(
                os.path.join(d, "%03d.png" % i))
    return root


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--out", default="data")
    a = ap.parse_args()
    print(index_folder(a.root, a.out))
