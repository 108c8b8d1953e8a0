"""What the experiment scripts share: the synthetic tabular task (or a libsvm
file), the search space over it, a device."""
import argparse
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..", "..")]

from singa_amd import device  # noqa: E402
from trails import dataset, space  # noqa: E402


def parser(description):
    ap = argparse.ArgumentParser(description=description)
    ap.add_argument("--data", default=None,
                    help="libsvm file (default: a synthetic task)")
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--choices", default="8,16,32,64")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--out", default=None, help="write the result as JSON")
    return ap


def setup(args):
    data = dataset.load_libsvm(args.data) if args.data else \
        dataset.synthetic(n=args.rows, seed=args.seed)
    train, val = data.split(0.8, seed=args.seed)
    sp = space.MlpSpace(data.nfield, data.nfeat, num_layers=args.layers,
                        layer_choices=tuple(int(c) for c in
                                            args.choices.split(",")))
    dev = device.create_cuda_gpu() if args.gpu else \
        device.get_default_device()
    batch = next(train.batches(args.batch))
    return sp, train, val, dev, batch


def finish(args, result):
    import json
    text = json.dumps(result, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    return result
