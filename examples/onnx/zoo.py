"""Shared runner of the ONNX model-zoo vision examples.

Every network has its own script next to this file (resnet18.py, vgg16.py,
vgg19.py, mobilenet.py, densenet121.py, shufflenetv1.py, shufflenetv2.py,
squeezenet.py, arcface.py, fer_emotion.py, tiny_yolov2.py,
superresolution.py) holding its architecture as an ONNX graph builder and its
pre / post-processing.  This module imports a graph through ``sonnx``, runs it
on the device, and offers three sources for the graph:

    python resnet18.py --model /data/resnet18v1.onnx --image cat.jpg
                                      a downloaded zoo file
    python resnet18.py [--width 0.25] [--size 64]
                                      the script's own graph with random
                                      weights (no files needed); the output is
                                      checked against the node-by-node
                                      evaluator of vision_graphs.py
    python resnet18.py --self-test    export a native network of this
                                      framework to ONNX, import it back and
                                      compare outputs

Workloads: the reference's examples/onnx/<model>.py scripts.
"""
import argparse
import importlib
import logging
import os
import time

import numpy as np

import utils  # sets sys.path
from singa_amd import device, models, sonnx, tensor
from singa_amd import onnx_proto as onnx

logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")

MODELS = ("resnet18", "vgg16", "vgg19", "mobilenet", "densenet121",
          "shufflenetv1", "shufflenetv2", "squeezenet", "arcface",
          "fer_emotion", "tiny_yolov2", "superresolution")

_MEAN = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
_STD = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)


# ------------------------------------------------ shared pre / post-processing
def imagenet_preprocess(img, size=224, resize=256):
    """Resize to 256, centre crop, scale to [0, 1], normalise per channel."""
    img = img.convert("RGB").resize((resize, resize))
    off = (resize - size) // 2
    img = img.crop((off, off, off + size, off + size))
    x = np.asarray(img, np.float32).transpose(2, 0, 1) / 255.0
    return ((x - _MEAN) / _STD)[None]


def gray_preprocess(size):
    def f(img):
        x = np.asarray(img.convert("L").resize((size, size)), np.float32)
        return x[None, None]
    return f


def rgb_raw_preprocess(size):
    def f(img):
        x = np.asarray(img.convert("RGB").resize((size, size)), np.float32)
        return x.transpose(2, 0, 1)[None]
    return f


def top5(y, labels=None):
    p = utils.softmax(np.squeeze(y))
    for i in np.argsort(p)[::-1][:5]:
        logging.info("class=%s ; probability=%f",
                     labels[i] if labels else i, p[i])
    return p


# ------------------------------------------------------------------- running
class ZooModel(sonnx.SONNXModel):
    """Imported graph as a Model; first output only."""

    def forward(self, *x):
        return super(ZooModel, self).forward(*x)[0]

    def train_one_batch(self, x, y):
        raise NotImplementedError("inference example")


def create_device():
    return device.create_cuda_gpu() if device.get_num_gpus() \
        else device.get_default_device()


def run_onnx(onnx_model, x_np, dev, use_graph=True, repeat=0):
    """Output of the imported graph; with ``repeat`` also the mean time of
    that many further inferences in milliseconds."""
    model = ZooModel(onnx_model)
    x = tensor.Tensor(device=dev, data=x_np)
    model.compile([x], is_train=False, use_graph=use_graph, sequential=True)
    # with use_graph the forward pass is recorded and replayed (one hipGraph
    # launch per inference on a GPU)
    model.graph_inference = use_graph
    model.eval()
    y = tensor.to_numpy(model(x))
    if not repeat:
        return y
    for _ in range(3):
        model(x)
    dev.Sync()
    t0 = time.perf_counter()
    for _ in range(repeat):
        out = model(x)
    tensor.to_numpy(out)
    dev.Sync()
    return y, (time.perf_counter() - t0) * 1e3 / repeat


def synthetic_image(shape, seed=0):
    """A random PIL image with ``shape[0]`` channels."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    c, h, w = shape
    px = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    img = Image.fromarray(px, "RGB")
    return img.convert("L") if c == 1 else img


def run_synthetic(name, dev, width=1.0, size=None, check=True, repeat=0,
                  use_graph=True, tol=2e-3):
    """Build the script's own graph with random weights, run it through the
    importer on a random image and compare with the node-by-node evaluator.
    Returns (post-processed result, max abs error relative to the output
    scale or None)."""
    import vision_graphs
    mod = importlib.import_module(name)
    kw = {"width": width}
    if size is not None:
        kw["size"] = size
    onnx_model = mod.build(**kw)
    shape = tuple(d.dim_value for d in
                  onnx_model.graph.input[0].type.tensor_type.shape.dim)
    full = mod.preprocess(synthetic_image(mod.INPUT)).astype(np.float32)
    # a reduced-size run keeps the value range of the real preprocessing
    x_np = np.ascontiguousarray(full[:, :, :shape[2], :shape[3]])
    assert x_np.shape == shape, (x_np.shape, shape)
    n_nodes = len(onnx_model.graph.node)
    n_par = sum(int(np.prod(t.dims)) for t in onnx_model.graph.initializer)
    logging.info("%s: %d nodes, %.2f M parameters, input %s", name, n_nodes,
                 n_par / 1e6, shape)
    res = run_onnx(onnx_model, x_np, dev, use_graph=use_graph, repeat=repeat)
    if repeat:
        y, ms = res
        logging.info("%s: %.3f ms / inference (batch 1)", name, ms)
    else:
        y = res
    err = None
    if check:
        want = vision_graphs.evaluate(
            onnx_model, {onnx_model.graph.input[0].name: x_np})[0]
        err = float(np.abs(y - want).max() / max(1.0, np.abs(want).max()))
        logging.info("%s: importer vs node-by-node evaluator, max relative "
                     "difference %.3g", name, err)
        assert np.isfinite(y).all() and err < tol, (name, err)
    return mod.postprocess(y, None), err


def self_test(name, dev):
    """Export a native network, import it back, compare the outputs."""
    native = models.resnet.resnet18(num_channels=3, num_classes=10) \
        if name.startswith("resnet") else models.create(
            "cnn", num_channels=3, num_classes=10)
    size = native.input_size
    x_np = np.random.RandomState(0).randn(1, 3, size, size).astype(np.float32)
    x = tensor.Tensor(device=dev, data=x_np)
    native.compile([x], is_train=False, use_graph=False, sequential=True)
    native.eval()
    y = native.forward(x)
    onnx_model = sonnx.to_onnx([x], [y])
    y2 = run_onnx(onnx_model, x_np, dev, use_graph=False)
    err = float(np.abs(tensor.to_numpy(y) - y2).max())
    logging.info("round-trip max abs difference: %g", err)
    assert err < 1e-3
    return err


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("name", choices=MODELS)
    ap.add_argument("--model", default=None, help=".onnx file")
    ap.add_argument("--image", default=None)
    ap.add_argument("--image2", default=None,
                    help="second face (arcface: similarity of the two)")
    ap.add_argument("--labels", default=None, help="synset.txt")
    ap.add_argument("--self-test", action="store_true")
    ap.add_argument("--width", type=float, default=1.0,
                    help="channel multiplier of the synthetic graph")
    ap.add_argument("--size", type=int, default=None,
                    help="input size of the synthetic graph")
    ap.add_argument("--repeat", type=int, default=0,
                    help="time this many inferences")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args(argv)
    dev = create_device()
    if args.self_test:
        return self_test(args.name, dev)
    mod = importlib.import_module(args.name)
    path = args.model
    if path is None:
        zoo_file = os.path.join(utils.download_dir, mod.ARCHIVE[1])
        if os.path.isfile(zoo_file):
            path = zoo_file
    if path is None:
        logging.info("no model file: building %s with random weights",
                     args.name)
        return run_synthetic(args.name, dev, args.width, args.size,
                             check=not args.no_check, repeat=args.repeat)[0]
    logging.info("onnx load model...")
    onnx_model = onnx.load(path)
    logging.info("preprocessing...")
    if args.image:
        from PIL import Image
        x_np = mod.preprocess(Image.open(args.image))
    else:
        data_dir = os.path.join(os.path.dirname(path), "test_data_set_0")
        if os.path.isdir(data_dir):
            x_np = utils.load_dataset(data_dir)[0][0]
        else:
            x_np = mod.preprocess(synthetic_image(mod.INPUT))
    labels = None
    if args.labels:
        labels = [l.rstrip() for l in open(args.labels)]
    logging.info("model running...")
    y = run_onnx(onnx_model, x_np.astype(np.float32), dev)
    logging.info("postprocessing...")
    out = mod.postprocess(y, labels)
    if args.image2 and hasattr(mod, "similarity"):
        from PIL import Image
        y2 = run_onnx(onnx_model, mod.preprocess(Image.open(args.image2))
                      .astype(np.float32), dev)
        sim, dist = mod.similarity(out, mod.postprocess(y2, labels))
        logging.info("similarity %.4f, squared distance %.4f", sim, dist)
    return out


if __name__ == "__main__":
    main()
