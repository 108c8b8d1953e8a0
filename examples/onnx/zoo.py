"""Run ONNX model-zoo vision networks through the importer.

    python zoo.py resnet18 --model /data/resnet18v1.onnx --image cat.jpg
    python zoo.py resnet18 --self-test        # no files needed

One registry replaces the reference's per-model scripts (examples/onnx/
{resnet18,vgg16,vgg19,mobilenet,densenet121,shufflenetv1,shufflenetv2,
squeezenet,arcface,fer_emotion,tiny_yolov2,superresolution}.py), which differ
only in pre/post-processing.  ``--self-test`` exports a network of this
framework's own zoo to ONNX, imports it back and compares outputs, so the
whole path is exercised on a machine without the downloaded files.
"""
import argparse
import logging
import os

import numpy as np

import utils  # sets sys.path
from singa_amd import device, models, sonnx, tensor
from singa_amd import onnx_proto as onnx

logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")

_MEAN = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
_STD = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)


def _imagenet(img, size=224, resize=256):
    img = img.convert("RGB").resize((resize, resize))
    off = (resize - size) // 2
    img = img.crop((off, off, off + size, off + size))
    x = np.asarray(img, np.float32).transpose(2, 0, 1) / 255.0
    return ((x - _MEAN) / _STD)[None]


def _gray(size):
    def f(img):
        x = np.asarray(img.convert("L").resize((size, size)), np.float32)
        return x[None, None]
    return f


def _rgb_raw(size):
    def f(img):
        x = np.asarray(img.convert("RGB").resize((size, size)), np.float32)
        return x.transpose(2, 0, 1)[None]
    return f


def _sr_pre(img):
    y = img.convert("YCbCr").resize((224, 224)).split()[0]
    return (np.asarray(y, np.float32) / 255.0)[None, None]


def _top5(y, labels=None):
    p = utils.softmax(np.squeeze(y))
    for i in np.argsort(p)[::-1][:5]:
        logging.info("class=%s ; probability=%f",
                     labels[i] if labels else i, p[i])
    return p


def _embedding(y, labels=None):
    v = np.squeeze(y)
    v = v / max(np.linalg.norm(v), 1e-12)
    logging.info("embedding dim=%d, first values %s", v.size, v[:4])
    return v


def _yolo(y, labels=None, thresh=0.3):
    """Tiny-YOLOv2 head (1,125,13,13): 5 anchors x (4 box + 1 obj + 20 cls)."""
    anchors = np.array([[1.08, 1.19], [3.42, 4.41], [6.63, 11.38],
                        [9.42, 5.11], [16.62, 10.52]], np.float32)
    out = np.squeeze(y).reshape(5, 25, 13, 13)
    gy, gx = np.meshgrid(np.arange(13), np.arange(13), indexing="ij")
    sig = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
    cx = (sig(out[:, 0]) + gx) * 32
    cy = (sig(out[:, 1]) + gy) * 32
    w = np.exp(out[:, 2]) * anchors[:, 0, None, None] * 32
    h = np.exp(out[:, 3]) * anchors[:, 1, None, None] * 32
    conf = sig(out[:, 4])[:, None] * utils.softmax(out[:, 5:], axis=1)
    cls = conf.argmax(1)
    score = conf.max(1)
    keep = score > thresh
    boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2,
                      score, cls], -1)[keep]
    logging.info("%d boxes above %.2f", len(boxes), thresh)
    return boxes


def _sr_post(y, labels=None):
    img = np.clip(np.squeeze(y) * 255.0, 0, 255).astype(np.uint8)
    logging.info("super-resolved luma %s", img.shape)
    return img


# name -> (zoo archive, onnx path inside it, preprocess, postprocess)
REGISTRY = {
    "resnet18": ("resnet18v1.tar.gz", "resnet18v1/resnet18v1.onnx",
                 _imagenet, _top5),
    "vgg16": ("vgg16.tar.gz", "vgg16/vgg16.onnx", _imagenet, _top5),
    "vgg19": ("vgg19.tar.gz", "vgg19/vgg19.onnx", _imagenet, _top5),
    "mobilenet": ("mobilenetv2-1.0.tar.gz",
                  "mobilenetv2-1.0/mobilenetv2-1.0.onnx", _imagenet, _top5),
    "densenet121": ("densenet121.tar.gz", "densenet121/model.onnx",
                    _imagenet, _top5),
    "shufflenetv1": ("shufflenet.tar.gz", "shufflenet/model.onnx",
                     _imagenet, _top5),
    "shufflenetv2": ("shufflenet-v2-10.tar.gz", "model/test_shufflenetv2/"
                     "model.onnx", _imagenet, _top5),
    "squeezenet": ("squeezenet1.1.tar.gz", "squeezenet1.1/squeezenet1.1.onnx",
                   _imagenet, _top5),
    "arcface": ("resnet100.tar.gz", "resnet100/resnet100.onnx",
                _rgb_raw(112), _embedding),
    "fer_emotion": ("emotion_ferplus.tar.gz", "emotion_ferplus/model.onnx",
                    _gray(64), _top5),
    "tiny_yolov2": ("tiny_yolov2.tar.gz", "tiny_yolov2/Model.onnx",
                    _rgb_raw(416), _yolo),
    "superresolution": ("super_resolution.onnx", "super_resolution.onnx",
                        _sr_pre, _sr_post),
}


class ZooModel(sonnx.SONNXModel):
    """Imported graph as a Model; first output only."""

    def forward(self, *x):
        return super(ZooModel, self).forward(*x)[0]

    def train_one_batch(self, x, y):
        pass


def create_device():
    return device.create_cuda_gpu() if device.get_num_gpus() \
        else device.get_default_device()


def run_onnx(onnx_model, x_np, dev, use_graph=True):
    model = ZooModel(onnx_model)
    x = tensor.Tensor(device=dev, data=x_np)
    model.compile([x], is_train=False, use_graph=use_graph, sequential=True)
    return tensor.to_numpy(model.forward(x))


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
    ap.add_argument("name", choices=sorted(REGISTRY))
    ap.add_argument("--model", default=None, help=".onnx file")
    ap.add_argument("--image", default=None)
    ap.add_argument("--labels", default=None, help="synset.txt")
    ap.add_argument("--self-test", action="store_true")
    args = ap.parse_args(argv)
    dev = create_device()
    if args.self_test:
        return self_test(args.name, dev)
    archive, inner, pre, post = REGISTRY[args.name]
    path = args.model
    if path is None:
        path = os.path.join(utils.download_dir, inner)
        if not os.path.isfile(path):
            utils.download_model(archive)
    logging.info("onnx load model...")
    onnx_model = onnx.load(path)
    logging.info("preprocessing...")
    if args.image:
        from PIL import Image
        x_np = pre(Image.open(args.image))
    else:
        inputs, _ = utils.load_dataset(os.path.join(os.path.dirname(path),
                                                    "test_data_set_0"))
        x_np = inputs[0]
    labels = None
    if args.labels:
        labels = [l.rstrip() for l in open(args.labels)]
    logging.info("model running...")
    y = run_onnx(onnx_model, x_np.astype(np.float32), dev)
    logging.info("postprocessing...")
    return post(y, labels)


if __name__ == "__main__":
    main()
