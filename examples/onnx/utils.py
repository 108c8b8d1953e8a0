"""Helpers of the ONNX examples: locating model-zoo files, reading the
``test_data_set_*`` protobuf tensors that ship with zoo models, batch-size
patching.  There is no downloader: put the files under ``$SINGA_DATA``
(default /tmp) as the reference's scripts would have (examples/onnx/utils.py).
"""
import glob
import os
import sys
import tarfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..")]

from singa_amd import onnx_proto as onnx  # noqa: E402

download_dir = os.environ.get("SINGA_DATA", "/tmp")


def check_exist_or_download(url):
    """Path of the file named by ``url`` inside the data directory."""
    name = url.rsplit("/", 1)[-1]
    path = os.path.join(download_dir, name)
    if not os.path.isfile(path):
        raise FileNotFoundError(
            "%s is not present and this build never downloads; fetch %s "
            "yourself and place it there" % (path, url))
    return path


def download_model(url):
    """Unpack ``<name>.tar.gz`` (already placed in the data directory)."""
    path = check_exist_or_download(url)
    if path.endswith((".tar.gz", ".tgz")):
        with tarfile.open(path) as tar:
            tar.extractall(path=download_dir, filter="data")
    return path


def load_dataset(test_data_dir):
    """(inputs, reference outputs) stored as TensorProto files."""
    def read(pattern):
        out = []
        for f in sorted(glob.glob(os.path.join(test_data_dir, pattern))):
            t = onnx.TensorProto()
            with open(f, "rb") as fd:
                t.ParseFromString(fd.read())
            out.append(onnx.numpy_helper.to_array(t))
        return out

    return read("input_*.pb"), read("output_*.pb")


def update_batch_size(onnx_model, batch_size):
    for vi in list(onnx_model.graph.input) + list(onnx_model.graph.output):
        dims = vi.type.tensor_type.shape.dim
        if len(dims):
            dims[0].ClearField("dim_param")
            dims[0].dim_value = batch_size
    return onnx_model


def softmax(x, axis=-1):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)
