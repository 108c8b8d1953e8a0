"""Entry points a database calls for in-database inference with a sliced
model: every function takes one JSON string and returns one JSON string, the
calling convention of the reference's model_slicing/pg_interface.py
(``model_inference_load_model``, ``model_inference_compute``,
``records_results``), and never raises across the boundary.

    load     {"model_path": "...", "nfield": F, "nfeat": N, ...}  builds the
             model, loads its states and caches it for the session
    compute  {"where_cond": {"2": 17}, "mini_batch": [{"id": [...], "value":
             [...]}, ...]}  the rows the query selected and its predicate
"""
import json
import os
import sys
import time
import traceback

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..", "..")]

from singa_amd import device, tensor  # noqa: E402
from slicing import sams  # noqa: E402

_SESSION = {"model": None, "dev": None, "config": None, "timing": {}}


def _entry(fn):
    def run(encoded):
        try:
            return json.dumps({"ok": True, "result": fn(json.loads(encoded))})
        except Exception as e:      # noqa: BLE001 (reported to the caller)
            return json.dumps({"ok": False, "error": "%s: %s" % (
                type(e).__name__, e), "trace": traceback.format_exc()})
    run.__name__ = fn.__name__
    run.__doc__ = fn.__doc__
    return run


def _model():
    if _SESSION["model"] is None:
        raise RuntimeError("no model loaded: call model_inference_load_model")
    return _SESSION["model"], _SESSION["dev"]


@_entry
def model_inference_load_model(cfg):
    """Build the model of ``cfg`` and, when ``model_path`` names a file saved
    with ``Model.save_states``, load it."""
    t0 = time.perf_counter()
    dev = device.create_cuda_gpu() if cfg.get("gpu") else \
        device.get_default_device()
    m = sams.SlicedMoE(int(cfg["nfield"]), int(cfg["nfeat"]),
                       int(cfg.get("nemb", 8)), int(cfg.get("experts", 4)),
                       int(cfg.get("hidden", 32)), int(cfg.get("top", 2)),
                       int(cfg.get("num_classes", 2)))
    F = m.nfield
    ids = tensor.from_numpy(np.zeros((1, F), np.int32), dev)
    m.compile([ids, tensor.from_numpy(np.ones((1, F), np.float32), dev), ids],
              is_train=False, use_graph=False, sequential=False)
    path = cfg.get("model_path")
    if path:
        m.load_states(path)
    m.eval()
    _SESSION.update(model=m, dev=dev, config=cfg)
    _SESSION["timing"]["load_model_s"] = time.perf_counter() - t0
    return {"experts": m.K, "top": m.top,
            "seconds": _SESSION["timing"]["load_model_s"]}


def records_to_batch(records, nfield):
    ids = np.zeros((len(records), nfield), np.int32)
    values = np.zeros((len(records), nfield), np.float32)
    for r, rec in enumerate(records):
        if len(rec["id"]) != nfield or len(rec["value"]) != nfield:
            raise ValueError("row %d does not have %d fields" % (r, nfield))
        ids[r], values[r] = rec["id"], rec["value"]
    return ids, values


@_entry
def model_inference_compute(req):
    """Class probabilities of the rows of ``mini_batch`` under the slice the
    predicate ``where_cond`` selects."""
    m, dev = _model()
    t0 = time.perf_counter()
    ids, values = records_to_batch(req["mini_batch"], m.nfield)
    if ids.size and (ids.min() < 0 or ids.max() >= m.nfeat):
        raise ValueError("feature id outside [0, %d)" % m.nfeat)
    pred = np.tile(sams.predicate_ids(m, req.get("where_cond", {})),
                   (len(ids), 1))
    t1 = time.perf_counter()
    logits = tensor.to_numpy(m.forward(
        tensor.from_numpy(ids, dev), tensor.from_numpy(values, dev),
        tensor.from_numpy(pred, dev)))
    z = logits - logits.max(axis=1, keepdims=True)
    p = np.exp(z) / np.exp(z).sum(axis=1, keepdims=True)
    gate = m.gate_weights(tensor.from_numpy(pred[:1], dev))[0]
    t2 = time.perf_counter()
    return {"probabilities": p.tolist(),
            "experts_used": [int(k) for k in np.nonzero(gate)[0]],
            "seconds": {"decode": t1 - t0, "compute": t2 - t1}}


@_entry
def records_results(_req):
    """Timings of the session so far."""
    return dict(_SESSION["timing"])
