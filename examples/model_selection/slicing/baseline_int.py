"""Latency of sliced in-database inference end to end on this machine: train
a small model on the synthetic task, save it, load it through the database
entry points and time predicate-conditioned batches (role of the reference's
model_slicing/baseline_int.py).

    python slicing/baseline_int.py [--rows 2048] [--batch 256] [--repeat 20]
"""
import argparse
import json
import os
import sys
import tempfile
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_HERE, ".."), os.path.join(_HERE, "..", "..", "..")]

from singa_amd import device  # noqa: E402
from slicing import pg_interface, sams  # noqa: E402
from trails import dataset  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--repeat", type=int, default=20)
    ap.add_argument("--epochs", type=int, default=2)
    args = ap.parse_args(argv)
    data = dataset.synthetic(n=args.rows)
    dev = device.get_default_device()
    m = sams.SlicedMoE(data.nfield, data.nfeat)
    losses = sams.train(m, dev, data, epochs=args.epochs)
    path = os.path.join(tempfile.mkdtemp(prefix="sams_"), "model.zip")
    m.save_states(path)
    cfg = {"nfield": data.nfield, "nfeat": data.nfeat, "model_path": path}
    loaded = json.loads(pg_interface.model_inference_load_model(
        json.dumps(cfg)))
    assert loaded["ok"], loaded
    ids, values, _ = next(data.batches(args.batch))
    rows = [{"id": i.tolist(), "value": v.tolist()}
            for i, v in zip(ids, values)]
    req = json.dumps({"where_cond": {"0": int(ids[0, 0])},
                      "mini_batch": rows})
    t0 = time.perf_counter()
    for _ in range(args.repeat):
        out = json.loads(pg_interface.model_inference_compute(req))
        assert out["ok"], out
    per = (time.perf_counter() - t0) / args.repeat
    result = {"train_loss_first_last": [losses[0], losses[-1]],
              "load_model_s": loaded["result"]["seconds"],
              "rows_per_request": args.batch,
              "ms_per_request": 1e3 * per,
              "experts_used": out["result"]["experts_used"]}
    print(json.dumps(result, indent=1))
    return result


if __name__ == "__main__":
    main()
