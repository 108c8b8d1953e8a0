"""Smoke tests of the example programs on tiny budgets (CPU).  They check
that each script's public entry points run and that training moves the loss
in the right direction; they are not accuracy benchmarks."""
import os
import socket
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                    "examples")


def _import(subdir, name):
    import importlib
    path = os.path.join(ROOT, subdir)
    sys.path.insert(0, path)
    try:
        if name in sys.modules:
            del sys.modules[name]
        return importlib.import_module(name)
    finally:
        sys.path.remove(path)


def test_mlp_native_and_model():
    native = _import("mlp", "native")
    assert native.main(["-m", "201"]) < 0.5
    model = _import("mlp", "model")
    # three hidden units: an unlucky draw of the initial weights leaves too few
    # of them alive, so the draw is fixed
    assert model.main(["-m", "201", "--seed", "1"]) < 0.3


def test_train_cnn_helpers_and_epoch(capsys):
    tc = _import("cnn", "train_cnn")
    x = np.random.randn(6, 3, 8, 8).astype(np.float32)
    a = tc.augmentation(x.copy())
    assert a.shape == x.shape
    r = tc.resize_dataset(x, 16)
    assert r.shape == (6, 3, 16, 16)
    np.testing.assert_allclose(tc.resize_dataset(x, 8), x)
    from singa_amd import opt
    hist = tc.run(0, 1, 0, 2, 16, "cnn", "mnist", opt.SGD(0.005, 0.9, 1e-5),
                  True, 0, max_batches=12)
    assert hist[1][0] < hist[0][0]          # loss decreases
    assert hist[1][2] > 0.5                 # synthetic data is separable


def test_char_rnn_runs():
    cr = _import("rnn", "char_rnn")
    data = cr.Data(cr.__file__, batch_size=4, seq_length=12)
    m, dev, losses = cr.train(data, 2, hidden_size=12, seq_length=12,
                              batch_size=4, lr=0.5, max_batches=6)
    assert losses[1] < losses[0]
    text = cr.sample(m, data, dev, "def ", 5)
    assert len(text) == 9


def test_onnx_tiny_gpt_import():
    tm = _import("onnx", "text_models")
    from singa_amd import device
    assert tm.self_test(device.get_default_device()) < 1e-3


def _hfl():
    sys.path.insert(0, os.path.join(ROOT, "hfl"))
    try:
        for k in [k for k in sys.modules if k == "src" or
                  k.startswith("src.")]:
            sys.modules.pop(k)
        from src import bank, client, server
        from src.proto import interface, utils
    finally:
        sys.path.pop(0)
    return bank, client, server, interface, utils


def test_hfl_wire_and_fedavg():
    bank, client, server, proto, utils = _hfl()
    a, b = socket.socketpair()
    w = {"w": np.arange(6, dtype=np.float32).reshape(2, 3),
         "step": np.array([3], np.int32),
         "h": np.array([0.5, -2.0], np.float16)}
    msg = proto.WeightsExchange()
    msg.op_type = proto.GATHER
    for k, v in w.items():
        msg.weights[k] = utils.serialize_array(v)
    t = threading.Thread(target=utils.send_message, args=(a, msg))
    t.start()
    got = utils.receive_message(b, proto.WeightsExchange())
    t.join()
    assert got.op_type == proto.GATHER and set(got.weights) == set(w)
    for k, v in w.items():
        back = utils.deserialize_array(got.weights[k])
        assert back.dtype == v.dtype
        np.testing.assert_array_equal(back, v)
    # the envelope is the reference's: field 1 varint op, field 2 map entries
    wire = msg.SerializeToString()
    assert wire[:2] == b"\x08\x02" and wire[2:3] == b"\x12"
    # an object array is refused on both sides
    with pytest.raises(ValueError):
        utils.serialize_array(np.array([{"a": 1}], dtype=object))
    import io
    import pickle
    evil = io.BytesIO()
    np.lib.format.write_array(evil, np.array([{"a": 1}], dtype=object),
                              allow_pickle=True)
    with pytest.raises(ValueError):
        utils.deserialize_array(evil.getvalue())
    with pytest.raises(Exception):
        utils.deserialize_array(pickle.dumps(np.zeros(3), protocol=0))
    # a length prefix beyond the limit is refused before allocating
    t = threading.Thread(target=utils.send_int, args=(a, 1 << 40))
    t.start()
    with pytest.raises(ValueError):
        utils.receive_message(b, proto.WeightsExchange())
    t.join()
    s = server.Server(num_clients=2)
    out = s.aggregate({"w": [w["w"], 3 * w["w"]],
                       "step": [np.array([3], np.int32)] * 2})
    np.testing.assert_allclose(out["w"], 2 * w["w"])
    assert out["step"].dtype == np.int32 and out["step"][0] == 3
    s.sample_counts = [1, 3]
    np.testing.assert_allclose(
        s.aggregate({"w": [w["w"], 3 * w["w"]]})["w"], 2.5 * w["w"])
    s.sock.close()


def test_hfl_bank_pipeline(tmp_path):
    bank = _hfl()[0]
    src = bank.synthesize(str(tmp_path / bank.SOURCE), rows=600)
    import pandas as pd
    raw = pd.read_csv(src, sep=";")
    assert list(raw.columns) == bank.COLUMNS and set(raw["y"]) == {"yes", "no"}
    enc = bank.encode(raw.drop(columns=["y"]))
    # ten numeric columns + one column per category value that occurs
    assert enc.shape[1] == 10 + sum(raw[c].nunique() for c in bank.CATEGORICAL)
    per_train, per_test = bank.split(3, src, str(tmp_path))
    assert (per_train, per_test) == (160, 40)
    tx, ty, vx, vy, classes = bank.load(1, str(tmp_path))
    assert classes == 2 and tx.shape == (160, enc.shape[1]) and \
        vx.shape == (40, enc.shape[1])
    assert tx.dtype == np.float32 and ty.dtype == np.int32
    assert tx.min() == 0.0 and tx.max() == 1.0 and set(ty) <= {0, 1}
    # label-skewed shards: the first client sees (almost) only "no"
    bank.split(3, src, str(tmp_path), non_iid=True)
    y0 = bank.load(0, str(tmp_path))[1]
    y2 = bank.load(2, str(tmp_path))[1]
    assert y0.mean() < 0.05 < y2.mean()


def test_hfl_round_trip_one_client(tmp_path):
    bank, client, server, proto, utils = _hfl()
    from singa_amd import device, opt
    src = bank.synthesize(str(tmp_path / bank.SOURCE), rows=1500)
    bank.split(1, src, str(tmp_path))
    probe = socket.socket()
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    probe.close()
    srv = server.Server(1, "127.0.0.1", port)
    rounds = 3

    def serve():
        srv.start()
        for i in range(rounds):
            if i > 0:
                srv.push()
            srv.pull()
        srv.close()
    t = threading.Thread(target=serve)
    t.start()
    lines = []
    hist = client.run(0, 1, 0, rounds, 32, "mlp", "bank", "iid",
                      opt.SGD(lr=0.05, momentum=0.9), False, 0,
                      port=port, data_dir=str(tmp_path),
                      dev=device.get_default_device(), log=lines.append)
    t.join(timeout=60)
    assert not t.is_alive()
    assert len(hist) == rounds and hist[-1] > 0.6
    assert any("Training loss" in ln for ln in lines)
    # with one client the average is that client's weights
    assert set(srv.weights) and all(np.isfinite(v).all()
                                    for v in srv.weights.values())


def test_gan_step():
    sys.path.insert(0, os.path.join(ROOT, "gan"))
    try:
        for k in ("utils", "model", "vanilla"):
            sys.modules.pop(k, None)
        import vanilla
        from singa_amd import device
        t = vanilla.GANTrainer(device.get_default_device(), iterations=3,
                               interval=1, hidden_size=16, batch=16,
                               file_dir=os.path.join("/tmp", "gan_test_out"))
        hist = t.train()
        assert len(hist) == 3 and all(np.isfinite(h[1]) for h in hist)
    finally:
        sys.path.pop(0)
        for k in ("utils", "model", "vanilla"):
            sys.modules.pop(k, None)


def test_raw_autograd_mnist_with_name_clashes():
    """Layers used outside a Model share parameter names ("W", "b"); the
    optimizer must still keep one momentum buffer per parameter."""
    mod = _import(os.path.join("cnn", "autograd"), "mnist_cnn")
    acc = mod.main(["-m", "2", "-b", "32", "--max-batches", "12"])
    assert acc > 0.5


def test_raw_autograd_resnet_helpers():
    """The image helpers of the raw-autograd ResNet script: bilinear
    enlargement keeps constants and means, augmentation keeps shape."""
    mod = _import(os.path.join("cnn", "autograd"), "resnet_cifar10")
    rng = np.random.RandomState(0)
    x = rng.rand(3, 2, 8, 8).astype(np.float32)
    big = mod.enlarge(x, 32)
    assert big.shape == (3, 2, 32, 32)
    np.testing.assert_allclose(big.mean((2, 3)), x.mean((2, 3)), atol=0.03)
    np.testing.assert_allclose(mod.enlarge(np.ones_like(x), 20), 1.0,
                               rtol=1e-6)
    assert np.array_equal(mod.enlarge(x, 8), x)
    aug = mod.augment(x, rng)
    assert aug.shape == x.shape and np.isfinite(aug).all()


def test_folder_dataset_with_prefetch(tmp_path, monkeypatch):
    monkeypatch.setattr("tempfile.tempdir", str(tmp_path))
    mod = _import("largedataset_cnn", "train_largedata")
    loss, train_acc, val_acc = mod.main(["cnn", "--synthetic", "-m", "4",
                                         "-b", "16"])
    assert val_acc > 0.6


def test_malaria_folder_pipeline_and_models(tmp_path):
    """Drawn stand-in dataset -> decode / resize / cache -> standardise; the
    CNN separates the two classes within a few epochs."""
    import subprocess
    root = os.path.join(ROOT, "malaria_cnn")
    code = (
        "import sys, numpy as np; sys.argv=['x'];"
        "sys.path.insert(0, %r); import train_cnn as t;"
        "from data import malaria as m;"
        "d = m.write_synthetic(%r, 96, 32);"
        "x, y = m.load_train_data(d, (64, 64));"
        "assert x.shape == (96, 3, 64, 64) and x.dtype == np.uint8;"
        "assert sorted(np.bincount(y)) == [48, 48];"
        "x2, _ = m.load_train_data(d, (64, 64));"      # from the cache file
        "assert (x2 == x).all();"
        "n = m.normalize(x); assert abs(float(n.mean())) < 2.5;"
        "h = t.run(t.parse(['cnn', 'malaria', '-dir', d, '--image-size',"
        " '64', '-m', '6', '-b', '16', '-l', '0.01']));"
        "assert h[-1][0] < h[0][0] and h[-1][2] > 0.8, h;"
        "h = t.run(t.parse(['mlp', 'malaria', '-dir', d, '--image-size',"
        " '64', '-m', '1', '-b', '16']));"
        "print('MALARIA-OK')" % (root, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True,
                       text=True, timeout=600)
    assert "MALARIA-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_lime_finds_the_informative_region():
    mod = _import("singa_easy", "lime")
    label, w = mod.demo()
    assert int(np.argmax(w)) == label


def test_model_selection_two_phase():
    eng = _import("model_selection", "engine")
    data, val = eng.toy_data(n=384)
    out = eng.model_selection(data, val, n_candidates=6, top_k=2)
    assert out["val_acc"] > 0.8
    assert len(out["filtered"]) == 2 and len(out["rounds"]) == 2
    for name in ("grad_norm", "synflow"):
        s = eng.SCORES[name]((16,), data[0][:32], data[1][:32], data[2],
                             eng.device.get_default_device())
        assert np.isfinite(s) and s > 0


@pytest.mark.parametrize("script,args", [
    ("cifar_distributed_cnn/train_mpi.py",
     ["cnn", "mnist", "-b", "8", "--iters", "2"]),
    ("cifar_distributed_cnn/train_multiprocess.py",
     ["cnn", "mnist", "-b", "8", "--iters", "2", "-w", "2"]),
    ("cnn/train_multiprocess.py",
     ["cnn", "mnist", "-m", "1", "--max-batches", "2", "-w", "2"]),
    ("cnn/train_mpi.py", ["cnn", "mnist", "-m", "1", "--max-batches", "2"]),
])
def test_distributed_example_launchers(script, args, tmp_path):
    """The launcher scripts as a user starts them (own process, any working
    directory): the harness modules import the right files, ranks started by
    multiprocessing find them too, and the processes end cleanly."""
    import subprocess
    env = dict(os.environ, MASTER_PORT=str(29700 + os.getpid() % 200),
               CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args,
                         cwd=str(tmp_path), env=env, capture_output=True,
                         text=True, timeout=600)
    tail = out.stdout[-1500:] + out.stderr[-1500:]
    assert out.returncode == 0, tail
    assert "terminate called" not in tail and "Traceback" not in tail, tail
    assert "Throughput" in out.stdout or "accuracy" in out.stdout, tail


def test_caffe_lenet_example_trains_and_restores(tmp_path):
    import subprocess
    script = os.path.join(ROOT, "caffe", "train.py")
    saved = str(tmp_path / "lenet.caffemodel")
    out = subprocess.run(
        [sys.executable, script, "--cpu", "--iters", "40", "--batch", "16",
         "--save", saved], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    losses = [float(line.split("loss")[1].split()[0])
              for line in out.stdout.splitlines() if "loss" in line]
    assert losses[-1] < 0.5 * losses[0]
    again = subprocess.run(
        [sys.executable, script, "--cpu", "--iters", "1", "--batch", "16",
         "--weights", saved], capture_output=True, text=True, timeout=300)
    assert again.returncode == 0, again.stderr[-2000:]
    assert "restored ['conv1', 'conv2', 'ip1', 'ip2']" in again.stdout
    first = float(again.stdout.split("loss")[1].split()[0])
    assert first < 0.5 * losses[0]      # starts where the first run ended


@pytest.mark.parametrize("subdir", ["cnn", "cifar_distributed_cnn",
                                    "largedataset_cnn"])
def test_model_modules_under_the_reference_layout(subdir):
    """``from model import resnet`` inside an example directory gives the
    library's model zoo."""
    import importlib
    from singa_amd import models
    path = os.path.join(ROOT, subdir)
    sys.path.insert(0, path)
    stale = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k == "model" or k.startswith("model.")}
    try:
        for name in ("cnn", "alexnet", "resnet", "xceptionnet"):
            mod = importlib.import_module("model." + name)
            zoo = getattr(models, name)
            for public in zoo.__all__:
                assert getattr(mod, public) is getattr(zoo, public)
    finally:
        sys.path.remove(path)
        for k in list(sys.modules):
            if k == "model" or k.startswith("model."):
                del sys.modules[k]
        sys.modules.update(stale)


def test_prediction_service_batches_concurrent_requests():
    """examples/singa_easy/serve.py: concurrent requests share forward passes
    and every caller gets the rows of its own samples."""
    from starlette.testclient import TestClient
    serve = _import("singa_easy", "serve")
    from singa_amd import models, tensor

    built = []

    def build():
        built.append(models.cnn.create_model(num_channels=1, num_classes=10))
        return built[0]

    predictor = serve.BatchingPredictor(build, (1, 28, 28), max_batch=4,
                                        max_wait=0.05, graph=False)
    try:
        assert predictor.ready.wait(60) and predictor.failed is None
        rng = np.random.RandomState(0)
        samples = rng.randn(6, 1, 28, 28).astype(np.float32)
        # reference: the model applied to each sample on its own
        want = np.concatenate([predictor.predict(s) for s in samples], 0)
        before = predictor.stats["batches"]
        pending = [predictor.submit(s) for s in samples]
        got = np.concatenate([p.result(30) for p in pending], 0)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
        assert predictor.stats["batches"] - before < 6      # they were stacked
        big = predictor.predict(samples)        # more samples than one batch
        np.testing.assert_allclose(big, want, rtol=1e-5, atol=1e-6)
        with pytest.raises(ValueError):
            predictor.submit(np.zeros((3, 3, 3), np.float32))

        client = TestClient(serve.create_app(predictor, explain_samples=24))
        assert client.get("/healthz").json() == {"status": "ok"}
        r = client.post("/predict", json={"inputs": samples[:2].tolist()})
        assert r.status_code == 200
        np.testing.assert_allclose(np.array(r.json()["outputs"]), want[:2],
                                   rtol=1e-4, atol=1e-5)
        assert r.json()["classes"] == np.argmax(want[:2], 1).tolist()
        buf = __import__("io").BytesIO()
        np.save(buf, samples[2])
        r = client.post("/predict", content=buf.getvalue(),
                        headers={"content-type": "application/x-npy"})
        np.testing.assert_allclose(np.array(r.json()["outputs"]), want[2:3],
                                   rtol=1e-4, atol=1e-5)
        assert client.post("/predict", json={"inputs": [1, 2]}
                           ).status_code == 400
        r = client.post("/explain", json={"inputs": samples[0].tolist()})
        assert r.status_code == 200 and \
            len(r.json()["segment_weights"]) == 16
        text = client.get("/metrics").text
        assert "singa_serve_samples" in text and "# TYPE" in text
    finally:
        predictor.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_prediction_service_replays_a_hipgraph_from_its_worker_thread(dtype):
    serve = _import("singa_easy", "serve")
    from singa_amd import device, models, tensor
    dev = device.create_cuda_gpu()
    dt = getattr(tensor, dtype)
    holder = []

    def build():
        holder.append(models.resnet.resnet18(num_channels=3, num_classes=10))
        return holder[0]

    predictor = serve.BatchingPredictor(build, (3, 224, 224), max_batch=4,
                                        max_wait=0.02, dev=dev, dtype=dt,
                                        graph=True)
    try:
        assert predictor.ready.wait(120)
        assert predictor.failed is None, predictor.failed
        rng = np.random.RandomState(0)
        samples = rng.randn(7, 3, 224, 224).astype(np.float32)
        single = np.concatenate([predictor.predict(s) for s in samples], 0)
        pending = [predictor.submit(s) for s in samples]
        stacked = np.concatenate([p.result(60) for p in pending], 0)
        tol = 1e-4 if dtype == "float32" else 3e-2
        np.testing.assert_allclose(stacked, single, rtol=tol, atol=tol)
        np.testing.assert_allclose(predictor.predict(samples), single,
                                   rtol=tol, atol=tol)
        assert np.isfinite(single).all() and np.abs(single).max() > 0
        # one recorded pass per batch size 1, 2, 4
        assert len(holder[0]._inference.plans) == 3
    finally:
        predictor.close()
