"""The vision networks of the ONNX model-zoo examples (examples/onnx/
<model>.py): each script's own graph is built at a reduced width and input
size, imported through sonnx, run, and compared with the node-by-node
evaluator of examples/onnx/vision_graphs.py (numpy + float64 PyTorch CPU
convolutions, independent of the importer).  Post-processing helpers are
checked on hand-made inputs.  The reference runs the downloaded zoo files
(examples/onnx/*.py); those files are not available here."""
import importlib
import os
import sys

import numpy as np
import pytest

from singa_amd import device

ZOO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                   "examples", "onnx")


@pytest.fixture(scope="module")
def zoo():
    sys.path.insert(0, ZOO)
    try:
        for name in ("zoo", "vision_graphs", "utils"):
            sys.modules.pop(name, None)
        yield importlib.import_module("zoo")
    finally:
        sys.path.remove(ZOO)
        for name in list(sys.modules):
            f = getattr(sys.modules[name], "__file__", None) or ""
            if os.path.dirname(os.path.abspath(f)) == os.path.abspath(ZOO):
                del sys.modules[name]


# name, width, input size, expected output shape of the graph
CASES = [
    ("resnet18", 0.25, 64, (1, 1000)),
    ("vgg16", 0.125, 32, (1, 1000)),
    ("vgg19", 0.125, 32, (1, 1000)),
    ("mobilenet", 0.25, 64, (1, 1000)),
    ("densenet121", 0.25, 64, (1, 1000, 1, 1)),
    ("shufflenetv1", 0.25, 64, (1, 1000)),
    ("shufflenetv2", 0.25, 64, (1, 1000)),
    ("squeezenet", 0.25, 64, (1, 1000)),
    ("arcface", 0.125, 32, (1, 512)),
    ("fer_emotion", 0.125, 32, (1, 8)),
    ("tiny_yolov2", 0.125, 96, (1, 125, 3, 3)),
    ("superresolution", 0.25, 32, (1, 1, 96, 96)),
]


def _run(zoo, dev, name, width, size, shape):
    mod = importlib.import_module(name)
    model = mod.build(width=width, size=size)
    out_dims = tuple(d.dim_value for d in
                     model.graph.output[0].type.tensor_type.shape.dim)
    assert out_dims == shape
    _, err = zoo.run_synthetic(name, dev, width, size, check=True,
                               use_graph=False)
    assert err < 2e-3


@pytest.mark.parametrize("name,width,size,shape", CASES,
                         ids=[c[0] for c in CASES])
def test_zoo_network_host(zoo, name, width, size, shape):
    _run(zoo, device.get_default_device(), name, width, size, shape)


@pytest.mark.gpu
@pytest.mark.parametrize("name,width,size,shape", CASES,
                         ids=[c[0] for c in CASES])
def test_zoo_network_gpu(zoo, name, width, size, shape):
    _run(zoo, device.create_cuda_gpu(), name, width, size, shape)


def test_zoo_architectures_at_full_size(zoo):
    """Node and parameter counts of the full-size graphs (no execution)."""
    expect = {  # name -> (convolutions, parameters in millions, +-3 %)
        "resnet18": (20, 11.7), "vgg16": (13, 138.4), "vgg19": (16, 143.7),
        "mobilenet": (53, 3.5), "densenet121": (121, 8.0),
        "squeezenet": (26, 1.24), "shufflenetv2": (56, 2.3),
    }
    for name, (convs, mparams) in expect.items():
        g = importlib.import_module(name).build().graph
        n_conv = sum(n.op_type == "Conv" for n in g.node)
        n_par = sum(int(np.prod(t.dims)) for t in g.initializer
                    if len(t.dims) > 1) / 1e6
        assert n_conv == convs, (name, n_conv)
        assert abs(n_par - mparams) / mparams < 0.03, (name, n_par)


def test_yolo_decode_and_nms(zoo):
    ty = importlib.import_module("tiny_yolov2")
    y = np.full((1, 125, 13, 13), -10.0, np.float32)
    # anchor 1 at cell (row 4, col 6): centred box, confident, class 11 (dog)
    a, r, c = 1, 4, 6
    y[0, a * 25 + 0, r, c] = 0.0       # sigmoid -> 0.5 => x = (6.5) * 32
    y[0, a * 25 + 1, r, c] = 0.0
    y[0, a * 25 + 2, r, c] = 0.0       # exp(0) * anchor
    y[0, a * 25 + 3, r, c] = 0.0
    y[0, a * 25 + 4, r, c] = 8.0
    y[0, a * 25 + 5 + 11, r, c] = 12.0
    # a weaker overlapping duplicate one cell to the right, same class
    y[0, a * 25:(a + 1) * 25, r, c + 1] = y[0, a * 25:(a + 1) * 25, r, c]
    y[0, a * 25 + 4, r, c + 1] = 2.0
    boxes = ty.decode(y)
    assert len(boxes) == 2
    kept = ty.nms(boxes)
    assert len(kept) == 1
    b = kept[0]
    np.testing.assert_allclose([(b[0] + b[2]) / 2, (b[1] + b[3]) / 2],
                               [6.5 * 32, 4.5 * 32], rtol=1e-5)
    np.testing.assert_allclose([b[2] - b[0], b[3] - b[1]],
                               [3.42 * 32, 4.41 * 32], rtol=1e-5)
    assert int(b[5]) == 11 and b[4] > 0.99


def test_arcface_similarity_and_preprocessing(zoo):
    af = importlib.import_module("arcface")
    e1 = af.postprocess(np.array([[3.0, 4.0, 0.0]]))
    e2 = af.postprocess(np.array([[4.0, 3.0, 0.0]]))
    np.testing.assert_allclose(np.linalg.norm(e1), 1.0)
    sim, dist = af.similarity(e1, e2)
    np.testing.assert_allclose([sim, dist], [0.96, 0.08], atol=1e-9)
    img = zoo.synthetic_image((3, 50, 70))
    x = zoo.imagenet_preprocess(img)
    assert x.shape == (1, 3, 224, 224) and abs(float(x.mean())) < 1.0
    assert zoo.gray_preprocess(64)(img).shape == (1, 1, 64, 64)
    sr = importlib.import_module("superresolution")
    luma = sr.postprocess(np.full((1, 1, 6, 6), 0.5, np.float32))
    assert luma.dtype == np.uint8 and luma[0, 0] == 127
    assert sr.merge_colour(np.zeros((672, 672), np.uint8), img).size == \
        (672, 672)


# ---- text examples: tokenizers and answer extraction --------------------------
def _text_module(sub, name):
    path = os.path.join(ZOO, sub) if sub else ZOO
    sys.path.insert(0, path)
    try:
        sys.modules.pop(name, None)
        return importlib.import_module(name)
    finally:
        sys.path.remove(path)


def test_wordpiece_windows_and_spans(zoo):
    st = _text_module("bert", "squad_text")
    vocab = {t: i for i, t in enumerate(
        ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "where", "is", "the", "center",
         "?", "mos", "##cone", "in", "san", "francisco", ".", "un", "##aff",
         "##able"])}
    assert st.basic_tokens("Où est-il?") == ["ou", "est", "-", "il", "?"]
    assert st.wordpiece("moscone", vocab) == ["mos", "##cone"]
    assert st.wordpiece("unaffable", vocab) == ["un", "##aff", "##able"]
    assert st.wordpiece("xyz", vocab) == ["[UNK]"]
    wins, words = st.windows("Where is the center?",
                             "The Moscone Center is in San Francisco.", vocab,
                             max_seq=24)
    assert len(wins) == 1
    w = wins[0]
    assert w.tokens[:7] == ["[CLS]", "where", "is", "the", "center", "?",
                            "[SEP]"]
    assert w.tokens[7:10] == ["the", "mos", "##cone"]
    assert w.input_ids.shape == (24,) and w.input_mask.sum() == len(w.tokens)
    assert w.segment_ids[:7].sum() == 0 and w.segment_ids[7] == 1
    assert w.token_to_word[8] == w.token_to_word[9] == 1   # "Moscone"
    # logits that point at "San Francisco."
    s = np.full(24, -5.0)
    e = np.full(24, -5.0)
    s[w.tokens.index("san")] = 4.0
    e[w.tokens.index(".")] = 3.0
    e[2] = 9.0            # inside the question: never an answer
    spans = st.best_spans(wins, words, [s], [e])
    assert spans[0] == ("San Francisco.", 7.0)
    # a long context is covered by overlapping windows
    long_ctx = " ".join(["the center in san francisco ."] * 30)
    wins, _ = st.windows("where ?", long_ctx, vocab, max_seq=32, stride=16)
    assert len(wins) > 5
    assert all(x.input_ids.shape == (32,) for x in wins)
    assert wins[1].token_to_word[wins[1].context_start] == \
        wins[0].token_to_word[wins[0].context_start + 16]


def test_byte_pair_tokenizer(zoo, tmp_path):
    bp = _text_module("", "bpe")
    import json
    b2c = bp._byte_alphabet()
    assert len(set(b2c.values())) == 256 and b2c[ord("a")] == "a"
    sp = b2c[ord(" ")]
    vocab = {c: i for i, c in enumerate(sorted(set(b2c.values())))}
    merges = [("l", "o"), ("lo", "w"), (sp, "low"), ("e", "r")]
    for a, b in merges:
        vocab.setdefault(a + b, len(vocab))
    json.dump(vocab, open(os.path.join(str(tmp_path), "vocab.json"), "w"))
    open(os.path.join(str(tmp_path), "merges.txt"), "w").write(
        "#version: 0.2\n" + "\n".join(a + " " + b for a, b in merges))
    tok = bp.BPETokenizer(str(tmp_path))
    ids = tok.encode("low lower")
    assert [tok.decoder[i] for i in ids] == ["low", sp + "low", "er"]
    assert tok.decode(ids) == "low lower"
    assert tok.decode(tok.encode("héllo, wörld 42!")) == "héllo, wörld 42!"
    bt = bp.ByteTokenizer()
    assert bt.decode(bt.encode("añb")) == "añb"
    g2 = _text_module("gpt2", "gpt2")
    rng = np.random.RandomState(0)
    logits = np.array([0.1, 5.0, 4.9, -3.0])
    assert g2.next_token(logits, 1, rng) == 1
    assert {g2.next_token(logits, 2, rng) for _ in range(40)} == {1, 2}
    out = g2.generate(lambda ids: np.eye(5)[(ids[-1] + 1) % 5], [0], 4)
    assert out == [1, 2, 3, 4]


# ---- a float32 transformer graph imported in 16 bits ---------------------------
@pytest.mark.parametrize("dtype_name,tol", [("float16", 0.02),
                                            ("bfloat16", 0.15)])
@pytest.mark.parametrize("graph", [False, True])
def test_bert_import_in_16_bits(zoo, dev, dtype_name, tol, graph):
    """Inference against the float64 evaluation of the same weights, then a
    few fine-tuning steps (float32 master weights) that must lower the loss;
    with ``graph`` the steps are recorded and replayed, which covers the
    conversion of the initializers while a step is being recorded."""
    from singa_amd import opt, tensor
    tm = importlib.import_module("text_models")
    dt = getattr(tensor, dtype_name)
    cfg = dict(seq=16, hidden=128, heads=2, layers=2, ffn=256, vocab=100)
    model, W = tm.bert_encoder(batch=2, squad_head=False, **cfg)
    rs = np.random.RandomState(0)
    ids = rs.randint(0, 100, (2, 16)).astype(np.int32)
    mask = np.ones((2, 16), np.int32)
    mask[1, 10:] = 0
    seg = np.zeros((2, 16), np.int32)
    seg[:, 8:] = 1
    xs = [tensor.from_numpy(a, dev) for a in (ids, mask, seg)]
    want = tm.bert_encoder_numpy(W, *[a.astype(np.int64)
                                      for a in (ids, mask, seg)], 2, 2)
    m = tm.TextModel(model, dtype=dt)
    m.compile(xs, is_train=False, use_graph=graph, sequential=True)
    m.eval()
    for _ in range(3 if graph else 1):
        y = m(*xs)[0]
    assert y.dtype == dt
    got = tensor.to_numpy(y.as_type(tensor.float32))
    assert np.abs(got - want).max() < tol * np.abs(want).max()
    dev.ResetGraph()
    ty = tensor.from_numpy(np.array([0, 1], np.int32), dev)
    clf = tm.BertClassifier(model, 2, dtype=dt)
    dev.SetRandSeed(7)
    clf.set_optimizer(opt.SGD(lr=2e-3, momentum=0.9))
    clf.compile(xs, is_train=True, use_graph=graph, sequential=False)
    clf.train()
    losses = []
    for _ in range(12):
        _, loss = clf(*xs, ty)
        losses.append(float(tensor.to_numpy(
            loss.as_type(tensor.float32)).reshape(-1)[0]))
    assert np.isfinite(losses).all() and losses[-1] < 0.5 * losses[0], losses
    params = clf.get_params()
    assert any(p.dtype == dt for p in params.values())
    dev.ResetGraph()


def test_recorded_transformer_reads_the_ids_of_each_batch(zoo, dev):
    """The embedding gathers take their indices from the device tensor the
    model is fed with: a replayed step must follow new token ids."""
    from singa_amd import tensor
    tm = importlib.import_module("text_models")
    cfg = dict(seq=8, hidden=64, heads=1, layers=1, ffn=128, vocab=50)
    model, W = tm.bert_encoder(batch=2, squad_head=False, **cfg)
    rs = np.random.RandomState(3)
    mask = np.ones((2, 8), np.int32)
    seg = np.zeros((2, 8), np.int32)
    ids = rs.randint(0, 50, (2, 8)).astype(np.int32)
    xs = [tensor.from_numpy(a, dev) for a in (ids, mask, seg)]
    m = tm.TextModel(model)
    m.compile(xs, is_train=False, use_graph=True, sequential=True)
    m.eval()
    for _ in range(4):
        ids = rs.randint(0, 50, (2, 8)).astype(np.int32)
        xs[0].copy_from_numpy(ids)
        got = tensor.to_numpy(m(*xs)[0])
        want = tm.bert_encoder_numpy(W, *[a.astype(np.int64)
                                          for a in (ids, mask, seg)], 1, 1)
        np.testing.assert_allclose(got, want, atol=2e-4)
    dev.ResetGraph()


def test_fused_transformer_exports_and_reimports(zoo, dev):
    """An imported encoder runs on the fused operators (attention, linear
    with bias, layer norm, GELU); exporting that computation writes the
    standard chains again, and importing the export reproduces the output."""
    from singa_amd import autograd, sonnx, tensor
    tm = importlib.import_module("text_models")
    cfg = dict(seq=16, hidden=128, heads=2, layers=1, ffn=256, vocab=60)
    model, W = tm.bert_encoder(batch=2, squad_head=False, **cfg)
    rs = np.random.RandomState(5)
    ids = rs.randint(0, 60, (2, 16)).astype(np.int32)
    mask = np.ones((2, 16), np.int32)
    mask[0, 12:] = 0
    seg = np.zeros((2, 16), np.int32)
    xs = [tensor.from_numpy(a, dev) for a in (ids, mask, seg)]
    m = tm.TextModel(model)
    m.compile(xs, is_train=False, use_graph=False, sequential=True)
    m.eval()
    y = m(*xs)[0]
    kinds = [type(o).__name__ for _, o in m.sg_ir.layers]
    assert "Attention" in kinds and "LinearBias" in kinds
    exported = sonnx.to_onnx(xs, [y])
    ops = [n.op_type for n in exported.graph.node]
    assert "Softmax" in ops and "MatMul" in ops       # standard nodes only
    assert not any(o.startswith("Fused") for o in ops)
    again = tm.TextModel(exported)
    again.compile(xs, is_train=False, use_graph=False, sequential=True)
    again.eval()
    y2 = again(*xs)[0]
    assert "Attention" in [type(o).__name__ for _, o in again.sg_ir.layers]
    np.testing.assert_allclose(tensor.to_numpy(y2), tensor.to_numpy(y),
                               atol=2e-5)
    want = tm.bert_encoder_numpy(W, *[a.astype(np.int64)
                                      for a in (ids, mask, seg)], 2, 1)
    np.testing.assert_allclose(tensor.to_numpy(y2), want, atol=2e-4)
