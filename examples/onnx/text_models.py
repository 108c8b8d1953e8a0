"""Transformer graphs shared by the text examples (gpt2/gpt2.py,
bert/bert-squad.py, ro_bert_a/ro_bert_a.py, training of a BERT classifier):
ONNX graph builders with random weights, their numpy evaluations, and the
Model wrappers around the imported graphs.

``tiny_gpt(...)`` builds a small GPT-style decoder directly as an ONNX graph
(embedding Gather, LayerNorm from primitives, multi-head causal attention, Erf
GELU MLP); ``--self-test`` runs it through the importer against a numpy
evaluation of the same weights, which covers the operator set the real models
use on a machine that has none of their files.
"""
import argparse
import logging

import numpy as np

import utils  # sets sys.path
from singa_amd import autograd, device, layer, sonnx, tensor
from singa_amd import onnx_proto as onnx
from singa_amd.onnx_proto import helper, numpy_helper

logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
F, I64 = onnx.DataType.FLOAT, onnx.DataType.INT64


class TextModel(sonnx.SONNXModel):

    def forward(self, *x):
        return super(TextModel, self).forward(*x)

    def train_one_batch(self, *a):
        raise NotImplementedError("inference wrapper; see BertClassifier")


class BertClassifier(sonnx.SONNXModel):
    """Fine-tuning of an imported BERT encoder: the graph's float
    initializers become trainable parameters, a Linear head reads the first
    token of the sequence output, and one training step back-propagates
    through the whole encoder (the counterpart of the reference's
    examples/onnx/training/train.py:105-142 for a transformer backbone)."""

    def __init__(self, onnx_model, num_classes=2, sequence_output=0,
                 dtype=None):
        super(BertClassifier, self).__init__(onnx_model,
                                             trainable_weights=True,
                                             dtype=dtype)
        self.sequence_output = sequence_output
        self.head = layer.Linear(num_classes)
        self.loss = layer.SoftMaxCrossEntropy()

    def forward(self, ids, mask, segments):
        seq = super(BertClassifier, self).forward(
            ids, mask, segments)[self.sequence_output]
        B, S, Hd = seq.shape
        # first token of every sequence: rows 0, S, 2 S, ... of (B * S, Hd)
        flat = autograd.reshape(seq, (B, S * Hd))
        cls = autograd.slice(flat, [0], [Hd], [1])
        return self.head(cls)

    def train_one_batch(self, ids, mask, segments, y):
        out = self.forward(ids, mask, segments)
        loss = self.loss(out, y)
        self.optimizer(loss)
        return out, loss


def create_device():
    return device.create_cuda_gpu() if device.get_num_gpus() \
        else device.get_default_device()


def run(onnx_model, inputs, dev):
    """Feed numpy inputs (ints as int32, floats as float32); numpy outputs."""
    m = TextModel(onnx_model)
    ts = [tensor.from_numpy(np.ascontiguousarray(
        a.astype(np.int32) if a.dtype.kind in "iu" else a.astype(np.float32)),
        dev) for a in inputs]
    m.compile(ts, is_train=False, use_graph=False, sequential=True)
    return [tensor.to_numpy(y) for y in m.forward(*ts)]


# --------------------------------------------------------------- tiny GPT
def tiny_gpt(vocab=50, seq=8, dim=16, heads=2, layers=2, seed=0):
    """-> (ModelProto, weights dict).  Input ``ids`` (1, seq) int64, output
    ``logits`` (1, seq, vocab)."""
    rng = np.random.RandomState(seed)
    W, nodes = {}, []
    uid = [0]

    def w(name, *shape, scale=0.2):
        W[name] = (rng.randn(*shape) * scale).astype(np.float32)
        return name

    def const(name, arr):
        W[name] = np.asarray(arr)
        return name

    def op(kind, ins, **attrs):
        uid[0] += 1
        out = "%s_%d" % (kind.lower(), uid[0])
        nodes.append(helper.make_node(kind, ins, [out], name=out, **attrs))
        return out

    def layer_norm(x, p):
        mu = op("ReduceMean", [x], axes=[-1], keepdims=1)
        d = op("Sub", [x, mu])
        var = op("ReduceMean", [op("Mul", [d, d])], axes=[-1], keepdims=1)
        std = op("Sqrt", [op("Add", [var, const(p + "eps",
                                                np.float32(1e-5))])])
        y = op("Div", [d, std])
        return op("Add", [op("Mul", [y, w(p + "g", dim, scale=1.0)]),
                          w(p + "b", dim)])

    def linear(x, p, i, o):
        return op("Add", [op("MatMul", [x, w(p + "w", i, o)]), w(p + "b", o)])

    hd = dim // heads
    x = op("Add", [op("Gather", [w("wte", vocab, dim), "ids"], axis=0),
                   w("wpe", 1, seq, dim)])
    mask = const("mask", np.triu(np.full((seq, seq), -1e9, np.float32), 1))
    shp = const("shape_heads", np.array([1, seq, heads, hd], np.int64))
    shp_back = const("shape_back", np.array([1, seq, dim], np.int64))
    for l in range(layers):
        p = "l%d_" % l
        h = layer_norm(x, p + "ln1_")
        qkv = []
        for n in "qkv":
            t = op("Reshape", [linear(h, p + n + "_", dim, dim), shp])
            qkv.append(op("Transpose", [t], perm=[0, 2, 1, 3]))
        kt = op("Transpose", [qkv[1]], perm=[0, 1, 3, 2])
        att = op("Mul", [op("MatMul", [qkv[0], kt]),
                         const(p + "scale", np.float32(1.0 / np.sqrt(hd)))])
        att = op("Softmax", [op("Add", [att, mask])], axis=-1)
        ctx = op("Transpose", [op("MatMul", [att, qkv[2]])],
                 perm=[0, 2, 1, 3])
        ctx = op("Reshape", [ctx, shp_back])
        x = op("Add", [x, linear(ctx, p + "proj_", dim, dim)])
        h = layer_norm(x, p + "ln2_")
        h = linear(h, p + "fc_", dim, 4 * dim)
        # exact GELU: 0.5 * h * (1 + erf(h / sqrt(2)))
        e = op("Erf", [op("Div", [h, const(p + "sqrt2",
                                           np.float32(np.sqrt(2.0)))])])
        h = op("Mul", [op("Mul", [h, const(p + "half", np.float32(0.5))]),
                       op("Add", [e, const(p + "one", np.float32(1.0))])])
        x = op("Add", [x, linear(h, p + "out_", 4 * dim, dim)])
    x = layer_norm(x, "lnf_")
    logits = op("MatMul", [x, w("head", dim, vocab)])
    nodes.append(helper.make_node("Identity", [logits], ["logits"],
                                  name="output"))
    graph = helper.make_graph(
        nodes, "tiny_gpt",
        [helper.make_tensor_value_info("ids", I64, [1, seq])],
        [helper.make_tensor_value_info("logits", F, [1, seq, vocab])],
        initializer=[numpy_helper.from_array(v, k) for k, v in W.items()])
    return helper.make_model(graph), W


def tiny_gpt_numpy(W, ids, heads=2, layers=2):
    from scipy.special import erf

    def ln(x, p):
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(var + 1e-5) * W[p + "g"] + W[p + "b"]

    x = W["wte"][ids] + W["wpe"]
    _, seq, dim = x.shape
    hd = dim // heads
    for l in range(layers):
        p = "l%d_" % l
        h = ln(x, p + "ln1_")
        q, k, v = [(h @ W[p + n + "_w"] + W[p + n + "_b"])
                   .reshape(1, seq, heads, hd).transpose(0, 2, 1, 3)
                   for n in "qkv"]
        att = q @ k.transpose(0, 1, 3, 2) / np.sqrt(hd) + W["mask"]
        att = utils.softmax(att, -1)
        ctx = (att @ v).transpose(0, 2, 1, 3).reshape(1, seq, dim)
        x = x + ctx @ W[p + "proj_w"] + W[p + "proj_b"]
        h = ln(x, p + "ln2_") @ W[p + "fc_w"] + W[p + "fc_b"]
        h = 0.5 * h * (1 + erf(h / np.sqrt(2.0)))
        x = x + h @ W[p + "out_w"] + W[p + "out_b"]
    return ln(x, "lnf_") @ W["head"]


# ------------------------------------------------------------------- BERT
def bert_encoder(batch=1, seq=256, hidden=768, heads=12, layers=12, ffn=3072,
                 vocab=30522, types=2, seed=0, squad_head=True):
    """-> (ModelProto, weights).  A BERT encoder written the way the model
    zoo's bert-squad graph is (reference examples/onnx/bert/bert-squad.py):
    embedding Gathers, layer normalisation and exact GELU from primitive
    nodes, attention as MatMul / Div / Add(mask) / Softmax / MatMul between
    Reshape + Transpose pairs.  Inputs ``input_ids``, ``input_mask``,
    ``segment_ids`` (batch, seq) int64; outputs ``sequence_output`` (batch,
    seq, hidden) and, with ``squad_head``, ``logits`` (batch, seq, 2).
    Weights are random (no checkpoint is needed to run or time it)."""
    rng = np.random.RandomState(seed)
    W, nodes = {}, []
    uid = [0]
    hd = hidden // heads

    def w(name, *shape, scale=0.02):
        W[name] = (rng.standard_normal(shape) * scale).astype(np.float32)
        return name

    def const(name, arr):
        W[name] = np.asarray(arr)
        return name

    def op(kind, ins, **attrs):
        uid[0] += 1
        out = "%s_%d" % (kind.lower(), uid[0])
        nodes.append(helper.make_node(kind, ins, [out], name=out, **attrs))
        return out

    two = const("two", np.float32(2.0))

    def layer_norm(x, p):
        mu = op("ReduceMean", [x], axes=[-1], keepdims=1)
        d = op("Sub", [x, mu])
        var = op("ReduceMean", [op("Pow", [d, two])], axes=[-1], keepdims=1)
        std = op("Sqrt", [op("Add", [var, const(p + "eps",
                                                np.float32(1e-12))])])
        y = op("Div", [d, std])
        W[p + "g"] = np.ones(hidden, np.float32) + \
            (rng.standard_normal(hidden) * 0.02).astype(np.float32)
        return op("Add", [op("Mul", [y, p + "g"]), w(p + "b", hidden)])

    def linear(x, p, i, o):
        return op("Add", [op("MatMul", [x, w(p + "w", i, o)]), w(p + "b", o)])

    x = op("Add", [op("Gather", [w("word_emb", vocab, hidden), "input_ids"],
                      axis=0),
                   op("Gather", [w("type_emb", types, hidden), "segment_ids"],
                      axis=0)])
    x = layer_norm(op("Add", [x, w("pos_emb", 1, seq, hidden)]), "emb_ln_")
    # additive attention mask (batch, 1, 1, seq): 0 keeps, -10000 hides
    m = op("Unsqueeze", [op("Cast", ["input_mask"], to=F)], axes=[1, 2])
    mask = op("Mul", [op("Sub", [const("one", np.float32(1.0)), m]),
                      const("neg", np.float32(-10000.0))])
    split = const("shape_split", np.array([batch, seq, heads, hd], np.int64))
    merge = const("shape_merge", np.array([batch, seq, hidden], np.int64))
    sqrt_d = const("sqrt_d", np.float32(np.sqrt(hd)))
    for l in range(layers):
        p = "l%d_" % l
        q = op("Transpose", [op("Reshape", [linear(x, p + "q_", hidden, hidden),
                                            split])], perm=[0, 2, 1, 3])
        k = op("Transpose", [op("Reshape", [linear(x, p + "k_", hidden, hidden),
                                            split])], perm=[0, 2, 3, 1])
        v = op("Transpose", [op("Reshape", [linear(x, p + "v_", hidden, hidden),
                                            split])], perm=[0, 2, 1, 3])
        att = op("Add", [op("Div", [op("MatMul", [q, k]), sqrt_d]), mask])
        att = op("Softmax", [att], axis=-1)
        ctx = op("Reshape", [op("Transpose", [op("MatMul", [att, v])],
                                perm=[0, 2, 1, 3]), merge])
        x = layer_norm(op("Add", [x, linear(ctx, p + "o_", hidden, hidden)]),
                       p + "ln1_")
        h = linear(x, p + "fc_", hidden, ffn)
        e = op("Erf", [op("Div", [h, const("sqrt2", np.float32(np.sqrt(2.0)))])])
        h = op("Mul", [op("Mul", [h, op("Add", [e, "one"])]),
                       const("half", np.float32(0.5))])
        x = layer_norm(op("Add", [x, linear(h, p + "out_", ffn, hidden)]),
                       p + "ln2_")
    nodes.append(helper.make_node("Identity", [x], ["sequence_output"],
                                  name="sequence_output_id"))
    outs = [helper.make_tensor_value_info("sequence_output", F,
                                          [batch, seq, hidden])]
    if squad_head:
        lg = linear(x, "squad_", hidden, 2)
        nodes.append(helper.make_node("Identity", [lg], ["logits"],
                                      name="logits_id"))
        outs.append(helper.make_tensor_value_info("logits", F,
                                                  [batch, seq, 2]))
    graph = helper.make_graph(
        nodes, "bert_encoder",
        [helper.make_tensor_value_info(n, I64, [batch, seq])
         for n in ("input_ids", "input_mask", "segment_ids")],
        outs,
        initializer=[numpy_helper.from_array(v, k) for k, v in W.items()])
    return helper.make_model(graph), W


def bert_encoder_numpy(W, ids, mask, seg, heads, layers):
    """float64 evaluation of ``bert_encoder`` with the same weights."""
    from scipy.special import erf
    Wd = {k: np.asarray(v, np.float64) for k, v in W.items()
          if np.asarray(v).dtype.kind == 'f'}

    def ln(x, p):
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(var + 1e-12) * Wd[p + "g"] + Wd[p + "b"]

    x = ln(Wd["word_emb"][ids] + Wd["type_emb"][seg] + Wd["pos_emb"],
           "emb_ln_")
    B, S, Hd = x.shape
    hd = Hd // heads
    am = (1.0 - mask[:, None, None, :].astype(np.float64)) * -10000.0
    for l in range(layers):
        p = "l%d_" % l
        q, k, v = [(x @ Wd[p + n + "_w"] + Wd[p + n + "_b"])
                   .reshape(B, S, heads, hd).transpose(0, 2, 1, 3)
                   for n in "qkv"]
        att = q @ k.transpose(0, 1, 3, 2) / np.sqrt(hd) + am
        att = np.exp(att - att.max(-1, keepdims=True))
        att /= att.sum(-1, keepdims=True)
        ctx = (att @ v).transpose(0, 2, 1, 3).reshape(B, S, Hd)
        x = ln(x + ctx @ Wd[p + "o_w"] + Wd[p + "o_b"], p + "ln1_")
        h = x @ Wd[p + "fc_w"] + Wd[p + "fc_b"]
        h = 0.5 * h * (1 + erf(h / np.sqrt(2.0)))
        x = ln(x + h @ Wd[p + "out_w"] + Wd[p + "out_b"], p + "ln2_")
    return x


def self_test(dev):
    model, W = tiny_gpt()
    ids = np.random.RandomState(1).randint(0, 50, (1, 8))
    got = run(model, [ids], dev)[0]
    want = tiny_gpt_numpy(W, ids)
    err = float(np.abs(got - want).max())
    logging.info("tiny GPT importer vs numpy: max abs diff %g", err)
    assert err < 1e-3, err
    return err


# The task scripts live next to this module: bert/bert-squad.py (WordPiece
# windows and span extraction in bert/squad_text.py), gpt2/gpt2.py and
# ro_bert_a/ro_bert_a.py (byte-pair tokenizer in bpe.py).


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("task", choices=["self-test"], nargs="?",
                    default="self-test")
    ap.parse_args(argv)
    err = self_test(create_device())
    print("self-test: max abs difference to the numpy reference %.3g" % err)
    return err


if __name__ == "__main__":
    main()
