"""Transformer language models through the ONNX importer: GPT-2 generation,
BERT-SQuAD span extraction, RoBERTa sequence classification (counterparts of
the reference's examples/onnx/gpt2/gpt2.py, bert/bert-squad.py and
ro_bert_a/ro_bert_a.py).  Models and vocabularies are read from local paths.

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
from singa_amd import device, sonnx, tensor
from singa_amd import onnx_proto as onnx
from singa_amd.onnx_proto import helper, numpy_helper

logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
F, I64 = onnx.DataType.FLOAT, onnx.DataType.INT64


class TextModel(sonnx.SONNXModel):

    def forward(self, *x):
        return super(TextModel, self).forward(*x)

    def train_one_batch(self, *a):
        pass


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


def self_test(dev):
    model, W = tiny_gpt()
    ids = np.random.RandomState(1).randint(0, 50, (1, 8))
    got = run(model, [ids], dev)[0]
    want = tiny_gpt_numpy(W, ids)
    err = float(np.abs(got - want).max())
    logging.info("tiny GPT importer vs numpy: max abs diff %g", err)
    assert err < 1e-3, err
    return err


# ---------------------------------------------------------- real models
def gpt2_generate(model_path, vocab_dir, prompt, length, dev):
    """Greedy decoding with the zoo's gpt2-lm-head (no KV cache: the graph is
    re-run on the growing sequence, as the reference's script does)."""
    from transformers import GPT2Tokenizer
    tok = GPT2Tokenizer.from_pretrained(vocab_dir, local_files_only=True)
    ids = tok.encode(prompt)
    model = onnx.load(model_path)
    for _ in range(length):
        x = np.asarray(ids, np.int64).reshape(1, 1, -1)
        logits = run(model, [x], dev)[0]
        ids.append(int(np.argmax(logits.reshape(-1, logits.shape[-1])[-1])))
    return tok.decode(ids)


def bert_squad(model_path, vocab_file, question, context, dev, max_seq=256):
    from transformers import BertTokenizer
    tok = BertTokenizer(vocab_file)
    enc = tok(question, context, max_length=max_seq, padding="max_length",
              truncation="only_second", return_tensors="np")
    inputs = [np.array([0], np.int64), enc["token_type_ids"],
              enc["attention_mask"], enc["input_ids"]]
    outs = run(onnx.load(model_path), inputs, dev)
    end, start = [o.reshape(-1) for o in outs[:2]] if len(outs) > 2 \
        else (outs[1].reshape(-1), outs[0].reshape(-1))
    s = int(np.argmax(start))
    e = int(s + np.argmax(end[s:s + 30]))
    return tok.decode(enc["input_ids"][0][s:e + 1])


def roberta_classify(model_path, vocab_dir, text, dev):
    from transformers import RobertaTokenizer
    tok = RobertaTokenizer.from_pretrained(vocab_dir, local_files_only=True)
    ids = np.asarray(tok.encode(text, add_special_tokens=True),
                     np.int64)[None]
    logits = run(onnx.load(model_path), [ids], dev)[0]
    return int(np.argmax(logits))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("task", choices=["gpt2", "bert", "roberta", "self-test"])
    ap.add_argument("--model")
    ap.add_argument("--vocab", help="tokenizer directory or vocab file")
    ap.add_argument("--text", default="Here is some text to encode : Hello")
    ap.add_argument("--question", default="What is the framework built on?")
    ap.add_argument("--length", type=int, default=20)
    args = ap.parse_args(argv)
    dev = create_device()
    if args.task == "self-test":
        err = self_test(dev)
        print("self-test: max abs difference to the numpy reference %.3g" % err)
        return err
    if args.model is None or args.vocab is None:
        ap.error("%s needs --model (an .onnx file) and --vocab (tokenizer "
                 "files); this build never downloads them" % args.task)
    if args.task == "gpt2":
        out = gpt2_generate(args.model, args.vocab, args.text, args.length,
                            dev)
    elif args.task == "bert":
        out = bert_squad(args.model, args.vocab, args.question, args.text,
                         dev)
    else:
        out = roberta_classify(args.model, args.vocab, args.text, dev)
    print(out)
    return out


if __name__ == "__main__":
    main()
