"""Model: a Layer whose training step can be recorded once and replayed.

A subclass defines ``forward`` and ``train_one_batch``.  When the model runs in
graph mode, the first training call executes the Python body exactly once
with the device recording (every tensor op lands on the device graph instead
of running); that call and all later ones then ask the device to replay the
recorded step, which on the MI355X is a single hipGraph launch, and hand back
the output tensors of the recording.  The caller therefore feeds the same
input tensor objects every iteration and refreshes their contents in place.

The step's life cycle is owned by one ``_RecordedStep`` object per model
(idle -> recorded -> replaying; ``Model.compile`` / ``graph`` / ``train`` put
it back to idle), not inferred from call counts.

API parity: reference python/singa/model.py (graph buffering of
``train_one_batch`` :39-100, ``compile`` :156-184, ``save_states`` /
``load_states`` :244-354).  Checkpoints keep the reference's container: a zip
with ``tensor_dict.npz`` (one array per state) and ``states_attr.json``
(state type, shape and dtype per name).
"""
import contextlib
import functools
import gc
import io
import json
import os
import zipfile

import numpy as np

from . import autograd
from . import layer
from . import tensor
from .tensor import Tensor


def _walk_tensors(obj):
    """Every Tensor inside a (possibly nested) result structure."""
    if isinstance(obj, Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _walk_tensors(v)
    elif isinstance(obj, (list, tuple, set)):
        for v in obj:
            yield from _walk_tensors(v)


@contextlib.contextmanager
def _recording(dev):
    """Operations issued inside are recorded on ``dev`` instead of run."""
    dev.EnableGraph(True)
    try:
        yield dev
    finally:
        dev.EnableGraph(False)


def _first_tensor(args):
    for t in _walk_tensors(args):
        return t
    return None


class _RecordedStep(object):
    """Recording and replay of one model's training step."""

    def __init__(self):
        self.outputs = None
        self.recorded = False

    def reset(self):
        self.outputs = None
        self.recorded = False

    def run(self, model, body, args, kwargs):
        lead = _first_tensor(args[:1])
        if lead is None:
            raise ValueError('a graph-mode training step takes at least one '
                             'input tensor (first positional argument)')
        dev = lead.device
        if not self.recorded:
            with _recording(dev):
                self.outputs = body(model, *args, **kwargs)
                dev.Sync()           # recorded as the step's closing sync node
            self.recorded = True
            # The outputs keep their storage but let go of the autograd tape:
            # afterwards only tensors the user can still reach pin blocks, and
            # the scheduler may recycle everything else between replays.
            for t in _walk_tensors(self.outputs):
                t.creator = None
            gc.collect()
        dev.RunGraph(model.sequential)
        return self.outputs


class _RecordedInference(object):
    """The forward pass of a model in evaluation mode, recorded once per input
    signature (shapes, types, non-tensor arguments) and replayed afterwards --
    on a GPU as one captured hipGraph, so that a small-batch inference costs
    one launch instead of one Python dispatch per operator.  Every signature
    owns a graph state of the device (``Device.new_graph_state``), next to the
    one of the training step.  Inputs passed later are copied into the
    tensors the recording read, unless they are those tensors.

    Only meaningful for a forward pass whose sequence of operators does not
    depend on tensor values read on the host.  The reference never records
    evaluation (python/singa/model.py:214-228 dispatches it eagerly)."""

    def __init__(self):
        self.plans = {}

    def reset(self):
        self.plans = {}

    @staticmethod
    def _signature(args, kwargs):
        sig = []
        for a in list(args) + sorted(kwargs.items()):
            ts = list(_walk_tensors([a]))
            if ts:
                sig.append(tuple((tuple(t.shape), t.dtype) for t in ts))
            else:
                sig.append(repr(a))
        return tuple(sig)

    def run(self, model, args, kwargs):
        tensors = list(_walk_tensors([list(args), kwargs]))
        assert tensors, 'a recorded inference takes at least one tensor'
        dev = tensors[0].device
        key = self._signature(args, kwargs)
        plan = self.plans.get(key)
        if plan is None:
            plan = dict(state=dev.new_graph_state(), inputs=tensors)
            with dev.graph_state(plan["state"]):
                with _recording(dev):
                    plan["outputs"] = model.forward(*args, **kwargs)
                    dev.Sync()
            for t in _walk_tensors(plan["outputs"]):
                t.creator = None
            self.plans[key] = plan
        else:
            for dst, src in zip(plan["inputs"], tensors):
                if dst is not src and dst.data is not src.data:
                    dst.copy_data(src)
        with dev.graph_state(plan["state"]):
            dev.RunGraph(model.sequential)
        return plan["outputs"]


def _graphed(body):
    """Wrap ``train_one_batch``: replay through the model's recorded step in
    graph mode, plain call otherwise."""

    @functools.wraps(body)
    def train_one_batch(self, *args, **kwargs):
        if self.graph_mode and self.training:
            return self._step.run(self, body, args, kwargs)
        return body(self, *args, **kwargs)

    train_one_batch._is_graphed = True
    return train_one_batch


class ModelMeta(layer.LayerMeta):
    """Wraps the ``train_one_batch`` a subclass defines."""

    def __new__(mcs, name, bases, namespace):
        step = namespace.get('train_one_batch')
        if step is not None and not getattr(step, '_is_graphed', False):
            namespace['train_one_batch'] = _graphed(step)
        return super(ModelMeta, mcs).__new__(mcs, name, bases, namespace)


class Model(layer.Layer, metaclass=ModelMeta):
    """Base class of user models; subclass and define ``forward`` and
    ``train_one_batch``."""

    TENSOR_DICT_FILENAME = '/tensor_dict.npz'
    STATES_ATTR_FILENAME = '/states_attr.json'
    MODEL_STATE_TYPE = 0
    AUX_STATE_TYPE = 1

    def __init__(self):
        super(Model, self).__init__()
        self.training, self.graph_mode, self.sequential = True, True, False
        self._step = _RecordedStep()
        # evaluation through a recorded forward pass as well (graph mode
        # only); off unless asked for, see _RecordedInference
        self.graph_inference = os.environ.get("SINGA_GRAPH_INFERENCE",
                                              "0") == "1"
        self._inference = _RecordedInference()

    # ---- execution mode ---------------------------------------------------------
    def compile(self, inputs, is_train=True, use_graph=False, sequential=False):
        """Create the parameters from the shapes of ``inputs`` and choose the
        execution mode.  The forward pass runs with the device recording, so
        only the (eagerly executed) parameter initialisation costs anything;
        the recorded operations are thrown away."""
        lead = _first_tensor(list(inputs)[:1])
        assert lead is not None, 'compile takes a list of input tensors'
        try:
            with _recording(lead.device):
                self.forward(*inputs)
        finally:
            lead.device.ResetGraph()
        self._step.reset()
        self._inference.reset()
        self.train(is_train)
        self.graph(use_graph, sequential)

    def forward(self, *input):
        raise NotImplementedError

    def train_one_batch(self, *input, **kwargs):
        raise NotImplementedError

    def train(self, mode=True):
        self.training = bool(mode)
        autograd.training = bool(mode)

    def eval(self):
        self.train(False)

    def graph(self, mode=True, sequential=False):
        self.graph_mode, self.sequential = mode, sequential

    def __get_name__(self):
        return type(self).__name__

    def __call__(self, *input, **kwargs):
        if self.training:
            return self.train_one_batch(*input, **kwargs)
        if self.graph_mode and self.graph_inference:
            return self._inference.run(self, input, kwargs)
        return self.forward(*input, **kwargs)

    def set_optimizer(self, optimizer):
        self.optimizer = optimizer

    # ---- checkpoints --------------------------------------------------------------
    def save_states(self, fpath, aux_states={}):
        """Write the model's states and ``aux_states`` (name -> Tensor, e.g.
        optimizer moments) to the zip file ``fpath``.  An existing file is
        never overwritten."""
        assert not os.path.exists(fpath), \
            "refusing to overwrite the existing checkpoint %s" % fpath
        arrays, attrs = {}, {}
        tagged = [(self.MODEL_STATE_TYPE, self.get_states()),
                  (self.AUX_STATE_TYPE, aux_states)]
        for state_type, states in tagged:
            for name, value in states.items():
                assert isinstance(value, Tensor), "Only tensor state is allowed"
                arrays[name] = tensor.to_numpy(value)
                attrs[name] = dict(state_type=state_type, shape=value.shape,
                                   dtype=value.dtype)
        blob = io.BytesIO()
        np.savez(blob, **arrays)
        with zipfile.ZipFile(fpath, mode="w",
                             compression=zipfile.ZIP_DEFLATED) as archive:
            archive.writestr(self.TENSOR_DICT_FILENAME.lstrip('/'),
                             blob.getvalue())
            archive.writestr(self.STATES_ATTR_FILENAME.lstrip('/'),
                             json.dumps(attrs))

    def load_states(self, fpath):
        """Restore the model's states from ``fpath`` and return the auxiliary
        states stored with them (name -> Tensor)."""
        assert os.path.exists(fpath), "no checkpoint at %s" % fpath
        with zipfile.ZipFile(fpath, 'r') as archive:
            attrs = json.loads(archive.read(
                self.STATES_ATTR_FILENAME.lstrip('/')).decode())
            arrays = np.load(io.BytesIO(archive.read(
                self.TENSOR_DICT_FILENAME.lstrip('/'))))
            by_type = {self.MODEL_STATE_TYPE: {}, self.AUX_STATE_TYPE: {}}
            for name in arrays.files:
                kind = attrs[name]['state_type']
                if kind in by_type:
                    by_type[kind][name] = tensor.from_numpy(arrays[name])
        self.set_states(by_type[self.MODEL_STATE_TYPE])
        # A recorded (or captured) step keeps reading the optimizer's own
        # tensors: anything it derives from the parameters -- float32 master
        # copies of 16-bit weights -- is refreshed in place, never rebound.
        opt_ = getattr(self, "optimizer", None)
        for holder in (opt_, getattr(opt_, "opt", None)):
            sync = getattr(holder, "sync_masters", None)
            if sync is not None:
                sync(self.get_params())
        return by_type[self.AUX_STATE_TYPE]
