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
        run = self.train_one_batch if self.training else self.forward
        return run(*input, **kwargs)

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
