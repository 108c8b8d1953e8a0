"""Model: a Layer with a training step that can be buffered as a graph.

``train_one_batch`` of every subclass is wrapped: in graph mode the first call
executes the Python body once with graph recording on (ops are recorded, not
run), every call (including the first) then replays the recorded graph on the
device and returns the cached output tensors.  Inputs must therefore be the
same tensor objects on every call; their contents are refreshed in place.

API parity: reference python/singa/model.py (ModelMeta.buffer_operation
:39-100, compile :156-184, save_states / load_states :244-354: a zip holding
``tensor_dict.npz`` and ``states_attr.json``).
"""
import gc
import json
import os
import tempfile
import zipfile
from collections.abc import Iterable
from functools import wraps

import numpy as np

from . import autograd
from . import layer
from . import tensor
from .tensor import Tensor


class ModelMeta(layer.LayerMeta):

    def buffer_operation(func):

        def remove_creator(tensors):
            if not tensors:
                return
            if isinstance(tensors, tensor.Tensor):
                tensors.creator = None
            elif isinstance(tensors, Iterable):
                for item in tensors:
                    if isinstance(item, tensor.Tensor):
                        item.creator = None
                    elif isinstance(item, Iterable) and not isinstance(
                            item, (str, bytes)):
                        remove_creator(item)

        @wraps(func)
        def wrapper(self, *args, **kwargs):
            if self.graph_mode and self.training:
                if len(args) == 0:
                    raise ValueError('expect at least one input tensor')
                first = args[0][0] if isinstance(args[0], list) else args[0]
                assert isinstance(first, Tensor), (
                    'function expects PlaceHolders or Tensors')
                dev = first.device

                if not self._buffered:
                    # record the step: the python body runs exactly once
                    dev.EnableGraph(True)
                    self._results = func(self, *args, **kwargs)
                    dev.Sync()
                    dev.EnableGraph(False)
                    self._buffered = True
                    # drop the autograd tape so that only tensors the user can
                    # still reach pin their blocks
                    remove_creator(self._results)
                    gc.collect()

                dev.RunGraph(self.sequential)
                return self._results
            else:
                return func(self, *args, **kwargs)

        return wrapper

    def __new__(cls, name, bases, attr):
        if 'train_one_batch' in attr:
            attr['train_one_batch'] = ModelMeta.buffer_operation(
                attr['train_one_batch'])
        return super(ModelMeta, cls).__new__(cls, name, bases, attr)


class Model(layer.Layer, metaclass=ModelMeta):
    """Base class of user models; subclass and define ``forward`` and
    ``train_one_batch``."""

    TENSOR_DICT_FILENAME = '/tensor_dict.npz'
    STATES_ATTR_FILENAME = '/states_attr.json'
    MODEL_STATE_TYPE = 0
    AUX_STATE_TYPE = 1

    def __init__(self):
        super(Model, self).__init__()
        self.training = True
        self.graph_mode = True
        self.sequential = False
        self._buffered = False
        self._results = None

    def compile(self, inputs, is_train=True, use_graph=False, sequential=False):
        """Derive parameter shapes from the input placeholders with a dry
        forward pass (recorded, then discarded) and set the execution mode."""
        assert len(inputs) > 0 and isinstance(inputs[0], Tensor), (
            'compile function expects PlaceHolders or Tensors')
        dev = inputs[0].device
        dev.EnableGraph(True)
        self.forward(*inputs)
        dev.EnableGraph(False)
        dev.ResetGraph()

        autograd.training = is_train
        self.training = is_train
        self.graph_mode = use_graph
        self.sequential = sequential

    def forward(self, *input):
        raise NotImplementedError

    def train_one_batch(self, *input, **kwargs):
        raise NotImplementedError

    def train(self, mode=True):
        self.training = mode
        autograd.training = mode

    def eval(self):
        self.train(mode=False)

    def graph(self, mode=True, sequential=False):
        self.graph_mode = mode
        self.sequential = sequential

    def __get_name__(self):
        return self.__class__.__name__

    def __call__(self, *input, **kwargs):
        if self.training:
            return self.train_one_batch(*input, **kwargs)
        return self.forward(*input, **kwargs)

    def set_optimizer(self, optimizer):
        self.optimizer = optimizer

    def save_states(self, fpath, aux_states={}):
        """Write model states plus ``aux_states`` (name -> Tensor, e.g.
        optimizer moments) to the zip file ``fpath``; refuses to overwrite."""
        assert not os.path.isfile(fpath), (
            "Failed to save states, %s is already existed." % fpath)
        states = self.get_states()
        tensor_dict = {}
        states_attr = {}
        for kind, group in ((self.MODEL_STATE_TYPE, states),
                            (self.AUX_STATE_TYPE, aux_states)):
            for k, v in group.items():
                assert isinstance(v, tensor.Tensor), \
                    "Only tensor state is allowed"
                tensor_dict[k] = tensor.to_numpy(v)
                states_attr[k] = {
                    'state_type': kind,
                    'shape': v.shape,
                    'dtype': v.dtype
                }
        tmp_dir = tempfile.mkdtemp(prefix='singa_save_states_')
        tensor_dict_fp = tmp_dir + self.TENSOR_DICT_FILENAME
        states_attr_fp = tmp_dir + self.STATES_ATTR_FILENAME
        np.savez(tensor_dict_fp, **tensor_dict)
        with open(states_attr_fp, 'w') as fp:
            json.dump(states_attr, fp)
        with zipfile.ZipFile(fpath, mode="w") as zf:
            for p in (tensor_dict_fp, states_attr_fp):
                zf.write(p, os.path.basename(p),
                         compress_type=zipfile.ZIP_DEFLATED)
        os.remove(tensor_dict_fp)
        os.remove(states_attr_fp)
        os.rmdir(tmp_dir)

    def load_states(self, fpath):
        """Restore model states from ``fpath``; returns the aux states."""
        assert os.path.isfile(fpath), (
            "Failed to load states, %s is not exist." % fpath)
        tmp_dir = tempfile.mkdtemp(prefix='singa_load_states_')
        with zipfile.ZipFile(fpath, 'r') as zf:
            zf.extractall(tmp_dir)
        tensor_dict_fp = tmp_dir + self.TENSOR_DICT_FILENAME
        states_attr_fp = tmp_dir + self.STATES_ATTR_FILENAME
        with open(states_attr_fp) as f:
            states_attr = json.load(f)
        tensor_dict = np.load(tensor_dict_fp)
        model_states = dict()
        aux_states = dict()
        for k in tensor_dict.files:
            if states_attr[k]['state_type'] == self.MODEL_STATE_TYPE:
                model_states[k] = tensor.from_numpy(tensor_dict[k])
            elif states_attr[k]['state_type'] == self.AUX_STATE_TYPE:
                aux_states[k] = tensor.from_numpy(tensor_dict[k])
        tensor_dict.close()
        self.set_states(model_states)
        # float32 master copies held by the optimizer describe the old values
        forget = getattr(getattr(self, "optimizer", None), "forget_masters",
                         None)
        if forget is None:
            forget = getattr(getattr(getattr(self, "optimizer", None), "opt",
                                     None), "forget_masters", None)
        if forget is not None:
            forget()
        os.remove(tensor_dict_fp)
        os.remove(states_attr_fp)
        os.rmdir(tmp_dir)
        return aux_states
