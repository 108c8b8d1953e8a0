"""Periodic checkpoints of a training job and resuming from the newest one.

The reference has the two building blocks (``Model.save_states`` /
``load_states`` with user-supplied ``aux_states``, python/singa/model.py:
244-354) but no periodic checkpointing and no resume path (its examples do
not checkpoint).  ``CheckpointManager`` adds them:

* a checkpoint is one ``Model.save_states`` file holding the model's states
  plus the optimizer's (step counter, moments, float32 master weights,
  flattened to ``opt/...`` names) and a few scalars of the training loop
  (``extra``: epoch, step, best accuracy, ...);
* files are written under a temporary name and renamed, so a job that dies
  while writing never leaves a half-written newest checkpoint;
* only rank 0 of a data-parallel job writes (all ranks hold the same
  states); every rank reads;
* the ``keep`` newest files are kept.

Together with the failure detection of ``singa_amd.parallel.health`` (a rank
that loses a peer leaves with status 75) and a launcher that starts the job
again on that status (``tools/launch_restart.py``) a job survives the loss of
a rank at the cost of the steps since the last checkpoint.
"""
import json
import os
import re
import zipfile

import numpy as np

from .. import tensor

_NAME = re.compile(r"^ckpt-(\d{10})\.zip$")
_OPT, _EXTRA = "opt/", "extra/"


def _flatten(states, prefix, dev, out):
    for key, value in states.items():
        name = "%s%s" % (prefix, key)
        if isinstance(value, dict):
            _flatten(value, name + "/", dev, out)
        elif isinstance(value, tensor.Tensor):
            out[name] = value
        else:       # python / numpy scalar
            out[name + "#scalar"] = tensor.from_numpy(
                np.asarray([value], np.float64).astype(np.float32), dev)


def _unflatten(flat, prefix):
    nested = {}
    for name, value in flat.items():
        if not name.startswith(prefix):
            continue
        parts = name[len(prefix):].split("/")
        leaf = parts[-1]
        node = nested
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        if leaf.endswith("#scalar"):
            node[leaf[:-len("#scalar")]] = float(
                tensor.to_numpy(value).reshape(-1)[0])
        else:
            node[leaf] = value
    return nested


class CheckpointManager(object):

    def __init__(self, directory, keep=2, rank=0):
        self.directory, self.keep, self.rank = directory, int(keep), int(rank)
        if self.rank == 0:
            os.makedirs(directory, exist_ok=True)

    # ---- files ------------------------------------------------------------
    def _path(self, step):
        return os.path.join(self.directory, "ckpt-%010d.zip" % step)

    def steps(self):
        """Steps of the checkpoints present, oldest first."""
        try:
            names = os.listdir(self.directory)
        except OSError:
            return []
        return sorted(int(m.group(1)) for m in map(_NAME.match, names) if m)

    def latest(self):
        """(path, step) of the newest checkpoint, or None."""
        steps = self.steps()
        return (self._path(steps[-1]), steps[-1]) if steps else None

    # ---- save / restore -----------------------------------------------------
    def save(self, model, step, optimizer=None, extra=None):
        """Write the checkpoint of ``step``.  ``optimizer``: the optimizer (or
        the DistOpt wrapping it); ``extra``: name -> number kept with it.
        Ranks other than 0 do nothing.  Returns the path."""
        path = self._path(step)
        if self.rank != 0:
            return path
        dev = next(iter(model.get_states().values())).device
        aux = {}
        inner = getattr(optimizer, "opt", optimizer)
        if inner is not None:
            _flatten(inner.get_states(), _OPT, dev, aux)
        _flatten(dict(extra or {}, step=step), _EXTRA, dev, aux)
        tmp = "%s.tmp.%d" % (path, os.getpid())
        if os.path.exists(tmp):
            os.remove(tmp)
        model.save_states(tmp, aux)
        os.replace(tmp, path)
        for old in self.steps()[:-self.keep] if self.keep > 0 else []:
            try:
                os.remove(self._path(old))
            except OSError:
                pass
        return path

    def restore(self, model, optimizer=None):
        """Load the newest checkpoint into ``model`` (and ``optimizer``).
        Returns its ``extra`` dictionary (with ``step``), or None when there
        is no checkpoint.  A file that cannot be read (the job died while an
        older version of this module was writing it) is skipped in favour of
        the one before."""
        for step in reversed(self.steps()):
            path = self._path(step)
            try:
                aux = model.load_states(path)
            except (OSError, ValueError, KeyError, json.JSONDecodeError,
                    AssertionError, EOFError, zipfile.BadZipFile):
                continue
            inner = getattr(optimizer, "opt", optimizer)
            if inner is not None:
                inner.set_states(_unflatten(aux, _OPT))
            extra = _unflatten(aux, _EXTRA)
            extra["step"] = int(extra.get("step", step))
            return extra
        return None
