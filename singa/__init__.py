"""Drop-in import name: ``from singa import tensor, autograd, layer, ...``
resolves to the same module objects as ``singa_amd.<name>`` so programs written
against the reference's python/singa package run unchanged on MI355X.

Submodules are aliased lazily through a meta-path finder: ``singa.x.y`` is
``singa_amd.x.y`` (one module object, two names), nothing is imported twice.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import singa_amd as _impl

__version__ = getattr(_impl, "__version__", "0")


class _AliasLoader(importlib.abc.Loader):

    def __init__(self, target):
        self._target = target

    def create_module(self, spec):
        mod = importlib.import_module(self._target)
        self._spec = mod.__spec__
        return mod

    def exec_module(self, module):
        # already executed under its real name; the import machinery has just
        # stamped the alias spec on it, put the real one back
        module.__spec__ = self._spec


class _AliasFinder(importlib.abc.MetaPathFinder):

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("singa."):
            return None
        real = "singa_amd." + fullname[len("singa."):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())


def __getattr__(name):
    try:
        return importlib.import_module("singa." + name)
    except ImportError:
        raise AttributeError(name)
