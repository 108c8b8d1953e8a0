"""Loads the in-tree HIP kernel extension.  There is no fallback: on a GPU
device every hot op runs a hand-written kernel, so a missing extension is an
error rather than a silent slow path."""
import importlib

_mod = None


def load():
    global _mod
    if _mod is None:
        try:
            _mod = importlib.import_module("singa_amd._hip")
        except ImportError as e:
            raise ImportError(
                "singa_amd._hip is not built; run `python -m singa_amd.build` "
                "(needs hipcc, targets gfx950): %s" % e)
    return _mod
