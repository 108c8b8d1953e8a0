"""singa_amd: an MI355X-native deep-learning framework with the public API of
Apache SINGA (tensor, device, autograd, layer, model, opt, sonnx, ...).

Compute path: PyTorch-ROCm supplies memory and streams, every hot op is a
hand-written HIP kernel for gfx950 (``singa_amd/csrc/hip``), collectives run
on RCCL through ``torch.distributed`` and steady-state training steps replay
as hipGraphs.  The runtime around it (graph scheduler, record files,
snapshot, logging) is C++ (``singa_amd/csrc/runtime``).
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys


def _load_alternative_runtime():
    """SINGA_RT_LIBRARY=<path of an _rt build> makes the package use that
    build of the C++ runtime instead of the in-tree one (the sanitizer job
    points it at an AddressSanitizer / UBSan build, tool/ci/sanitize.sh)."""
    path = _os.environ.get("SINGA_RT_LIBRARY")
    if not path:
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location(__name__ + "._rt", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _sys.modules[__name__ + "._rt"] = mod
    globals()["_rt"] = mod


_load_alternative_runtime()

from . import singa_wrap  # noqa: F401
from . import device  # noqa: F401
from . import tensor  # noqa: F401
from . import autograd  # noqa: F401
from . import layer  # noqa: F401
from . import model  # noqa: F401
from . import opt  # noqa: F401
from . import initializer  # noqa: F401
from . import utils  # noqa: F401
