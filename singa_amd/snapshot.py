"""Named-tensor checkpoints in the reference's snapshot format: ``<prefix>.bin``
holds one TensorProto record per parameter in a key/value record file and
``<prefix>.desc`` a readable line per parameter (reference
python/singa/snapshot.py, src/io/snapshot.cc).  Encoding, decoding and file
handling are native (singa_amd/csrc/runtime/snapshot.cc)."""
import numpy as np

from . import _rt
from . import tensor
from .core import dtypes


class Snapshot(object):
    """``Snapshot(prefix, True)`` writes, ``Snapshot(prefix, False)`` reads."""

    def __init__(self, f, mode, buffer_size=10):
        self.snapshot = _rt.Snapshot(f, bool(mode), int(buffer_size))
        self.mode = bool(mode)

    def write(self, param_name, param_val):
        """Append tensor ``param_val`` under ``param_name``."""
        a = np.ascontiguousarray(tensor.to_numpy(param_val))
        st = _rt.SnapshotTensor()
        st.shape = list(a.shape)
        st.dtype = param_val.dtype
        st.data = a.tobytes()
        self.snapshot.Write(param_name, st)

    def read(self):
        """dict name -> Tensor (on the host) of everything in the file."""
        out = {}
        for name in self.snapshot.Names():
            st = self.snapshot.Read(name)
            npdt = dtypes.to_numpy_dtype(st.dtype)
            a = np.frombuffer(st.data, dtype=npdt).reshape(list(st.shape))
            out[name] = tensor.from_numpy(a.copy())
        return out

    def close(self):
        self.snapshot.Close()

    def __del__(self):
        try:
            self.snapshot.Close()
        except Exception:
            pass
