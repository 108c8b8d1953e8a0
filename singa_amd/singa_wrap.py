"""Flat module with the names user code and the Python API layer expect from
the reference's SWIG module ``singa_wrap`` (reference src/api/*.i): the core
``Tensor``, ``Device``/``Platform``, tensor free functions, NN handles,
the communicator and build flags.  ``USE_CUDA`` reports whether an MI355X is
usable, since the reference's scripts use it to pick the accelerator path.
"""
import torch as _torch

from . import _rt
from .core.dtypes import (kFloat32, kFloat16, kInt, kChar, kDouble, kUChar,
                          kBFloat16)
from .core.device import (Device, CppCPU, HipGPU, Platform, kCpp, kHip,
                          get_default_device)
from .core.tensor_core import *  # noqa: F401,F403
from .core.tensor_core import (Tensor, VecTensor, Product, SizeOf, __add__,
                               __sub__, __mul__, __div__, __lt__, __le__,
                               __gt__, __ge__, __eq__, _submit)
from .core.nn_ops import *  # noqa: F401,F403
from .core.rnn_ops import *  # noqa: F401,F403
from .parallel.communicator import Communicator, NcclIdHolder

CudaGPU = HipGPU
kCuda = kHip

USE_CUDA = bool(_torch.cuda.is_available())
USE_HIP = USE_CUDA
USE_CUDNN = USE_CUDA      # conv / bn / pool / rnn handles are available
USE_DNNL = True           # the host path implements conv / bn / pool as well
USE_OPENCL = False   # Python tensors live on HIP; the OpenCL device is the C++ library's (singa_amd.native)
USE_PYTHON = True
USE_JAVA = False     # the Java classes (java/) bind the C interface at run time
USE_DIST = True
CUDNN_VERSION = 0
SINGA_MAJOR_VERSION = 4
SINGA_MINOR_VERSION = 2
SINGA_PATCH_VERSION = 0
SINGA_VERSION = 4200


def InitLogging(name="singa_amd"):
    _rt.InitLogging(str(name))


class VecInt(list):
    pass


class VecVecSize(list):
    pass


class PairSizeT(tuple):
    pass
