"""Device factory functions of the public API.

The reference names (``create_cuda_gpu*``) are kept so existing scripts run
unchanged; they return MI355X devices.  Parity: reference
python/singa/device.py:29-136.
"""
from .core.device import Device, Platform, get_default_device as _default, HipGPU, CppCPU


def get_num_gpus():
    return Platform.GetNumGPUs()


def get_gpu_ids():
    return Platform.GetGPUIDs()


def get_gpu_mem_size(id):
    return Platform.GetGPUMemSize(id)


def device_query(id, verbose=False):
    return Platform.DeviceQuery(id, verbose)


def create_cuda_gpus(num):
    """Create ``num`` GPU devices (ids 0..num-1)."""
    return Platform.CreateCudaGPUs(num)


def create_cuda_gpu():
    """Create a single GPU device on GPU 0."""
    return Platform.CreateCudaGPUs(1)[0]


def create_cuda_gpus_on(device_ids):
    return Platform.CreateCudaGPUsOn(list(device_ids))


def create_cuda_gpu_on(device_id):
    """Create a device on the given GPU id (one process drives one GPU)."""
    return Platform.CreateCudaGPUsOn([device_id])[0]


# MI355X-flavoured aliases
create_hip_gpu = create_cuda_gpu
create_hip_gpu_on = create_cuda_gpu_on
create_hip_gpus = create_cuda_gpus
create_hip_gpus_on = create_cuda_gpus_on


def get_num_opencl_platforms():
    """1 when the machine's OpenCL runtime exposes a GPU, else 0 (reference
    python/singa/device.py:44-51)."""
    return 1 if get_num_opencl_devices() > 0 else 0


def get_num_opencl_devices():
    from . import native
    return native.num_opencl_devices()


def create_opencl_device(index=0):
    """The GPU as an OpenCL device of the native C++ library
    (``singa::OpenclDevice``).  The returned ``native.Device`` creates
    ``native.Tensor`` objects; the autograd / layer / model stack of this
    package runs on the HIP devices (``create_cuda_gpu``)."""
    from . import native
    return native.Device.opencl(index)


def create_cpu_device():
    return _default()


default_device = _default()


def get_default_device():
    """The host (CppCPU) singleton."""
    return default_device


def enbale_lazy_alloc(enable):
    """Spelling kept from the reference (python/singa/device.py:128)."""
    Device.EnableLazyAlloc(enable)


enable_lazy_alloc = enbale_lazy_alloc
