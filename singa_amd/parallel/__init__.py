"""Data-parallel communication: ``Communicator`` (fused / half-precision /
sparsified gradient all-reduce over RCCL on GPUs, gloo on hosts) and
``NcclIdHolder`` (rendezvous token for ranks started with multiprocessing).
The optimizer-side driver is ``singa_amd.opt.DistOpt``."""
from .communicator import Communicator, NcclIdHolder  # noqa: F401
