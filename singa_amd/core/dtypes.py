"""Data-type codes shared by every layer of the framework.

The integer values of the first six codes are those of the reference's
``DataType`` enum (reference src/proto/core.proto:26-34) because they are part
of the checkpoint format and of the public Python API (``tensor.float32 == 0``).
``kBFloat16`` is an addition for MI355X, whose matrix cores run bf16 and fp16
at the same rate.
"""
import numpy as np
import torch

kFloat32 = 0
kFloat16 = 1
kInt = 2
kChar = 3
kDouble = 4
kUChar = 5
kBFloat16 = 7

_TORCH = {
    kFloat32: torch.float32,
    kFloat16: torch.float16,
    kInt: torch.int32,
    kChar: torch.int8,
    kDouble: torch.float64,
    kUChar: torch.uint8,
    kBFloat16: torch.bfloat16,
}
_FROM_TORCH = {v: k for k, v in _TORCH.items()}
_SIZE = {kFloat32: 4, kFloat16: 2, kInt: 4, kChar: 1, kDouble: 8, kUChar: 1,
         kBFloat16: 2}
_NUMPY = {kFloat32: np.float32, kFloat16: np.float16, kInt: np.int32,
          kChar: np.int8, kDouble: np.float64, kUChar: np.uint8}
_NAMES = {kFloat32: "float32", kFloat16: "float16", kInt: "int32",
          kChar: "char", kDouble: "double", kUChar: "uchar",
          kBFloat16: "bfloat16"}


def to_torch(dtype):
    return _TORCH[dtype]


def from_torch(tdtype):
    return _FROM_TORCH[tdtype]


def SizeOf(dtype):
    return _SIZE[dtype]


def to_numpy_dtype(dtype):
    return _NUMPY[dtype]


def name(dtype):
    return _NAMES.get(dtype, str(dtype))


def is_float(dtype):
    return dtype in (kFloat32, kFloat16, kDouble, kBFloat16)
