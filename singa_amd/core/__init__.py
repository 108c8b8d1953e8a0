"""The layer under the Python API: devices and graph replay (``device``), the
core tensor and its free functions (``tensor_core``), the math backends
(``math_hip`` over the HIP kernels, ``math_cpu`` on the host), NN operation
handles (``nn_ops``, ``rnn_ops``) and data types (``dtypes``)."""
