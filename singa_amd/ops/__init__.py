"""Operators built on several core operations: ``rnn_cells`` (a whole unrolled
LSTM as one autograd operator over the persistent sequence kernels) and
``hip_loader`` (locates and loads the in-tree ``_hip`` extension, failing
loudly on a GPU machine without it)."""
