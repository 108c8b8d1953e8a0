"""Windows of device memory that the GPUs of one node read from each other.

A rank allocates one *window* (uncached device memory) and one signal area,
exports both as IPC handles and maps every other rank's pair into its own
address space.  The all-reduce / all-gather kernels of ``csrc/hip/xgmi.hip``
then read the peers' windows directly: on an MI355X node every GPU has its own
xGMI link to each of the seven others, so a kernel that reads from all peers at
once uses all seven links, which a ring collective does not.  The reference
knows one collective path, the vendor library's ring all-reduce
(src/io/communicator.cc:255-301).

Buffers that take part in a collective are carved out of the window
(``tensor``): the pack kernel of a gradient bucket writes straight into it and
the unpack kernel reads the sum from the same place, so the collective adds no
copy.  Every rank must carve the same sequence of buffers (they are replicas of
one program), which gives a buffer the same offset in every window.

The kernels are ordinary launches on the caller's stream.  They can be captured
into a hipGraph: barrier epochs live in device memory.  A barrier wait that
exceeds ``timeout`` seconds sets a sticky error word; ``check`` raises for it.
"""
import os
import socket

import torch

# element type codes of the HIP kernels (csrc/hip/common.cuh)
_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 7}


class _Span(object):
    """Memory of the window seen through ``__cuda_array_interface__``."""

    def __init__(self, owner, ptr, n, typestr):
        self.owner = owner       # keeps the window alive
        self.__cuda_array_interface__ = {
            "shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
            "version": 2, "strides": None}


_TYPESTR = {torch.float32: "<f4", torch.float16: "<f2", torch.int32: "<i4",
            torch.uint8: "|u1"}


class XgmiWindow(object):
    """``exchange(obj)`` must return the list of every rank's ``obj`` in rank
    order (the host control group's all-gather)."""

    ALIGN = 256

    def __init__(self, rank, world, device, window_bytes, exchange,
                 timeout=30.0):
        from ..ops import hip_loader
        self.k = hip_loader.load()  # raises if the extension is missing
        if world > self.k.xgmi_max_ranks:
            raise RuntimeError("an xGMI window group spans one node (at most "
                               "%d ranks), got %d" % (self.k.xgmi_max_ranks,
                                                      world))
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device(device)
        self.bytes = int(window_bytes)
        self.timeout = float(timeout)
        self.one_shot_bytes = min(
            int(os.environ.get("SINGA_XGMI_ONE_SHOT_BYTES", str(128 << 10))),
            self.k.xgmi_one_shot_max_bytes())
        self._cursor = 0
        self._imported = []
        self.data = self.flags = self._err_host = None
        torch.cuda.set_device(self.device)
        torch.cuda.current_stream(self.device).synchronize()
        self.data = self.k.xgmi_alloc(self.bytes)
        self.flags = self.k.xgmi_alloc(self.k.xgmi_signal_bytes())
        self._err_host, self._err_dev = self.k.xgmi_error_word()
        mine = (socket.gethostname(), os.getpid(),
                self.k.xgmi_export(self.data), self.k.xgmi_export(self.flags))
        everyone = exchange(mine)
        if len(everyone) != self.world:
            raise RuntimeError("handle exchange returned %d entries for %d "
                               "ranks" % (len(everyone), self.world))
        if any(e[0] != mine[0] for e in everyone):
            raise RuntimeError("xGMI windows need all ranks on one node")
        self.peer_data, self.peer_flags = [], []
        for r, (_, pid, hd, hf) in enumerate(everyone):
            if r == self.rank:
                self.peer_data.append(self.data)
                self.peer_flags.append(self.flags)
                continue
            if pid == mine[1]:
                raise RuntimeError("two ranks in one process")
            d = self.k.xgmi_import(hd)
            self._imported.append(d)
            f = self.k.xgmi_import(hf)
            self._imported.append(f)
            self.peer_data.append(d)
            self.peer_flags.append(f)

    # --------------------------------------------------------------- buffers
    def tensor(self, n, tdtype):
        """A flat tensor of ``n`` elements inside the window.  Its storage is
        padded to whole 16-byte vectors, which the kernels move."""
        item = torch.empty(0, dtype=tdtype).element_size()
        nbytes = (int(n) * item + 15) // 16 * 16
        start = (self._cursor + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        if start + nbytes > self.bytes:
            raise RuntimeError(
                "the xGMI window (%d bytes, SINGA_XGMI_WINDOW_MB) is too "
                "small for another buffer of %d bytes" % (self.bytes, nbytes))
        self._cursor = start + nbytes
        if tdtype == torch.bfloat16:
            raw = torch.as_tensor(_Span(self, self.data + start, nbytes // 2,
                                        "<i2"), device=self.device)
            return raw.view(torch.bfloat16)[:n]
        span = _Span(self, self.data + start, nbytes // item,
                     _TYPESTR[tdtype])
        return torch.as_tensor(span, device=self.device)[:n]

    def owns(self, t):
        return (t.is_cuda and self.data is not None and
                self.data <= t.data_ptr() < self.data + self.bytes)

    def _span(self, t):
        off = t.data_ptr() - self.data
        nbytes = (t.numel() * t.element_size() + 15) // 16 * 16
        assert off % 16 == 0 and off + nbytes <= self.bytes
        return off, nbytes

    # ----------------------------------------------------------- collectives
    def all_reduce(self, t):
        """Sum ``t`` (a tensor from ``tensor``) over the ranks, in place, on
        the current stream."""
        assert self.owns(t) and t.is_contiguous()
        off, nbytes = self._span(t)
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.k.xgmi_allreduce(_CODE[t.dtype], self.peer_data,
                              self.peer_flags, self.rank, self._err_dev, off,
                              nbytes, nbytes <= self.one_shot_bytes,
                              self.timeout, s)

    def all_gather(self, out, t):
        """out = every rank's ``t`` in rank order; both from ``tensor``, a
        whole number of 32-bit words each."""
        assert self.owns(t) and self.owns(out)
        nbytes = t.numel() * t.element_size()
        assert nbytes % 4 == 0, "payload must be whole 32-bit words"
        assert out.numel() * out.element_size() == self.world * nbytes
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.k.xgmi_allgather(self.peer_data, self.peer_flags, self.rank,
                              self._err_dev, t.data_ptr() - self.data,
                              out.data_ptr() - self.data, nbytes,
                              self.timeout, s)

    def all_to_all(self, out, t):
        """``t`` holds one chunk per rank, in rank order; afterwards chunk p
        of ``out`` is the chunk rank p held for this rank.  Both from
        ``tensor``, not overlapping, chunks of whole 32-bit words."""
        assert self.owns(t) and self.owns(out)
        nbytes = t.numel() * t.element_size()
        assert out.numel() * out.element_size() == nbytes
        assert nbytes % (4 * self.world) == 0, \
            "every chunk must be whole 32-bit words"
        lo, hi = t.data_ptr(), out.data_ptr()
        assert lo + nbytes <= hi or hi + nbytes <= lo, "operands overlap"
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.k.xgmi_alltoall(self.peer_data, self.peer_flags, self.rank,
                             self._err_dev, lo - self.data, hi - self.data,
                             nbytes // self.world, self.timeout, s)

    def reduce_scatter(self, t):
        """Sum ``t`` (from ``tensor``; a multiple of 16 * world bytes) over
        the ranks and return the view of this rank's slice of it, which holds
        the sum; the other slices keep this rank's input."""
        assert self.owns(t) and t.is_contiguous()
        off, nbytes = t.data_ptr() - self.data, t.numel() * t.element_size()
        assert off % 16 == 0 and nbytes % (16 * self.world) == 0, \
            "every slice must be whole 16-byte vectors"
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.k.xgmi_reduce_scatter(_CODE[t.dtype], self.peer_data,
                                   self.peer_flags, self.rank, self._err_dev,
                                   off, nbytes, self.timeout, s)
        n = t.numel() // self.world
        return t.view(-1)[self.rank * n:(self.rank + 1) * n]

    def check(self):
        """Raise when a barrier of one of this rank's kernels has timed out
        (the word is in host memory: reading it waits for nothing)."""
        if self._err_host is None:
            return
        v = self.k.xgmi_read_word(self._err_host)
        if v:
            raise RuntimeError(
                "rank %d waited more than %.0f s for rank %d inside a direct "
                "xGMI collective" % (self.rank, self.timeout, v - 1))

    def close(self):
        if self.data is None:
            return
        try:
            torch.cuda.synchronize(self.device)
            for p in self._imported:
                self.k.xgmi_close(p)
            self._imported = []
            self.k.xgmi_free(self.data)
            self.k.xgmi_free(self.flags)
            self.k.xgmi_error_word_free(self._err_host)
        except Exception:       # noqa: BLE001 (best effort at shutdown)
            pass
        self.data = self.flags = self._err_host = None
