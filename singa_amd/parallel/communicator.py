"""Collective communication for synchronous data-parallel training.

One process drives one GPU.  Device collectives are RCCL's over xGMI, on a
communicator this module creates itself (``ncclCommInitRank``, the unique id
travelling over the host control group); host-side control traffic (rendezvous,
agreement on fallbacks, the convolution plan table) and CPU ranks use a gloo
group of ``torch.distributed``.  No NCCL process group of ``torch.distributed``
exists, hence no watchdog thread that polls events of the streams a hipGraph
capture runs on.  The interface is the reference's ``Communicator`` (include/singa/io/communicator.h
:76-152, src/io/communicator.cc) so that ``opt.DistOpt`` and user code carry
over, but the data path is laid out for MI355X:

* a fusion bucket is packed by ONE multi-tensor kernel into a flat buffer,
  all-reduced in place, and unpacked by one kernel -- the reference issues a
  ``cudaMemcpyAsync`` per tensor each way and needs a separate receive buffer;
* fp16-on-the-wire converts inside the pack / unpack kernels instead of in
  separate conversion passes;
* pack, collective and unpack run on three side streams chained by events, so
  a bucket's all-reduce overlaps with the rest of the backward pass on the
  compute stream;
* a sparsified exchange is ONE all-gather of fixed-size payloads
  ``[count | indices | values]`` and ONE scatter-add kernel over all ranks'
  payloads; the number of selected entries stays on the device, so the host
  never waits for the GPU;
* every step is recorded on the device graph as a ``Dist_*`` node.  With RCCL
  the nodes are captured into the training step's hipGraph together with the
  compute kernels (side streams forked from and joined back into the capture),
  so a replayed step issues no Python and no launch per bucket.  With a host
  backend (gloo rehearsals) they run eagerly between captured segments.

Ring collectives over xGMI are bound by the per-link rate, not by a switch:
fewer, larger buckets amortise the ring's latency better than many small ones.
"""
import atexit
import ctypes
import datetime
import math
import os
import socket

import torch
import torch.distributed as dist

from ..core import dtypes
from ..core import tensor_core as tc
from . import health


def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class NcclIdHolder(object):
    """Rendezvous token made by the parent process before it starts the ranks
    (reference communicator.h:69-74 wraps an ``ncclUniqueId``).  Here it is
    the TCP address the ranks meet at."""

    def __init__(self, addr="127.0.0.1", port=None):
        self.addr = addr
        self.port = int(port) if port is not None else _free_port()

    @property
    def init_method(self):
        return "tcp://%s:%d" % (self.addr, self.port)


_DEBUG = os.environ.get("SINGA_DIST_DEBUG", "0") == "1"
# diagnosis: bit 0 drains the device before a collective's pipeline starts,
# bit 1 after the compute stream has been joined with the side streams, bit 2
# skips the collective itself.  Results must not depend on any of them.
_STRICT = int(os.environ.get("SINGA_DIST_STRICT", "0"))
# SINGA_DIST_CAPTURE=0 keeps RCCL collectives out of the hipGraph (they then
# run eagerly between captured segments, like with a host backend)
_CAPTURE = os.environ.get("SINGA_DIST_CAPTURE", "1") != "0"


# SINGA_RCCL_DIRECT=0 keeps RCCL out of the picture: device tensors are then
# exchanged through host memory over the gloo control group (slow; a debugging
# aid and the path of rehearsals that put several ranks on one GPU)
_DIRECT = os.environ.get("SINGA_RCCL_DIRECT", "1") != "0"
# SINGA_ALLREDUCE_TWO_PHASE=<elements>: buckets of at least that many elements
# are summed as reduce-scatter + all-gather instead of one all-reduce (0 = off,
# the default: which of the two is faster on a given node is a measurement,
# tools/bench_allreduce.py prints both)
_TWO_PHASE = int(os.environ.get("SINGA_ALLREDUCE_TWO_PHASE", "0"))
# SINGA_XGMI_DIRECT=1: the ranks of a one-node job exchange gradient buckets
# with the kernels of csrc/hip/xgmi.hip, which read every peer's memory over
# that peer's own xGMI link, instead of RCCL's ring (parallel/xgmi.py).
# SINGA_XGMI_WINDOW_MB sizes the per-rank window the buckets live in,
# SINGA_XGMI_TIMEOUT (seconds) bounds a wait for a peer inside a kernel.
_XGMI = os.environ.get("SINGA_XGMI_DIRECT", "0") == "1"


class _UniqueId(ctypes.Structure):
    """``ncclUniqueId``: 128 opaque bytes, passed to RCCL by value."""
    _fields_ = [("internal", ctypes.c_ubyte * 128)]


class _Rccl(object):
    """An RCCL communicator owned by this process, driven through ctypes.

    Rank 0 draws the unique id (``ncclGetUniqueId``), the host control group
    carries it to the other ranks, and every rank joins with
    ``ncclCommInitRank`` on its own GPU (the role of reference
    src/io/communicator.cc:73-114, with the id broadcast over gloo instead of
    MPI).  A collective is one C call that enqueues one kernel on the
    caller's current HIP stream: nothing wraps it in a work object, no thread
    polls its completion, and it is captured into a hipGraph like any other
    kernel launch."""

    _DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4,
               torch.float16: 6, torch.float32: 7, torch.float64: 8,
               torch.bfloat16: 9}
    _SUM = 0

    def __init__(self, device, rank, world, ctl_group):
        path = os.environ.get("SINGA_RCCL_LIB") or os.path.join(
            os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.lib = lib = ctypes.CDLL(path)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclGetUniqueId.restype = ci
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), ci, _UniqueId, ci]
        lib.ncclCommInitRank.restype = ci
        lib.ncclCommDestroy.argtypes = [vp]
        lib.ncclCommDestroy.restype = ci
        lib.ncclAllReduce.argtypes = [vp, vp, sz, ci, ci, vp, vp]
        lib.ncclAllReduce.restype = ci
        lib.ncclAllGather.argtypes = [vp, vp, sz, ci, vp, vp]
        lib.ncclAllGather.restype = ci
        lib.ncclReduceScatter.argtypes = [vp, vp, sz, ci, ci, vp, vp]
        lib.ncclReduceScatter.restype = ci
        lib.ncclGetErrorString.argtypes = [ci]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclCommGetAsyncError.argtypes = [vp, ctypes.POINTER(ci)]
        lib.ncclCommGetAsyncError.restype = ci
        self.comm = None
        self.rank, self.world = rank, world
        uid = _UniqueId()
        blob = [None]
        if rank == 0:
            self._check(lib.ncclGetUniqueId(ctypes.byref(uid)),
                        "ncclGetUniqueId")
            blob[0] = ctypes.string_at(ctypes.byref(uid), 128)
        if world > 1:
            dist.broadcast_object_list(blob, src=0, group=ctl_group)
        assert len(blob[0]) == 128
        ctypes.memmove(ctypes.byref(uid), blob[0], 128)
        torch.cuda.set_device(device)
        torch.cuda.current_stream(device).synchronize()   # HIP context exists
        comm = vp()
        self._check(lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank),
                    "ncclCommInitRank")
        self.comm = comm.value
        if not self.comm:
            raise RuntimeError("ncclCommInitRank returned no communicator")

    def close(self):
        if self.comm:
            comm, self.comm = self.comm, None
            self.lib.ncclCommDestroy(comm)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (
                what, self.lib.ncclGetErrorString(rc).decode()))

    def check_async_error(self):
        """Raise when a collective already enqueued on this communicator has
        failed inside RCCL (a peer went away, a link error): such errors are
        not reported by the call that enqueued the kernel."""
        if not self.comm:
            return
        state = ctypes.c_int(0)
        self._check(self.lib.ncclCommGetAsyncError(self.comm,
                                                   ctypes.byref(state)),
                    "ncclCommGetAsyncError")
        if state.value != 0:
            raise RuntimeError("an RCCL collective failed asynchronously: %s"
                               % self.lib.ncclGetErrorString(
                                   state.value).decode())

    def all_reduce(self, flat):
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        self._check(self.lib.ncclAllReduce(
            flat.data_ptr(), flat.data_ptr(), flat.numel(),
            self._DTYPES[flat.dtype], self._SUM, self.comm, stream),
            "ncclAllReduce")

    def all_gather(self, out, flat):
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        self._check(self.lib.ncclAllGather(
            flat.data_ptr(), out.data_ptr(), flat.numel(),
            self._DTYPES[flat.dtype], self.comm, stream), "ncclAllGather")

    def reduce_scatter(self, out, flat):
        """out (n / world elements) = this rank's slice of the sum of every
        rank's ``flat`` (n elements)."""
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        self._check(self.lib.ncclReduceScatter(
            flat.data_ptr(), out.data_ptr(), out.numel(),
            self._DTYPES[flat.dtype], self._SUM, self.comm, stream),
            "ncclReduceScatter")

    def all_reduce_two_phase(self, flat, world, rank):
        """Sum over ranks as reduce-scatter + all-gather, in place; ``flat``
        must hold a multiple of ``world`` elements.  On a fully connected
        xGMI node both phases send every slice over its own link."""
        n = flat.numel() // world
        mine = flat[rank * n:(rank + 1) * n]
        self.reduce_scatter(mine, flat)
        self.all_gather(flat, mine)


def _env_int(names, default):
    for n in names:
        v = os.environ.get(n)
        if v not in (None, ""):
            return int(v)
    return default


class Communicator(object):
    """``Communicator(buffSize)`` takes rank / world size from the launcher's
    environment (torchrun, mpiexec, srun); ``Communicator(local_rank,
    world_size, holder, buffSize)`` is for ranks started with
    multiprocessing."""

    def __init__(self, *args):
        if len(args) == 1:
            self.max_size = int(args[0])
            self.world_size = _env_int(
                ["WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE",
                 "SLURM_NTASKS"], 1)
            self.global_rank = _env_int(
                ["RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_PROCID"], 0)
            self.local_rank = _env_int(
                ["LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK",
                 "MPI_LOCALRANKID", "SLURM_LOCALID"], self.global_rank)
            meet = (os.environ.get("MASTER_ADDR", "127.0.0.1"),
                    int(os.environ.get("MASTER_PORT", "29500")))
            init_method = "tcp://%s:%d" % meet
        elif len(args) == 4:
            self.local_rank, self.world_size, holder, self.max_size = (
                int(args[0]), int(args[1]), args[2], int(args[3]))
            self.global_rank = self.local_rank
            init_method = holder.init_method
            meet = (holder.addr, holder.port)
        else:
            raise TypeError("Communicator(buffSize) or Communicator("
                            "local_rank, world_size, nccl_id, buffSize)")
        self._gpu = torch.cuda.is_available()
        self._owns_group = False
        # SINGA_DIST_FORCE=1 builds the groups even for a single rank
        # (exercises the RCCL path on a one-GPU machine)
        force = os.environ.get("SINGA_DIST_FORCE", "0") == "1"
        want_host = os.environ.get("SINGA_DIST_BACKEND", "") == "gloo"
        if (self.world_size > 1 or force) and not dist.is_initialized():
            if self._gpu:
                # more ranks than devices only happens in rehearsals with
                # SINGA_DIST_BACKEND=gloo (RCCL refuses two ranks per GPU)
                torch.cuda.set_device(
                    self.local_rank % torch.cuda.device_count())
            # the host control group; device collectives never go through it
            # unless RCCL is unavailable or switched off
            # SINGA_COLLECTIVE_TIMEOUT (seconds): how long a host collective
            # waits for a peer before it raises (gloo's default is 30 min)
            limit = float(os.environ.get("SINGA_COLLECTIVE_TIMEOUT", "0") or 0)
            extra = {"timeout": datetime.timedelta(seconds=limit)} \
                if limit > 0 else {}
            if meet[0] in ("127.0.0.1", "localhost", "::1"):
                # a one-machine job: gloo would look its interface up by the
                # host name, which need not resolve inside a container
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group(
                backend="gloo", init_method=init_method,
                rank=self.global_rank, world_size=self.world_size, **extra)
            self._owns_group = True
            atexit.register(self.close)
        # without a process group a lone rank's all-reduce is the identity
        self._active = dist.is_initialized()
        # fraction of a bucket a threshold-sparsified payload can carry when
        # the entries held back are kept in an accumulation tensor (entries
        # beyond the capacity stay in the accumulator and go out later);
        # without accumulation the capacity is the whole bucket
        self.spars_cap = float(os.environ.get("SINGA_SPARS_CAP", "1.0"))
        self._rccl = None         # this process's RCCL communicator
        if _DIRECT and self._active and self._gpu and not want_host:
            self._rccl = self._join_rccl()
        self._xgmi = None         # this rank's window into the peers' memory
        if _XGMI and self._active and self._gpu:
            self._xgmi = self._open_window()
        # device tensors travel through host memory when there is neither
        self._host_backend = (self._active and self._rccl is None and
                              self._xgmi is None)
        # stream-ordered collectives can be part of a captured hipGraph
        self.capturable = ((self._rccl is not None or self._xgmi is not None)
                           and _CAPTURE and not _DEBUG and not _STRICT)
        probe = os.environ.get("SINGA_DIST_CAPTURE_PROBE", "1")
        if self.capturable and probe != "0" and \
                (self.world_size > 1 or probe == "force"):
            self.capturable = self._probe_capture()
        # failure detection (off unless the environment asks, see health.py)
        self.steps = 0
        self._fault = health.FaultPlan.from_env()
        self._heartbeat = health.heartbeat_from_env(
            self.global_rank, self.world_size, meet[0], meet[1]) \
            if self._active else None
        self._dev = None          # device of the last synchronised tensors
        self._blocks = []         # one tensor per block: what wait() orders against
        self._buffers = {}        # (slot, kind, dtype) -> persistent buffer
        self._slot = 0            # bucket index within the iteration
        self._warmed = False

    # ------------------------------------------------------------------ helpers
    def _join_rccl(self):
        """Create this rank's RCCL communicator; every rank learns whether all
        of them succeeded (one that could not load or initialise RCCL makes
        the whole job exchange through host memory instead of hanging the
        others in their first collective)."""
        device = torch.device("cuda", torch.cuda.current_device())
        rccl, err = None, None
        try:
            rccl = _Rccl(device, self.global_rank, self.world_size, None)
        except (OSError, AttributeError, RuntimeError) as e:
            err = e
        if not self.agree(rccl is not None):
            if err is not None:
                print("[rank %d] no RCCL communicator (%s); device tensors "
                      "are exchanged through host memory" % (
                          self.global_rank, err), flush=True)
            if rccl is not None:
                rccl.close()
            return None
        return rccl

    def _open_window(self):
        """Allocate this rank's xGMI window and map the peers' (see
        parallel/xgmi.py); all ranks use the windows or none does."""
        from . import xgmi
        device = torch.device("cuda", torch.cuda.current_device())

        def exchange(obj):
            if self.world_size <= 1:
                return [obj]
            box = [None] * self.world_size
            dist.all_gather_object(box, obj)
            return box

        win, err = None, None
        try:
            win = xgmi.XgmiWindow(
                self.global_rank, self.world_size, device,
                int(os.environ.get("SINGA_XGMI_WINDOW_MB", "512")) << 20,
                exchange,
                float(os.environ.get("SINGA_XGMI_TIMEOUT", "30")))
        except (OSError, RuntimeError, ImportError) as e:
            err = e
        if not self.agree(win is not None):
            if err is not None:
                print("[rank %d] no xGMI window (%s)" % (self.global_rank,
                                                          err), flush=True)
            if win is not None:
                win.close()
            return None
        return win

    def transport(self):
        """What carries device collectives: "xgmi-direct" (peer-reading
        kernels over IPC windows, RCCL for operands outside the windows),
        "rccl", "host" (staged through host memory over gloo) or "none" (a
        lone rank)."""
        if not self._active:
            return "none"
        if self._xgmi is not None:
            return "xgmi-direct"
        return "rccl" if self._rccl is not None else "host"

    def agree(self, ok):
        """True when ``ok`` holds on every rank.  The exchange runs over the
        host control group, so it never pairs with (or waits behind) a device
        collective."""
        if not self._active or self.world_size <= 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag[0]) == 1)

    def broadcast_object(self, obj, src=0):
        """``obj`` of rank ``src`` on every rank (host control group)."""
        if not self._active or self.world_size <= 1:
            return obj
        box = [obj if self.global_rank == src else None]
        dist.broadcast_object_list(box, src=src)
        return box[0]

    def on_step_begin(self):
        """A training step starts (the device is about to issue a recorded
        iteration, or an eager iteration reached its first collective)."""
        if self._heartbeat is not None:
            self._heartbeat.begin()

    def on_step(self):
        """A training step is complete (called by the device after every
        replayed or eagerly executed iteration): publish the progress, raise
        ``health.RankFailure`` when another rank has been found lost or hung,
        surface asynchronous RCCL errors, and fire an injected fault whose
        step has come."""
        self.steps += 1
        if self._heartbeat is not None:
            self._heartbeat.step()
            self._heartbeat.check()
        if self._rccl is not None:
            self._rccl.check_async_error()
        if self._xgmi is not None:
            self._xgmi.check()
        if self._fault is not None:
            self._fault.tick(self.global_rank, self.steps)

    def close(self):
        """Leave the groups this communicator created (also run at
        interpreter exit: a process that ends with the group's worker threads
        alive aborts in their destructors)."""
        hb, self._heartbeat = getattr(self, "_heartbeat", None), None
        if hb is not None:
            hb.close()
        try:
            if self._gpu and (self._rccl is not None or self._owns_group):
                torch.cuda.synchronize()
            if getattr(self, "_xgmi", None) is not None:
                self._buffers = {}
                if self.world_size > 1 and dist.is_initialized():
                    # a peer's last kernel may still read this rank's window:
                    # every rank drains its device (above) before any unmaps
                    try:
                        dist.monitored_barrier(
                            timeout=datetime.timedelta(seconds=20))
                    except Exception:   # noqa: BLE001 (a rank is gone)
                        pass
                self._xgmi.close()
                self._xgmi = None
            if self._rccl is not None:
                self._rccl.close()
                self._rccl = None
            if self._owns_group and dist.is_initialized():
                dist.destroy_process_group()
            self._owns_group = False
        except Exception:       # noqa: BLE001 (best effort at shutdown)
            pass

    def __del__(self):
        self.close()

    def _buffer(self, dev, slot, kind, n, tdtype):
        """Persistent scratch of this iteration slot: allocated on first use
        (the eager iteration that precedes graph capture) and reused by every
        later iteration, so steady-state steps allocate nothing."""
        key = (slot, kind, tdtype)
        buf = self._buffers.get(key)
        if buf is None or buf.numel() < n or buf.device != dev.torch_device:
            if self._xgmi is not None and kind in self._WINDOW_KINDS:
                # operands of a direct collective live in this rank's window
                buf = self._xgmi.tensor(max(int(n), 1), tdtype)
            else:
                buf = torch.empty(max(int(n), 1), dtype=tdtype,
                                  device=dev.torch_device)
            self._buffers[key] = buf
        return buf[:n]

    _WINDOW_KINDS = ("bucket", "payload", "gathered", "stage")

    def _next_slot(self):
        s = self._slot
        self._slot += 1
        return s

    def _remember(self, dev, tensors):
        if self._dev is not dev:
            self._dev = dev
            dev.communicator = self
        if not self._blocks:
            self.on_step_begin()    # first collective since the last wait()
            if dev.lang() != 0:
                dev.collectives_capturable = self.capturable
        seen = set(id(t.block()) for t in self._blocks)
        for t in tensors:
            if id(t.block()) not in seen:
                self._blocks.append(t)
                seen.add(id(t.block()))

    def _staged(self, t):
        """Device tensors without an RCCL communicator (rehearsals of the
        multi-rank path with several ranks on one GPU, RCCL switched off or
        unavailable) go through host memory explicitly: the device is drained,
        the collective runs on a host copy over gloo, the result is copied
        back on the current stream.  RCCL collectives are stream ordered and
        take the tensor as is."""
        return t.is_cuda and self._rccl is None and not self._windowed(t)

    def _windowed(self, t):
        """True when ``t`` lies in this rank's xGMI window."""
        return self._xgmi is not None and self._xgmi.owns(t)

    def _direct(self, t):
        """True when the collective on ``t`` is an RCCL call."""
        return (self._rccl is not None and t.is_cuda and t.is_contiguous()
                and t.dtype in _Rccl._DTYPES)

    def _probe_capture(self):
        """Capture one small all-reduce on a side stream into a hipGraph,
        replay it and check the sum: the training step relies on exactly this
        (a collective as a graph node, forked from and joined to the capturing
        stream).  Capture and replay are separate phases with an agreement
        over the host group in between, so a rank whose capture failed never
        leaves the others waiting inside a replayed collective.  Any failure,
        on any rank, makes every rank keep its collectives outside the
        captured graphs (the step is then cut into several hipGraphs around
        them)."""
        device = torch.device("cuda", torch.cuda.current_device())
        g = x = None
        ok = True
        try:
            if self._xgmi is not None:
                x = self._xgmi.tensor(4096, torch.float32)
                x.fill_(1.0)
            else:
                x = torch.ones(4096, dtype=torch.float32, device=device)
            self._all_reduce(x)
            torch.cuda.synchronize(device)
            g = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(device=device)
            side = torch.cuda.Stream(device=device)
            with torch.cuda.graph(g, stream=cap,
                                  capture_error_mode="thread_local"):
                x.mul_(1.0)
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    self._all_reduce(x)
                torch.cuda.current_stream(device).wait_stream(side)
                x.mul_(1.0)
        except Exception as e:          # noqa: BLE001 (capability probe)
            print("[rank %d] capture of an all-reduce failed: %s" % (
                self.global_rank, e), flush=True)
            ok = False
        agreed = self.agree(ok)
        if agreed:
            try:
                x.fill_(1.0)
                g.replay()
                g.replay()
                torch.cuda.synchronize(device)
                want = float(self.world_size) ** 2
                ok = abs(float(x[0].item()) - want) <= 1e-3 * want and \
                    abs(float(x[-1].item()) - want) <= 1e-3 * want
            except Exception as e:      # noqa: BLE001
                print("[rank %d] replay of a captured all-reduce failed: %s"
                      % (self.global_rank, e), flush=True)
                ok = False
            agreed = self.agree(ok)
        if not agreed and self.global_rank == 0:
            print("collectives stay outside the captured step "
                  "(SINGA_DIST_CAPTURE_PROBE=0 skips this check)", flush=True)
        return agreed

    @staticmethod
    def _to_host(t):
        torch.cuda.synchronize(t.device)
        return t.cpu()

    def _all_reduce(self, flat):
        if not self._active:
            return
        if _DEBUG:
            # SINGA_DIST_DEBUG=1: log the sequence of collectives (to compare
            # across ranks) with a checksum before and after
            if flat.is_cuda:
                torch.cuda.synchronize(flat.device)
            self._seq = getattr(self, "_seq", 0) + 1
            before = float(flat.float().abs().sum())
        if self._windowed(flat):
            self._xgmi.all_reduce(flat)
        elif self._staged(flat):
            host = self._to_host(flat)
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        elif self._direct(flat):
            if _TWO_PHASE and flat.numel() >= _TWO_PHASE and \
                    flat.numel() % self.world_size == 0:
                self._rccl.all_reduce_two_phase(flat, self.world_size,
                                                self.global_rank)
            else:
                self._rccl.all_reduce(flat)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if _DEBUG:
            print("  [rank %d] all_reduce #%d numel %d |x| %.6g -> %.6g" % (
                self.global_rank, self._seq, flat.numel(), before,
                float(flat.float().abs().sum())), flush=True)

    def _all_gather_into(self, out, flat):
        """out (world_size * n elements) = every rank's ``flat`` (n elements),
        in rank order: one collective."""
        if not self._active:
            out[:flat.numel()].copy_(flat)
        elif self._windowed(flat) and self._windowed(out) and \
                (flat.numel() * flat.element_size()) % 4 == 0:
            self._xgmi.all_gather(out, flat)
        elif self._staged(flat):
            host = self._to_host(flat)
            gathered = torch.empty(out.numel(), dtype=host.dtype)
            dist.all_gather_into_tensor(gathered, host)
            out.copy_(gathered)
        elif self._direct(flat):
            assert out.is_contiguous() and out.dtype == flat.dtype and \
                out.numel() == self.world_size * flat.numel()
            self._rccl.all_gather(out, flat)
        elif flat.is_cuda:
            dist.all_gather_into_tensor(out, flat)
        else:
            # gloo has no flat all-gather on every torch version
            n = flat.numel()
            dist.all_gather([out[r * n:(r + 1) * n]
                             for r in range(self.world_size)], flat)

    @staticmethod
    def _streams(dev):
        """(pack, collective, unpack) streams, or None on the host."""
        if dev.lang() == 0:
            return None
        return dev.stream_pack, dev.stream_comm, dev.stream_unpack

    def _skip_body(self, dev):
        """While the step is captured into hipGraph segments around eagerly
        executed collectives, a collective's body only binds its operands.
        The device's flag decides: it also drops to False when the capture of
        a step with its collectives inside failed on some rank."""
        return dev.phase == "capture" and not (
            self.capturable and getattr(dev, "collectives_capturable", True))

    @staticmethod
    def _operand_views(dev, state, ts):
        """Flat views of a collective's operands.  When collectives run
        eagerly between captured hipGraph segments, the memory the captured
        kernels write is the storage the operands had while the graph was
        being captured: those views are kept and used at every replay (by then
        the scheduler has released the blocks, and asking them again would
        hand out fresh, unrelated storage)."""
        if dev.phase == "replay" and "views" in state:
            return state["views"]
        views = [x.flat()[:x.Size()] for x in ts]
        if dev.phase == "capture":
            state["views"] = views
        return views

    def _run_pipeline(self, dev, stages):
        """Run ``stages`` = [(which_stream, fn), ...] one after another, each
        on its side stream, the first ordered after the compute stream.  The
        last stream's completion is what ``wait`` joins.  Inside a hipGraph
        capture the event waits fork the side streams into the capture."""
        st = self._streams(dev)
        if st is None:
            for _, fn in stages:
                fn()
            return
        names = {"pack": st[0], "comm": st[1], "unpack": st[2]}
        if _STRICT & 1:
            torch.cuda.synchronize(dev.torch_device)
        main = torch.cuda.current_stream(dev.torch_device)
        if _STRICT & 4:
            return          # skip the collective: graph segments + drains only
        prev = main
        for which, fn in stages:
            s = names[which]
            if s is not prev:
                s.wait_stream(prev)
            with torch.cuda.stream(s):
                fn()
            prev = s
        dev.side_streams_used = True

    # ----------------------------------------------------------------- all-reduce
    def synch(self, t):
        """In-place all-reduce (sum) of one tensor.  communicator.cc:304-328."""
        dev = t.device()
        self._remember(dev, [t])
        if not self._active:
            return

        state = {}
        slot = self._next_slot() if self._xgmi is not None else None

        def body(x):
            flat = self._operand_views(dev, state, [x])[0]
            if self._skip_body(dev):
                return
            if slot is not None and flat.dtype in (
                    torch.float32, torch.float16, torch.bfloat16):
                # a direct collective works on the window: copy in, sum, copy
                # back (a fused bucket is packed into the window instead)
                stage = self._buffer(dev, slot, "stage", flat.numel(),
                                     flat.dtype)
                self._run_pipeline(dev, [
                    ("pack", lambda: dev.math.copy(flat, stage)),
                    ("comm", lambda: self._all_reduce(stage)),
                    ("unpack", lambda: dev.math.copy(stage, flat))])
                return
            self._run_pipeline(dev, [("comm", lambda: self._all_reduce(flat))])

        tc._submit(dev, "Dist_s_synch_allreduce", body, [], [t])

    def fusedSynch(self, tensors, send=True):
        """All-reduce a bucket of tensors through one flat buffer.  Calls with
        ``send=False`` only announce tensors of the bucket; the call with
        ``send=True`` carries the whole bucket.  communicator.cc:212-302."""
        self._fused(list(tensors), send, None, "fusedSynch")

    def synchHalf(self, t):
        """All-reduce one fp32 tensor with fp16 on the wire.
        communicator.cc:416-467."""
        assert t.data_type() == dtypes.kFloat32, (
            "This function is only available for input tensor precision 32 "
            "bit, which are converted into 16 bits before transmit")
        self._fused([t], True, torch.float16, "synchHalf")

    def fusedSynchHalf(self, tensors, send=True):
        """Bucketed all-reduce of fp32 tensors with fp16 on the wire.
        communicator.cc:330-414."""
        tensors = list(tensors)
        assert len(tensors) > 0
        assert all(t.data_type() == dtypes.kFloat32 for t in tensors), (
            "This function is only available for input tensor precision 32 "
            "bit, which are converted into 16 bits before transmit")
        self._fused(tensors, send, torch.float16, "fusedSynchHalf")

    def _fused(self, tensors, send, wire, tag):
        assert len(tensors) > 0
        dev = tensors[0].device()
        self._remember(dev, tensors)
        if not send or not self._active:
            return
        # a flat bucket has one element type: tensors of different types
        # (16-bit weights next to float32 batch-norm parameters) travel in
        # one bucket per type
        by_type = {}
        for t in tensors:
            by_type.setdefault(t.data_type(), []).append(t)
        for group in by_type.values():
            self._fused_one_type(dev, group, wire, tag)

    def _fused_one_type(self, dev, tensors, wire, tag):
        slot = self._next_slot()
        total = sum(t.Size() for t in tensors)
        state = {}

        def body(*ts):
            tdt = dtypes.to_torch(ts[0].data_type())
            wdt = wire if wire is not None else tdt
            views = self._operand_views(dev, state, ts)
            key = tuple(v.data_ptr() for v in views)
            if state.get("key") != key:
                # operand addresses are stable under graph replay: the
                # pack table is built once
                state["key"] = key
                state["plan"] = dev.math.pack_plan(views)
            if self._skip_body(dev):
                return
            bucket = self._buffer(dev, slot, "bucket", total, wdt)
            plan = state["plan"]
            self._run_pipeline(dev, [
                ("pack", lambda: dev.math.pack(plan, views, bucket)),
                ("comm", lambda: self._all_reduce(bucket)),
                ("unpack", lambda: dev.math.unpack(plan, bucket, views, 1.0)),
            ])

        tc._submit(dev, "Dist_c1sc2_%s" % tag, body, [], tensors)

    # ------------------------------------------------------------- sparsification
    def sparsification(self, t, *args):
        """``sparsification(t, [accumulation,] threshold, topK)``: exchange only
        the large entries of ``t`` (|x| >= threshold, or the ceil(threshold *
        n) entries of largest magnitude), summing them over the ranks; with an
        accumulation tensor the entries not sent are carried over to the next
        call.  communicator.cc:469-530, 619-807."""
        acc, thr, topk = self._spars_args(args)
        self._sparse([t], acc, thr, topk, "sparsification")

    def fusedSparsification(self, tensors, *args):
        acc, thr, topk = self._spars_args(args)
        self._sparse(list(tensors), acc, thr, topk, "fusedSparsification")

    @staticmethod
    def _spars_args(args):
        if len(args) == 3:
            return args[0], float(args[1]), bool(args[2])
        if len(args) == 2:
            return None, float(args[0]), bool(args[1])
        raise TypeError("expected ([accumulation,] sparsThreshold, topK)")

    def _payload_capacity(self, total, thr, topk, with_acc):
        if topk:
            return min(total, max(1, int(math.ceil(thr * total))))
        if with_acc and self.spars_cap < 1.0:
            return min(total, max(1, int(math.ceil(self.spars_cap * total))))
        return total

    def _sparse(self, tensors, acc, thr, topk, tag):
        assert len(tensors) > 0
        dev = tensors[0].device()
        assert all(t.data_type() == dtypes.kFloat32 for t in tensors), \
            "sparsified exchange supports only fp32"
        self._remember(dev, tensors + ([acc] if acc is not None else []))
        slot = self._next_slot()
        total = sum(t.Size() for t in tensors)
        cap = self._payload_capacity(total, thr, topk, acc is not None)
        words = 1 + 2 * cap
        state = {}
        W = self.world_size if self._active else 1

        def body(*ts):
            if acc is not None:
                a, ts = ts[-1], ts[:-1]
            else:
                a = None
            bound = self._operand_views(
                dev, state, list(ts) + ([a] if a is not None else []))
            views = bound[:len(ts)]
            key = tuple(v.data_ptr() for v in views)
            if state.get("key") != key:
                state["key"] = key
                state["plan"] = dev.math.pack_plan(views)
            if self._skip_body(dev):
                return
            m = dev.math
            plan = state["plan"]
            send = self._buffer(dev, slot, "dense", total, torch.float32)
            payload = self._buffer(dev, slot, "payload", words, torch.int32)
            gathered = self._buffer(dev, slot, "gathered", W * words,
                                    torch.int32)
            if "work" not in state:
                state["work"] = m.sparse_workspace()
            av = bound[-1][:total] if a is not None else None

            def select():
                m.pack(plan, views, send)
                if av is not None:
                    m.axpy(1.0, av, send)          # add what was held back
                m.sparse_select(send, thr, topk, payload, cap, state["work"])
                if av is not None:
                    # residual = everything that is not sent
                    m.copy(send, av)
                    m.sparse_accumulate(payload, 1, cap, av, -1.0)

            def exchange():
                self._all_gather_into(gathered, payload)

            def combine():
                m.fill(send, 0.0)
                m.sparse_accumulate(gathered, W, cap, send, 1.0)
                m.unpack(plan, send, views, 1.0)

            self._run_pipeline(dev, [("pack", select), ("comm", exchange),
                                     ("unpack", combine)])

        ops = list(tensors) + ([acc] if acc is not None else [])
        tc._submit(dev, "Dist_c1c2_%s" % tag, body, [], ops)

    # ------------------------------------------------------------------------ wait
    def wait(self):
        """Order the compute stream after everything the communicator has in
        flight.  communicator.cc:169-186."""
        if self._dev is None:
            return
        dev = self._dev
        tensors = self._blocks
        self._blocks = []
        self._slot = 0
        if not self._active:
            return          # a lone rank put nothing in flight
        if not dev.graph_enabled():
            self.on_step()  # eager mode: wait() ends the iteration

        def join(*_):
            if self._skip_body(dev):
                return
            st = self._streams(dev)
            if st is None:
                return
            dev.join_side_streams()
            if _STRICT & 2:
                torch.cuda.synchronize(dev.torch_device)

        # submitted like every other operation: the recorded node keeps a
        # tensor copy per use, so the scheduler's count of graph references
        # matches the blocks' reference counts (a bare block list would make a
        # tensor the user still holds, e.g. a residual accumulator, look like
        # a graph-internal buffer that may be released after the wait)
        tc._submit(dev, "Dist_wait", join, tensors, tensors)
