"""Devices: the host (``CppCPU``) and one MI355X GPU (``HipGPU``).

A device owns a native graph scheduler (``_rt.Graph``), the streams its ops run
on and the random-number state.  ``Exec`` either runs an op immediately or, in
graph mode, records it together with the blocks it reads / writes.  On the GPU
the recorded graph is replayed through hipGraphs: the replay plan is cut at the
steps that cannot be captured (collectives, host syncs), every capturable range
is captured once into a ``torch.cuda.CUDAGraph`` that shares one private memory
pool with the others, and each later iteration is a handful of graph launches.

Behaviour parity: reference include/singa/core/device.h:57-174 (Exec, RunGraph,
ResetGraph, EnableGraph, Sync, SetRandSeed, SetVerbosity, SetSkipIteration,
PrintTimeProfiling), src/core/device/{device,cpp_cpu,cuda_gpu,platform}.cc.
"""
import os
import time

import torch

from .. import _rt
from . import dtypes

kCpp = 0
kHip = 1  # takes the slot the reference calls kCuda

# Plan steps with these name prefixes are never captured into a hipGraph.
_EAGER_PREFIXES = ("Dist_",)
_SYNC_NAME = "Waiting"


class Device(object):
    """Base device.  ``id() == -1`` is the host."""

    lazy_alloc = True

    def __init__(self, dev_id, lang, torch_device):
        self._id = dev_id
        self._lang = lang
        self.torch_device = torch_device
        self._graph = _rt.Graph()
        self._graph_enabled = False
        self._verbosity = 0
        self._skip_iteration = 5
        self._graph.set_hooks(self._timed_exec, self.SyncNow, self._collect)
        self._node_clock = {}
        self.phase = "eager"  # eager | capture | replay (see HipGPU.RunGraph)

    # -- identity ---------------------------------------------------------
    def id(self):
        return self._id

    def lang(self):
        return self._lang

    def host(self):
        return get_default_device()

    # -- memory -------------------------------------------------------------
    def NewBlock(self, nbytes):
        return _rt.Block(int(nbytes))

    def alloc(self, block, dtype):
        """Attach storage to ``block`` (lazy allocation happens here)."""
        n = max(1, block.size() // dtypes.SizeOf(dtype))
        block.data = self._new_storage(n, dtypes.to_torch(dtype))
        return block.data

    def _new_storage(self, n, tdtype):
        raise NotImplementedError

    def GetAllocatedMem(self):
        return 0

    # -- execution ----------------------------------------------------------
    def graph_enabled(self):
        return self._graph_enabled

    def EnableGraph(self, enable):
        self._graph_enabled = bool(enable)

    def Exec(self, fn, read_blocks, write_blocks, name="no_name"):
        if self._graph_enabled:
            self._graph.AddOperation(fn, list(read_blocks), list(write_blocks),
                                     name)
        else:
            fn()

    def ResetGraph(self):
        self._graph.Reset()
        self._node_clock = {}
        self._graph.set_verbosity(self._verbosity)
        self._graph.set_skip_iteration(self._skip_iteration)

    def RunGraph(self, serial=False):
        prev = self._graph_enabled
        self._graph_enabled = False
        try:
            if serial:
                self._graph.RunInSerial()
            else:
                self._graph.RunGraph()
        finally:
            self._graph_enabled = prev

    def Reset(self):
        self.EnableGraph(False)
        self.SyncNow()
        self.ResetGraph()
        self.SetVerbosity(0)

    def graph(self):
        return self._graph

    def PrintGraph(self):
        print(self._graph.Debug())

    # -- synchronisation ------------------------------------------------------
    def SyncNow(self):
        """Block the host until the device is idle (never recorded)."""
        pass

    def Sync(self):
        """Recorded as a sync node when graph mode is on, like any other op."""
        self.Exec(self.SyncNow, [], [], _SYNC_NAME)

    # -- profiling --------------------------------------------------------------
    def SetVerbosity(self, v):
        self._verbosity = int(v)
        self._graph.set_verbosity(self._verbosity)

    def verbosity(self):
        return self._verbosity

    def SetSkipIteration(self, n):
        self._skip_iteration = int(n)
        self._graph.set_skip_iteration(self._skip_iteration)

    def skip_iteration(self):
        return self._skip_iteration

    def PrintTimeProfiling(self):
        print(self._graph.TimeProfiling(), end="", flush=True)

    def _timed_exec(self, node_id, name):
        t0 = time.perf_counter()
        self._graph.run_node(node_id)
        self._node_clock[node_id] = time.perf_counter() - t0

    def _collect(self, node_id):
        return float(self._node_clock.pop(node_id, 0.0))

    # -- random numbers -----------------------------------------------------------
    def SetRandSeed(self, seed):
        raise NotImplementedError

    @staticmethod
    def EnableLazyAlloc(flag):
        Device.lazy_alloc = bool(flag)


class CppCPU(Device):
    """The host device.  Storage is zero-initialised torch CPU memory."""

    def __init__(self):
        super().__init__(-1, kCpp, torch.device("cpu"))
        from .math_cpu import CpuMath
        self.math = CpuMath(self)
        self.generator = torch.Generator(device="cpu")
        self.generator.manual_seed(int(time.time() * 1e6) & 0x7FFFFFFF)

    def _new_storage(self, n, tdtype):
        return torch.zeros(n, dtype=tdtype)

    def SetRandSeed(self, seed):
        self.generator.manual_seed(int(seed))


class HipGPU(Device):
    """One MI355X.  Memory comes from torch's HIP caching allocator, compute
    runs on torch's current stream and communication on three side streams
    (pack, collective, unpack)."""

    def __init__(self, gpu_id):
        super().__init__(gpu_id, kHip, torch.device("cuda", gpu_id))
        from ..ops import hip_loader
        self.kernels = hip_loader.load()  # raises if the extension is missing
        from .math_hip import HipMath
        self.math = HipMath(self)
        with torch.cuda.device(gpu_id):
            self.stream_comm = torch.cuda.Stream(device=gpu_id)   # "s"
            self.stream_pack = torch.cuda.Stream(device=gpu_id)   # "c1"
            self.stream_unpack = torch.cuda.Stream(device=gpu_id)  # "c2"
            # philox state: [seed, offset]; the offset is advanced on the device
            # so that a captured graph draws fresh numbers on every replay
            self.rng_state = torch.zeros(2, dtype=torch.int64,
                                         device=self.torch_device)
        self.SetRandSeed(int(time.time() * 1e6) & 0x7FFFFFFF)
        self.use_hipgraph = os.environ.get("SINGA_HIPGRAPH", "1") != "0"
        self.warmup_iters = 1
        self._eager_done = 0
        self._replay = None
        self._events = {}

    def _new_storage(self, n, tdtype):
        return torch.empty(n, dtype=tdtype, device=self.torch_device)

    def GetAllocatedMem(self):
        return torch.cuda.memory_allocated(self.torch_device)

    def SetRandSeed(self, seed):
        self.rng_state.copy_(torch.tensor([int(seed), 0], dtype=torch.int64))

    def SyncNow(self):
        if torch.cuda.is_current_stream_capturing():
            return
        torch.cuda.synchronize(self.torch_device)

    def stream_handle(self):
        return torch.cuda.current_stream(self.torch_device).cuda_stream

    # -- graph replay through hipGraphs ---------------------------------------
    def ResetGraph(self):
        super().ResetGraph()
        self._replay = None
        self._eager_done = 0

    def RunGraph(self, serial=False):
        g = self._graph
        if (not self.use_hipgraph) or self._verbosity > 0:
            return super().RunGraph(serial)
        prev = self._graph_enabled
        self._graph_enabled = False
        try:
            with torch.cuda.device(self.torch_device):
                if self._replay is not None and self._replay.serial != serial:
                    self._replay = None
                if self._replay is None:
                    if self._eager_done < self.warmup_iters:
                        # first iteration runs op by op: lazy handles, lazy
                        # parameter moves and allocator warm-up happen here
                        self._eager_done += 1
                        if serial:
                            g.RunInSerial()
                        else:
                            g.RunGraph()
                        return
                    self._replay = _HipGraphReplay(self, serial)
                self._replay.run()
                g.FinishIteration()
        finally:
            self._graph_enabled = prev

    # -- profiling with device events -------------------------------------------
    def _stream_for(self, name):
        if name.startswith("Dist_s"):
            return self.stream_comm
        if name.startswith("Dist_c1"):
            return self.stream_pack
        if name.startswith("Dist_c2"):
            return self.stream_unpack
        return torch.cuda.current_stream(self.torch_device)

    def _timed_exec(self, node_id, name):
        stream = self._stream_for(name)
        ev = self._events.get(node_id)
        if ev is None:
            ev = (torch.cuda.Event(enable_timing=True),
                  torch.cuda.Event(enable_timing=True))
            self._events[node_id] = ev
        ev[0].record(stream)
        self._graph.run_node(node_id)
        ev[1].record(stream)
        self._node_clock[node_id] = ev

    def _collect(self, node_id):
        ev = self._node_clock.pop(node_id, None)
        if ev is None:
            return 0.0
        return ev[0].elapsed_time(ev[1]) / 1000.0


class _HipGraphReplay(object):
    """Captured form of one device graph: alternating hipGraph launches and
    eagerly executed steps, in plan order."""

    def __init__(self, dev, serial):
        self.serial = serial
        self.dev = dev
        g = dev.graph()
        n = g.Prepare(serial)
        plan = g.plan()
        names = g.node_names()
        self.steps = []  # ("graph", CUDAGraph) | ("eager", node_id)
        ranges = []
        i = 0
        while i < n:
            nm = names[plan[i]]
            if nm == _SYNC_NAME:
                i += 1  # host syncs are dropped from the steady state
                ranges.append(("skip", i - 1, i))
            elif nm.startswith(_EAGER_PREFIXES):
                ranges.append(("eager", i, i + 1))
                i += 1
            else:
                j = i
                while (j < n and names[plan[j]] != _SYNC_NAME and
                       not names[plan[j]].startswith(_EAGER_PREFIXES)):
                    j += 1
                ranges.append(("graph", i, j))
                i = j
        pool = torch.cuda.graph_pool_handle()
        cap_stream = torch.cuda.Stream(device=dev.torch_device)
        dev.phase = "capture"
        try:
            for kind, a, b in ranges:
                if kind == "graph":
                    cg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(cg, pool=pool, stream=cap_stream):
                        g.RunRange(a, b)
                    self.steps.append(("graph", cg))
                elif kind == "eager":
                    # binds its tensors (no execution) while phase == capture
                    g.RunRange(a, b)
                    self.steps.append(("eager", plan[a]))
                else:
                    g.RunRange(a, b)  # only to release the blocks it frees
        finally:
            dev.phase = "eager"

    def run(self):
        dev = self.dev
        g = dev.graph()
        dev.phase = "replay"
        try:
            for kind, what in self.steps:
                if kind == "graph":
                    what.replay()
                else:
                    g.run_node(what)
        finally:
            dev.phase = "eager"


# ---------------------------------------------------------------------------
# Platform: device discovery and creation
# ---------------------------------------------------------------------------
_default_device = None
_gpu_cache = {}


def get_default_device():
    global _default_device
    if _default_device is None:
        _default_device = CppCPU()
    return _default_device


class Platform(object):
    """Static helpers mirroring reference include/singa/core/device.h:311-386."""

    @staticmethod
    def GetNumGPUs():
        return torch.cuda.device_count() if torch.cuda.is_available() else 0

    @staticmethod
    def GetGPUIDs():
        return list(range(Platform.GetNumGPUs()))

    @staticmethod
    def GetGPUMemSize(gpu_id=None):
        """(free, total) bytes of one GPU, or a list over all GPUs."""
        if gpu_id is None:
            return [Platform.GetGPUMemSize(i) for i in Platform.GetGPUIDs()]
        free, total = torch.cuda.mem_get_info(gpu_id)
        return (free, total)

    @staticmethod
    def DeviceQuery(gpu_id, verbose=False):
        p = torch.cuda.get_device_properties(gpu_id)
        s = ("Device %d: %s\n  arch: %s\n  compute units: %d\n  memory: %d MiB\n"
             % (gpu_id, p.name, getattr(p, "gcnArchName", "?"),
                p.multi_processor_count, p.total_memory >> 20))
        if verbose:
            s += "  wavefront size: %d\n" % getattr(p, "warp_size", 64)
        return s

    @staticmethod
    def CreateCudaGPUsOn(ids, init_size=0):
        """One device object per physical GPU; asking again for a GPU returns
        the same (reset) object, like the reference's UsedDevice cache."""
        out = []
        for i in ids:
            if not torch.cuda.is_available() or i >= torch.cuda.device_count():
                raise RuntimeError("GPU %d is not available" % i)
            dev = _gpu_cache.get(i)
            if dev is None:
                dev = HipGPU(i)
                _gpu_cache[i] = dev
            else:
                dev.Reset()
            out.append(dev)
        return out

    @staticmethod
    def CreateCudaGPUs(num, init_size=0):
        return Platform.CreateCudaGPUsOn(list(range(num)), init_size)

    @staticmethod
    def GetDefaultDevice():
        return get_default_device()
