"""Devices: the host (``CppCPU``) and one MI355X GPU (``HipGPU``).

A device owns a native graph scheduler (``_rt.Graph``), the streams its ops run
on and the random-number state.  ``Exec`` either runs an op immediately or, in
graph mode, records it together with the blocks it reads / writes.  On the GPU
the recorded graph is replayed through hipGraphs: the whole plan -- compute
kernels and RCCL collectives on their side streams -- is captured once into a
``torch.cuda.CUDAGraph`` and each later iteration is one graph launch.  Only
when the collectives are not stream ordered (a host backend) is the plan cut
at them into several graphs that share one private memory pool.

Behaviour parity: reference include/singa/core/device.h:57-174 (Exec, RunGraph,
ResetGraph, EnableGraph, Sync, SetRandSeed, SetVerbosity, SetSkipIteration,
PrintTimeProfiling), src/core/device/{device,cpp_cpu,cuda_gpu,platform}.cc.
"""
import contextlib
import os
import sys
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
        self._step_begins()
        try:
            if serial:
                self._graph.RunInSerial()
            else:
                self._graph.RunGraph()
        finally:
            self._graph_enabled = prev
        self._step_done()

    def _step_begins(self):
        comm = getattr(self, "communicator", None)
        if comm is not None:
            comm.on_step_begin()

    def _step_done(self):
        """One iteration of the recorded graph has been issued: the step
        boundary of the communicator's failure detection."""
        comm = getattr(self, "communicator", None)
        if comm is not None:
            comm.on_step()

    def Reset(self):
        self.EnableGraph(False)
        self.SyncNow()
        self.ResetGraph()
        self.SetVerbosity(0)

    # -- more than one recorded graph per device ------------------------------
    # The device records into and replays ONE graph at a time.  A second
    # recorded computation (a model's inference pass next to its training
    # step) owns a graph state of its own and swaps it in while it records or
    # replays.
    def new_graph_state(self):
        g = _rt.Graph()
        g.set_hooks(self._timed_exec, self.SyncNow, self._collect)
        g.set_verbosity(self._verbosity)
        g.set_skip_iteration(self._skip_iteration)
        return {"graph": g, "clock": {}}

    def _get_graph_state(self):
        return {"graph": self._graph, "clock": self._node_clock}

    def _set_graph_state(self, st):
        self._graph, self._node_clock = st["graph"], st["clock"]

    @contextlib.contextmanager
    def graph_state(self, st):
        """Record / replay with the graph state ``st`` (from
        ``new_graph_state``) in place of the device's current one."""
        outer = self._get_graph_state()
        self._set_graph_state(st)
        try:
            yield st
        finally:
            st.update(self._get_graph_state())
            self._set_graph_state(outer)

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
            # SINGA_COMM_STREAM_PRIORITY=-1 puts the collectives on a
            # high-priority HIP stream, so that their few short kernels are
            # scheduled ahead of the convolution workgroups queued by the
            # backward pass.  Off by default: on ROCm 7.0 / MI355X the first
            # cross-stream event wait between the compute stream and a
            # high-priority stream never completed (host blocked in
            # hipEventSynchronize, the step's kernels never ran), with and
            # without an RCCL kernel on that stream.
            self.stream_comm = torch.cuda.Stream(
                device=gpu_id,
                priority=int(os.environ.get("SINGA_COMM_STREAM_PRIORITY",
                                            "0")))                # "s"
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
        # set by the communicator: its collectives are stream ordered (RCCL)
        # and may be captured into the step's hipGraph
        self.collectives_capturable = False
        self.communicator = None   # set with the flag above
        self._watched = []         # (device word, message): see watch_flag
        self._stamps = None   # device-side profile of the captured graph

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

    # -- device-side error words -------------------------------------------------
    def watch_flag(self, word, message):
        """Register a one-element device tensor that kernels set non-zero when
        they detect that their result is invalid (e.g. a bounded grid barrier
        that timed out).  ``check_flags`` raises for it."""
        self._watched.append((word, message))

    def check_flags(self):
        """Called where the host reads device memory (the natural
        synchronisation points of a training loop): raise if a watched word
        has been set since the last check, and clear it."""
        if not self._watched or torch.cuda.is_current_stream_capturing():
            return
        for word, message in self._watched:
            if int(word.item()) != 0:
                word.zero_()
                raise RuntimeError(message)

    def side_streams(self):
        return (self.stream_pack, self.stream_comm, self.stream_unpack)

    def join_side_streams(self):
        """Order the current stream after the three communication streams.
        Inside a hipGraph capture only streams that were forked into the
        capture are joined (a dependency on uncaptured work is not allowed
        to enter a capture)."""
        main = torch.cuda.current_stream(self.torch_device)
        capturing = torch.cuda.is_current_stream_capturing()
        for s in self.side_streams():
            if capturing:
                with torch.cuda.stream(s):
                    if not torch.cuda.is_current_stream_capturing():
                        continue
            main.wait_stream(s)

    # -- graph replay through hipGraphs ---------------------------------------
    def ResetGraph(self):
        super().ResetGraph()
        self._replay = None
        self._eager_done = 0
        self._stamps = None

    def new_graph_state(self):
        st = super().new_graph_state()
        st.update(replay=None, eager_done=0, stamps=None, events={})
        return st

    def _get_graph_state(self):
        st = super()._get_graph_state()
        st.update(replay=self._replay, eager_done=self._eager_done,
                  stamps=self._stamps, events=self._events)
        return st

    def _set_graph_state(self, st):
        super()._set_graph_state(st)
        self._replay, self._eager_done = st["replay"], st["eager_done"]
        self._stamps, self._events = st["stamps"], st["events"]

    def hipgraph_segments(self):
        """Number of hipGraph launches one replayed iteration issues (0 while
        the graph is not captured)."""
        if self._replay is None:
            return 0
        return sum(1 for kind, _ in self._replay.steps if kind == "graph")

    def RunGraph(self, serial=False):
        g = self._graph
        if not self.use_hipgraph:
            return super().RunGraph(serial)
        prev = self._graph_enabled
        self._graph_enabled = False
        self._step_begins()
        try:
            self._run_recorded(g, serial)
        finally:
            self._graph_enabled = prev
        self._step_done()

    def _run_recorded(self, g, serial):
        with torch.cuda.device(self.torch_device):
            if self._replay is not None and (
                    self._replay.serial != serial or
                    self._replay.verbosity != self._verbosity):
                self._replay = None
            if self._replay is None:
                if self._eager_done < self.warmup_iters:
                    # first iteration runs op by op: lazy handles, lazy
                    # parameter moves and allocator warm-up happen here
                    self._eager_done += 1
                    g.set_verbosity(0)
                    try:
                        if serial:
                            g.RunInSerial()
                        else:
                            g.RunGraph()
                    finally:
                        g.set_verbosity(self._verbosity)
                    return
                self._replay = self._capture(serial)
            if self._stamps is not None:
                self._stamps.before_replay(g.iteration(),
                                           self._skip_iteration)
            self._replay.run()
            g.FinishIteration()

    def _capture(self, serial):
        """Capture the recorded step.  With collectives inside the capture
        every rank must end up with the same kind of step: the ranks agree on
        the outcome over the host control group, and if the capture failed
        anywhere, all of them capture again with the collectives between the
        hipGraph segments (a rank replaying a graph with the collective inside
        against one issuing it eagerly would pair different sequences)."""
        self.math.sync_conv_plans(self.communicator)
        if not self.collectives_capturable:
            return _HipGraphReplay(self, serial)
        replay, err = None, None
        try:
            replay = _HipGraphReplay(self, serial)
        except Exception as e:      # noqa: BLE001
            err = e
        comm = self.communicator
        agreed = comm.agree(replay is not None) if comm is not None \
            else replay is not None
        if agreed:
            return replay
        print("capture of the step with its collectives failed%s; "
              "capturing around them" % (
                  " (%s)" % (err,) if err is not None else " on another rank"),
              file=sys.stderr, flush=True)
        del replay
        self.collectives_capturable = False
        if comm is not None:
            comm.capturable = False
        torch.cuda.synchronize(self.torch_device)
        return _HipGraphReplay(self, serial)

    def PrintTimeProfiling(self):
        if self._stamps is not None:
            self._stamps.flush(self._graph)
        super().PrintTimeProfiling()

    # -- profiling ---------------------------------------------------------------
    def _stream_for(self, name):
        return self._streams_of(name)[0]

    def _streams_of(self, name):
        """(first, last) stream a node's work runs on, from its name: a
        ``Dist_<tags>_...`` node names the side streams it crosses (c1 pack,
        s collective, c2 unpack), everything else runs on the compute stream
        (the convention of reference src/core/device/cuda_gpu.cc:134-165)."""
        main = torch.cuda.current_stream(self.torch_device)
        if not name.startswith("Dist_"):
            return main, main
        tags = name[5:].split("_", 1)[0]
        table = (("c1", self.stream_pack), ("c2", self.stream_unpack),
                 ("s", self.stream_comm))
        found = []
        while tags:
            for tag, st in table:
                if tags.startswith(tag):
                    found.append(st)
                    tags = tags[len(tag):]
                    break
            else:
                found = []
                break
        if not found:
            return main, main
        return found[0], found[-1]

    def _timed_exec(self, node_id, name):
        if self._stamps is not None and self.phase == "capture":
            return self._stamps.around(node_id, name)
        stream = self._stream_for(name)
        ev = self._events.get(node_id)
        if ev is None:
            ev = (torch.cuda.Event(enable_timing=True),
                  torch.cuda.Event(enable_timing=True))
            self._events[node_id] = ev
        ev[0].record(stream)
        self._graph.run_node(node_id)
        ev[1].record(self._streams_of(name)[1])
        self._node_clock[node_id] = ev

    def _collect(self, node_id):
        ev = self._node_clock.pop(node_id, None)
        if ev is None:
            return 0.0
        return ev[0].elapsed_time(ev[1]) / 1000.0


class _GraphStamps(object):
    """Per-node timing of a captured step.  Timing events cannot be part of a
    hipGraph, so the capture contains one-thread kernels that read the
    device's constant-rate clock on the stream the profiled node uses and add
    the differences up on the device; the host reads the sums when the
    profile is printed.  Verbosity 1 stamps the forward / backward boundary,
    2 every node, 3 the communication (``Dist``) nodes; the whole step is
    always stamped.  What is timed is the production replay itself."""

    def __init__(self, dev, n_nodes, verbosity, names):
        self.dev = dev
        self.k = dev.kernels
        self.verbosity = verbosity
        self.n = n_nodes
        td = dev.torch_device
        self.slots = torch.zeros(n_nodes + 2, dtype=torch.int64, device=td)
        self.acc = torch.zeros(n_nodes + 1, dtype=torch.float32, device=td)
        self.gate = torch.zeros(1, dtype=torch.int32, device=td)
        self.gate_open = False
        self.sec_per_tick = 1.0 / float(self.k.wall_clock_hz())
        self.first_node = None
        self.fwd_closed = False
        self.names = names

    def _open(self, slot, stream):
        self.k.stamp_open(self.slots.data_ptr(), slot, stream.cuda_stream)

    def _close(self, slot, node, stream):
        self.k.stamp_close(self.slots.data_ptr(), slot, self.acc.data_ptr(),
                           node, self.gate.data_ptr(), self.sec_per_tick,
                           stream.cuda_stream)

    def begin(self):
        main = torch.cuda.current_stream(self.dev.torch_device)
        self._open(self.n, main)             # whole step
        if self.verbosity == 1:
            self._open(self.n + 1, main)     # forward part

    def end(self):
        main = torch.cuda.current_stream(self.dev.torch_device)
        if self.verbosity == 1 and not self.fwd_closed and \
                self.first_node is not None:
            self._close(self.n + 1, self.first_node, main)  # forward only
            self.fwd_closed = True
        self._close(self.n, self.n, main)

    def around(self, node_id, name):
        dev = self.dev
        g = dev.graph()
        main = torch.cuda.current_stream(dev.torch_device)
        if self.first_node is None:
            self.first_node = node_id
        if self.verbosity == 1:
            if not self.fwd_closed and "Backward" in name:
                self._close(self.n + 1, self.first_node, main)
                self.fwd_closed = True
            return g.run_node(node_id)
        if self.verbosity == 3 and "Dist" not in name:
            return g.run_node(node_id)
        first, last = dev._streams_of(name)
        if first is not main:
            first.wait_stream(main)          # the node's work starts here
        self._open(node_id, first)
        g.run_node(node_id)
        self._close(node_id, node_id, last)

    def before_replay(self, iteration, skip):
        if not self.gate_open and iteration >= skip:
            self.gate.fill_(1)
            self.gate_open = True

    def flush(self, g):
        """Move the sums measured so far into the scheduler's per-node
        times."""
        torch.cuda.synchronize(self.dev.torch_device)
        vals = self.acc.cpu().tolist()
        self.acc.zero_()
        for i, v in enumerate(vals[:self.n]):
            if v > 0:
                g.AddNodeTime(i, v)
        g.AddTimeElapsed(vals[self.n])


class _HipGraphReplay(object):
    """Captured form of one device graph.  With stream-ordered collectives
    (RCCL) the whole plan is ONE hipGraph: the communication nodes fork the
    side streams from the capturing stream and ``Dist_wait`` joins them back,
    so the collectives are graph nodes like the kernels around them.  With a
    host-side collective backend the plan is cut at the communication nodes,
    which then run eagerly between the captured segments."""

    # tests set this to make the next capture with collectives inside fail
    fail_next = False

    def __init__(self, dev, serial):
        self.serial = serial
        self.dev = dev
        self.verbosity = dev.verbosity()
        g = dev.graph()
        n = g.Prepare(serial)
        plan = g.plan()
        names = g.node_names()
        eager_dist = not dev.collectives_capturable

        def is_eager(nm):
            return eager_dist and nm.startswith(_EAGER_PREFIXES)

        self.steps = []  # ("graph", CUDAGraph) | ("eager", node_id)
        ranges = []
        i = 0
        while i < n:
            nm = names[plan[i]]
            if nm == _SYNC_NAME:
                i += 1  # host syncs are dropped from the steady state
                ranges.append(("skip", i - 1, i))
            elif is_eager(nm):
                ranges.append(("eager", i, i + 1))
                i += 1
            else:
                j = i
                while (j < n and names[plan[j]] != _SYNC_NAME and
                       not is_eager(names[plan[j]])):
                    j += 1
                ranges.append(("graph", i, j))
                i = j
        # a profile of the captured step needs the step in one piece
        n_graph = sum(1 for r in ranges if r[0] == "graph")
        stamps = None
        if self.verbosity > 0 and n_graph == 1 and \
                not any(r[0] == "eager" for r in ranges):
            stamps = _GraphStamps(dev, len(names), self.verbosity, names)
        dev._stamps = stamps
        if dev.collectives_capturable and _HipGraphReplay.fail_next:
            _HipGraphReplay.fail_next = False
            raise RuntimeError("capture failure requested by a test")
        pool = torch.cuda.graph_pool_handle()
        cap_stream = torch.cuda.Stream(device=dev.torch_device)
        dev.phase = "capture"
        g.set_force_profile(stamps is not None)
        try:
            for kind, a, b in ranges:
                if kind == "graph":
                    cg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(cg, pool=pool, stream=cap_stream,
                                          capture_error_mode="thread_local"):
                        if stamps is not None:
                            stamps.begin()
                        g.RunRange(a, b)
                        # every stream forked into the capture ends in it
                        dev.join_side_streams()
                        if stamps is not None:
                            stamps.end()
                    self.steps.append(("graph", cg))
                elif kind == "eager":
                    # binds its tensors (no execution) while phase == capture
                    g.RunRange(a, b)
                    self.steps.append(("eager", plan[a]))
                else:
                    g.RunRange(a, b)  # only to release the blocks it frees
        finally:
            g.set_force_profile(False)
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
