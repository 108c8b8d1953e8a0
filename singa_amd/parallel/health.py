"""Failure detection for multi-process training, and fault injection to test
it.

The reference has neither: a failed MPI / NCCL call prints and exits
(include/singa/io/communicator.h:40-67), a rank that dies or hangs leaves the
others blocked in their next collective for good.  Here:

* ``Heartbeat`` -- every rank publishes ``(beats, steps started, steps
  done)`` in a key-value store of its own (a ``torch.distributed.TCPStore``
  hosted by rank 0, apart from the rendezvous store) from a daemon thread and
  watches the other ranks' entries.  A rank whose beat counter stands still
  for ``timeout`` seconds is lost (process dead or frozen).  With
  ``progress_timeout`` set, a job in which no rank has completed a step for
  that long is stalled: the ranks that started fewer steps than this one are
  hung (alive, but not training -- the others sit in a collective waiting for
  them); when all counters are equal the hang is inside a step and every
  rank is named.  Rank 0 going away shows as a store error.
  What happens then is a policy: ``"abort"`` prints the diagnosis and ends
  this process with status 75 (EX_TEMPFAIL: a launcher may restart the job
  from its last checkpoint) -- the only way out of a collective that will
  never complete; ``"flag"`` records it for ``check()`` to raise
  ``RankFailure`` at the next step boundary.
* ``FaultPlan`` -- ``SINGA_FAULT_INJECT="rank=1,step=3,kind=exit"`` makes one
  rank fail at a step boundary: ``exit`` (status 17), ``kill`` (SIGKILL, no
  clean-up), ``hang`` (the training thread sleeps, the process lives), ``raise``
  (an exception in the training loop).

Switches, read by ``Communicator``: ``SINGA_HEARTBEAT=<seconds between
beats>`` (0 / unset: off), ``SINGA_HEARTBEAT_TIMEOUT`` (default 10 beats),
``SINGA_PROGRESS_TIMEOUT`` (0: off), ``SINGA_HEARTBEAT_PORT`` (default
MASTER_PORT + 1), ``SINGA_ON_RANK_FAILURE=abort|flag``.
"""
import datetime
import os
import signal
import sys
import threading
import time

EXIT_RANK_LOST = 75
EXIT_INJECTED = 17


class RankFailure(RuntimeError):
    """Another rank of the job is lost or hung; ``ranks`` maps rank ->
    diagnosis."""

    def __init__(self, ranks):
        self.ranks = dict(ranks)
        super(RankFailure, self).__init__("; ".join(
            "rank %d %s" % kv for kv in sorted(self.ranks.items())))


class Heartbeat(object):

    def __init__(self, rank, world, host, port, interval=1.0, timeout=None,
                 progress_timeout=0.0, on_failure="abort", store=None):
        import torch.distributed as dist
        if on_failure not in ("abort", "flag"):
            raise ValueError("on_failure is 'abort' or 'flag'")
        self.rank, self.world = int(rank), int(world)
        self.interval = float(interval)
        self.timeout = float(timeout) if timeout else 10.0 * self.interval
        self.progress_timeout = float(progress_timeout or 0.0)
        self.on_failure = on_failure
        self.failed = {}
        self.started = 0
        self.steps = 0
        self._progress_at = time.monotonic()
        self._beats = 0
        self._stop = threading.Event()
        self._lock = threading.Lock()
        self._store = store if store is not None else dist.TCPStore(
            host, int(port), self.world, self.rank == 0,
            timeout=datetime.timedelta(seconds=max(self.timeout, 5.0)),
            wait_for_workers=False)
        now = time.monotonic()
        # rank -> [beats, steps started, steps done, time the beats last
        # moved, time the steps done last moved]
        self._seen = {r: [-1, 0, 0, now, now] for r in range(self.world)
                      if r != self.rank}
        self._publish()
        self._thread = threading.Thread(target=self._run, daemon=True,
                                        name="singa-heartbeat")
        self._thread.start()

    # -- what the training thread calls ---------------------------------------
    def begin(self):
        """A training step starts (before its first collective)."""
        self.started = self.steps + 1

    def step(self):
        """A training step is complete."""
        self.steps += 1
        self.started = max(self.started, self.steps)
        self._progress_at = time.monotonic()

    def check(self):
        """Raise ``RankFailure`` when a rank has been found lost or hung."""
        with self._lock:
            if self.failed:
                raise RankFailure(self.failed)

    def close(self):
        self._stop.set()
        if self._thread.is_alive() and \
                threading.current_thread() is not self._thread:
            self._thread.join(timeout=2.0 * self.interval + 1.0)
        try:
            self._store.set("bye/%d" % self.rank, "1")
        except Exception:       # noqa: BLE001 (the store may be gone already)
            pass

    # -- the daemon thread -----------------------------------------------------
    def _publish(self):
        self._beats += 1
        self._store.set("hb/%d" % self.rank, "%d:%d:%d" % (
            self._beats, self.started, self.steps))

    def _left(self, r):
        try:
            return self._store.check(["bye/%d" % r])
        except Exception:       # noqa: BLE001
            return False

    def _run(self):
        while not self._stop.wait(self.interval):
            try:
                self._publish()
                self._watch()
            except Exception as e:      # noqa: BLE001 (store unreachable)
                if self._stop.is_set():
                    return
                host = 0 if self.rank != 0 else self.rank
                self._fail({host: "hosts the heartbeat store, which no "
                                  "longer answers (%s)" % e})
                return

    def _watch(self):
        now = time.monotonic()
        lost = {}
        for r, seen in self._seen.items():
            key = "hb/%d" % r
            if not self._store.check([key]):
                # never seen: the rank may still be starting up
                if now - seen[3] > max(self.timeout, 30.0):
                    lost[r] = "never announced itself"
                continue
            beats, started, steps = (int(v) for v in
                                     self._store.get(key).decode().split(":"))
            if beats != seen[0]:
                seen[0], seen[3] = beats, now
            elif now - seen[3] > self.timeout:
                if self._left(r):
                    continue        # finished and said goodbye
                lost[r] = "sent no heartbeat for %.0f s (process dead or " \
                          "frozen)" % (now - seen[3])
                continue
            if steps != seen[2]:
                seen[4] = now
            seen[1], seen[2] = started, steps
        limit = self.progress_timeout
        if limit > 0 and not lost and now - self._progress_at > limit and \
                all(now - s[4] > limit or self._left(r)
                    for r, s in self._seen.items()):
            active = {r: s for r, s in self._seen.items()
                      if not self._left(r)}
            mine = (self.started, self.steps)
            behind = {r: s for r, s in active.items() if (s[1], s[2]) < mine}
            ahead = {r: s for r, s in active.items() if (s[1], s[2]) > mine}
            stalled = now - self._progress_at
            if behind:
                for r, s in behind.items():
                    lost[r] = "is alive but has been at step %d for %.0f s " \
                              "while rank %d started step %d" % (
                                  s[2], now - s[4], self.rank, self.started)
            elif ahead:
                r, s = max(ahead.items(), key=lambda kv: (kv[1][1], kv[1][2]))
                lost[self.rank] = "(this process) has been at step %d for " \
                                  "%.0f s while rank %d started step %d" % (
                                      self.steps, stalled, r, s[1])
            else:
                for r in active:
                    lost[r] = "and rank %d have all been inside step %d " \
                              "for %.0f s" % (self.rank, self.started, stalled)
        if lost:
            self._fail(lost)

    def _fail(self, ranks):
        with self._lock:
            self.failed.update(ranks)
        if self.on_failure == "abort":
            sys.stderr.write("[rank %d] %s -- leaving with status %d\n" % (
                self.rank, RankFailure(ranks), EXIT_RANK_LOST))
            sys.stderr.flush()
            os._exit(EXIT_RANK_LOST)


class FaultPlan(object):
    """One injected failure: ``rank`` fails in way ``kind`` when it has
    completed ``step`` steps."""

    KINDS = ("exit", "kill", "hang", "raise")

    def __init__(self, rank, step, kind):
        if kind not in self.KINDS:
            raise ValueError("fault kind %r (one of %s)" % (
                kind, ", ".join(self.KINDS)))
        self.rank, self.at, self.kind = int(rank), int(step), kind
        self.fired = False

    @classmethod
    def from_env(cls, text=None):
        text = os.environ.get("SINGA_FAULT_INJECT", "") if text is None \
            else text
        if not text:
            return None
        fields = dict(kv.split("=", 1) for kv in text.split(",") if kv)
        unknown = set(fields) - {"rank", "step", "kind"}
        if unknown or not {"rank", "step"} <= set(fields):
            raise ValueError("SINGA_FAULT_INJECT wants rank=R,step=S"
                             "[,kind=%s], got %r" % ("|".join(cls.KINDS),
                                                     text))
        return cls(fields["rank"], fields["step"], fields.get("kind", "exit"))

    def tick(self, rank, step):
        if self.fired or rank != self.rank or step < self.at:
            return
        self.fired = True
        sys.stderr.write("[rank %d] injected fault '%s' at step %d\n" % (
            rank, self.kind, step))
        sys.stderr.flush()
        if self.kind == "exit":
            os._exit(EXIT_INJECTED)
        if self.kind == "kill":
            os.kill(os.getpid(), signal.SIGKILL)
        if self.kind == "hang":
            while True:
                time.sleep(3600)
        raise RuntimeError("injected fault at step %d" % step)


def heartbeat_from_env(rank, world, host, master_port):
    """The ``Heartbeat`` the environment asks for, or None."""
    interval = float(os.environ.get("SINGA_HEARTBEAT", "0") or 0)
    if interval <= 0 or world <= 1:
        return None
    port = int(os.environ.get("SINGA_HEARTBEAT_PORT", "0") or 0) or \
        int(master_port) + 1
    return Heartbeat(
        rank, world, host, port, interval,
        float(os.environ.get("SINGA_HEARTBEAT_TIMEOUT", "0") or 0) or None,
        float(os.environ.get("SINGA_PROGRESS_TIMEOUT", "0") or 0),
        os.environ.get("SINGA_ON_RANK_FAILURE", "abort"))
