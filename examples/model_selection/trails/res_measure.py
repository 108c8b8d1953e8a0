"""Resource usage of a phase: wall time, peak resident memory of the process,
CPU time and, on a GPU device, peak device memory -- sampled by a background
thread and appended as one JSON line per measured block (role of the
reference's src/tools/res_measure.py)."""
import json
import os
import threading
import time

try:
    import psutil
except ImportError:         # the numbers that need it are then left out
    psutil = None


class Measure(object):
    """``with Measure("filtering", log_path) as m: ...`` then ``m.result``."""

    def __init__(self, tag, path=None, dev=None, interval=0.05):
        self.tag, self.path, self.dev = tag, path, dev
        self.interval = interval
        self.result = None
        self._stop = threading.Event()
        self._peak_rss = 0

    def _sample(self):
        proc = psutil.Process(os.getpid())
        while not self._stop.wait(self.interval):
            self._peak_rss = max(self._peak_rss, proc.memory_info().rss)

    def __enter__(self):
        self._t0 = time.perf_counter()
        self._cpu0 = time.process_time()
        self._thread = None
        if psutil is not None:
            self._peak_rss = psutil.Process(os.getpid()).memory_info().rss
            self._thread = threading.Thread(target=self._sample, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
        self.result = {
            "tag": self.tag,
            "wall_s": time.perf_counter() - self._t0,
            "cpu_s": time.process_time() - self._cpu0,
        }
        if psutil is not None:
            self.result["peak_rss_mb"] = self._peak_rss / 2.0 ** 20
        if self.dev is not None and hasattr(self.dev, "GetAllocatedMem"):
            try:
                self.result["device_mem_mb"] = \
                    self.dev.GetAllocatedMem() / 2.0 ** 20
            except Exception:       # noqa: BLE001 (host device: no counter)
                pass
        if self.path:
            with open(self.path, "a") as f:
                f.write(json.dumps(self.result) + "\n")
        return False
