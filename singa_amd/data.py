"""Host-side mini-batch iterators that prefetch in a background process.

``ImageBatchIter`` has the call surface of the reference's
python/singa/data.py:64-172 (list file of ``path<delimiter>meta`` lines, a user
``image_transform(path) -> [PIL images]``, ``start() / next() / end()``).  The
producer here differs from the reference in three ways that matter for
feeding a fast accelerator:

* shutdown is an ``Event`` the child polls while it blocks on a bounded
  ``Queue.put`` (the reference flips a flag the child never sees and relies on
  ``terminate``);
* the consumer blocks on ``Queue.get`` instead of sleeping in 100 ms steps;
* decoding of the images of one batch is spread over a small thread pool
  (PIL releases the GIL while decoding/resizing).

``DevicePrefetcher`` stages a batch through pinned host memory and copies it on a side
HIP stream so the H2D transfer of batch i+1 overlaps the step on batch i.
"""
import os
import queue as _queue
import random
from concurrent.futures import ThreadPoolExecutor
from multiprocessing import Event, Process, Queue

import numpy as np


def _parse_list(path, delimiter):
    """-> ([(relative_path, meta or None)], every meta is a label index)."""
    items, all_index = [], True
    with open(path, "r") as fd:
        for line in fd:
            line = line.rstrip("\n")
            if not line.strip():
                continue
            parts = line.split(delimiter)
            if len(parts) < 2:
                all_index = False
                items.append((parts[0].strip(), None))
            else:
                meta = parts[1].strip()
                all_index = all_index and meta.isdigit()
                items.append((parts[0].strip(), meta))
    return items, all_index


class ImageBatchIter(object):
    """Iterate over an image list in mini-batches, forever (wraps around and
    reshuffles at the end of each pass).

    Args:
        img_list_file(str): text file, one ``image_path<delimiter>meta`` per
            line.  When every meta is a non-negative integer the labels come
            back as an int32 array of length batch_size; otherwise as a list of
            strings (or of None when a line has no meta).
        batch_size(int): images per batch, counting augmented copies.
        image_transform: callable(full_path) -> list of PIL images.
        shuffle(bool): shuffle the list before every pass.
        delimiter(str): separator between path and meta.
        image_folder(str): prefix joined to every image path.
        capacity(int): number of batches buffered ahead.
        num_workers(int): decode threads inside the producer process.
    """

    def __init__(self, img_list_file, batch_size, image_transform,
                 shuffle=True, delimiter=" ", image_folder=None, capacity=10,
                 num_workers=2):
        self.img_list_file = img_list_file
        self.batch_size = batch_size
        self.image_transform = image_transform
        self.shuffle = shuffle
        self.delimiter = delimiter
        self.image_folder = image_folder or ""
        self.num_workers = max(1, int(num_workers))
        self._capacity = capacity
        self.queue = Queue(capacity)
        self._stop = Event()
        self.p = None
        self.num_samples = len(_parse_list(img_list_file, delimiter)[0])

    # -- consumer side -------------------------------------------------------
    def start(self):
        self._stop.clear()
        self.p = Process(target=self.run, daemon=True)
        self.p.start()

    def __iter__(self):
        return self

    def __next__(self):
        assert self.p is not None, "call start before next"
        while True:
            try:
                return self.queue.get(timeout=1.0)
            except _queue.Empty:
                if not self.p.is_alive():
                    raise RuntimeError("image producer process died")

    next = __next__

    def end(self):
        """Stop the producer.  Batches still in flight are discarded: the
        child may exit in the middle of writing one, so the pipe is never read
        again (a partial message would block the reader forever); ``start``
        after ``end`` gets a fresh queue."""
        if self.p is None:
            return
        self._stop.set()
        self.p.join(timeout=2.0)
        if self.p.is_alive():
            self.p.terminate()
            self.p.join()
        self.p = None
        self.queue.close()
        self.queue.cancel_join_thread()
        self.queue = Queue(self._capacity)

    # -- producer side -------------------------------------------------------
    def _load(self, rel_path):
        imgs = self.image_transform(os.path.join(self.image_folder, rel_path))
        return [np.asarray(im.convert("RGB"), dtype=np.float32)
                .transpose(2, 0, 1) for im in imgs]

    def run(self):
        items, is_index = _parse_list(self.img_list_file, self.delimiter)
        if self.shuffle:
            random.shuffle(items)
        pos = 0
        pool = ThreadPoolExecutor(self.num_workers)
        carry = []  # (arrays, meta) decoded but not yet placed in a batch
        try:
            while not self._stop.is_set():
                x, y = [], []
                while len(x) < self.batch_size:
                    if not carry:
                        # decode a window of upcoming images in parallel
                        window = []
                        for _ in range(self.num_workers):
                            window.append(items[pos])
                            pos += 1
                            if pos == len(items):
                                pos = 0
                                if self.shuffle:
                                    random.shuffle(items)
                        arrays = pool.map(self._load, [w[0] for w in window])
                        carry = [(a, w[1]) for a, w in zip(arrays, window)]
                    arrs, meta = carry[0]
                    # augmented copies of one image never straddle batches
                    assert len(x) + len(arrs) <= self.batch_size, \
                        "too many images (%d) in a batch (%d)" % (
                            len(x) + len(arrs), self.batch_size)
                    carry.pop(0)
                    x.extend(arrs)
                    y.extend([int(meta) if is_index else meta] * len(arrs))
                batch = (np.stack(x),
                         np.asarray(y, dtype=np.int32) if is_index else y)
                while not self._stop.is_set():
                    try:
                        self.queue.put(batch, timeout=0.1)
                        break
                    except _queue.Full:
                        continue
        finally:
            pool.shutdown(wait=False)
            # do not block process exit on unread batches
            self.queue.cancel_join_thread()


class DevicePrefetcher(object):
    """Wrap any iterator of ``(x, y)`` numpy batches and deliver framework
    tensors already resident on ``dev``.  Each batch is copied through a pinned
    staging buffer on a dedicated copy stream one step ahead of its use, so the
    transfer hides behind the previous training step.  Non-array labels (lists
    of strings / None) pass through unchanged."""

    def __init__(self, it, dev):
        import torch
        self._torch = torch
        self.it = iter(it)
        self.dev = dev
        td = getattr(dev, "torch_device", None)
        self._on_gpu = td is not None and td.type == "cuda"
        self._stream = torch.cuda.Stream(td) if self._on_gpu else None
        self._next = None
        self._preload()

    def _upload(self, a):
        from . import tensor
        if not isinstance(a, np.ndarray):
            return a
        if not self._on_gpu:
            return tensor.from_numpy(a, self.dev)
        torch = self._torch
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        elif a.dtype == np.int64:
            a = a.astype(np.int32)
        t = tensor.Tensor(a.shape, self.dev,
                          {np.dtype(np.float32): tensor.float32,
                           np.dtype(np.float16): tensor.float16,
                           np.dtype(np.int32): tensor.int32}[a.dtype])
        dst = t.data.flat()[:a.size]
        src = torch.from_numpy(np.ascontiguousarray(a)).view(-1).pin_memory()
        dst.copy_(src, non_blocking=True)
        return t

    def _preload(self):
        try:
            x, y = next(self.it)
        except StopIteration:
            self._next = None
            return
        if self._on_gpu:
            cur = self._torch.cuda.current_stream(self.dev.torch_device)
            # storage handed out by the allocator may still be in use by work
            # queued on the compute stream
            self._stream.wait_stream(cur)
            with self._torch.cuda.stream(self._stream):
                self._next = (self._upload(x), self._upload(y))
        else:
            self._next = (self._upload(x), self._upload(y))

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        if self._on_gpu:
            self._torch.cuda.current_stream(self.dev.torch_device) \
                .wait_stream(self._stream)
        out = self._next
        self._preload()
        return out
