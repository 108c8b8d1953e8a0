"""Record files from Python: (key, value) tuples in the binary record format,
text files of one value per line, and LMDB environments, behind one reader /
writer interface with context-manager support.  The classes wrap the native
implementations of ``singa_amd._rt`` (csrc/runtime/{binfile,textfile,
lmdbfile}.h); the byte formats are those of reference src/io/binfile_*.cc,
textfile_*.cc and lmdb_*.cc.

    with io.open_writer("train.bin") as w:
        w.write("img0001", payload)
    for key, value in io.open_reader("train.bin"):
        ...

The store is chosen by the path: ``*.bin`` binary records, ``*.txt`` / ``*.csv``
text, a directory (or ``*.lmdb`` / ``*.mdb``) an LMDB environment."""
import os

from .. import _rt

kCreate, kAppend = 0, 1


def _kind(path, kind=None):
    if kind:
        return kind
    low = path.lower()
    if low.endswith((".lmdb", ".mdb")) or os.path.isdir(path):
        return "lmdb"
    if low.endswith((".txt", ".csv")):
        return "text"
    return "bin"


class Writer(object):
    """``write(key, value)`` appends a tuple; values are bytes (text files:
    str).  Closing flushes."""

    def __init__(self, path, mode=kCreate, kind=None, **options):
        self.path, self.kind = path, _kind(path, kind)
        append = mode == kAppend
        if self.kind == "lmdb":
            self._w = _rt.LMDBWriter()
            ok = self._w.Open(path, append, options.get("page_size", 4096))
        elif self.kind == "text":
            self._w = _rt.TextFileWriter()
            ok = self._w.Open(path, append)
        else:
            self._w = _rt.BinFileWriter()
            ok = self._w.Open(path, append,
                              options.get("capacity", 10485760))
        if not ok:
            raise IOError("cannot open %s for writing" % path)

    def write(self, key, value):
        if self.kind == "lmdb":
            key = key.encode() if isinstance(key, str) else key
            value = value.encode() if isinstance(value, str) else value
        elif self.kind == "text":
            value = value.decode() if isinstance(value, bytes) else value
        else:
            value = value.encode() if isinstance(value, str) else value
        return self._w.Write(key, value)

    def flush(self):
        self._w.Flush()

    def close(self):
        if self._w is not None:
            self._w.Close()
            self._w = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Reader(object):
    """Iterates over ``(key, value)`` tuples in file order (LMDB: key
    order); ``count()`` is the number of tuples, ``seek_to_first()`` rewinds."""

    def __init__(self, path, kind=None):
        self.path, self.kind = path, _kind(path, kind)
        self._r = {"lmdb": _rt.LMDBReader, "text": _rt.TextFileReader,
                   "bin": _rt.BinFileReader}[self.kind]()
        if not self._r.Open(path):
            raise IOError("cannot open %s for reading" % path)

    def read(self):
        """The next tuple, or None at the end."""
        kv = self._r.Read()
        if kv is None:
            return None
        key, value = kv
        if self.kind == "lmdb":
            key = key.decode(errors="surrogateescape")
        return key, value

    def count(self):
        return self._r.Count()

    def seek_to_first(self):
        self._r.SeekToFirst()

    def close(self):
        if self._r is not None:
            self._r.Close()
            self._r = None

    def __iter__(self):
        while True:
            kv = self.read()
            if kv is None:
                return
            yield kv

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open_writer(path, mode=kCreate, kind=None, **options):
    return Writer(path, mode, kind, **options)


def open_reader(path, kind=None):
    return Reader(path, kind)


BinFileWriter, BinFileReader = _rt.BinFileWriter, _rt.BinFileReader
TextFileWriter, TextFileReader = _rt.TextFileWriter, _rt.TextFileReader
LMDBWriter, LMDBReader = _rt.LMDBWriter, _rt.LMDBReader
