"""Disk-backed FIFO whose files the reference can read, and vice versa
(reference pfrl/collections/persistent_collections.py:19-401).

On-disk layout of one queue rooted at ``basedir``::

    basedir/meta.pkl               pickled dict: basedir, maxlen, comm_size (1), ancestor,
                                   timestamp, chunksize, trim (False)
    basedir/rank0/chunk.<g>.data   the pickled items of generation g, back to back
    basedir/rank0/chunk.<g>.idx    one 32-byte record per item, native struct "QQQIi":
                                   generation, byte offset, byte length, CRC-32 of the item's
                                   bytes, status (0)

The log is append-only: ``popleft`` and ``maxlen`` eviction only drop items from memory.  Every
process that opens the queue starts a new generation, and a generation is closed once MORE than
``chunk_size`` bytes have been written to it.  Re-opening loads the newest generations that
together hold at least ``maxlen`` items (every generation if ``maxlen`` is None) and lets the
in-memory FIFO keep the newest ``maxlen`` of them.  An ``ancestor`` directory (another queue's
``basedir``) is read first, following its own ``ancestor`` link while items are still missing,
so a new run can start from an older run's experience without writing into it.
"""
import binascii
import os
import pickle
import struct
from datetime import datetime

from pfrl_amd.collections.random_access_queue import RandomAccessQueue

_RECORD = struct.Struct("QQQIi")     # gen, offset, length, crc32, status


def _chunk_paths(datadir, gen):
    stem = os.path.join(datadir, "chunk.{}".format(gen))
    return stem + ".idx", stem + ".data"


def _read_index(datadir, gen):
    """The index records of one generation; a torn trailing record is ignored."""
    with open(_chunk_paths(datadir, gen)[0], "rb") as f:
        raw = f.read()
    whole = len(raw) - len(raw) % _RECORD.size
    return list(_RECORD.iter_unpack(raw[:whole]))


def _generations(datadir):
    """Consecutive generations 0, 1, ... that have both files."""
    gen = 0
    while all(os.path.exists(p) for p in _chunk_paths(datadir, gen)):
        yield gen
        gen += 1


def _read_items(datadir, gen, unpickle=True):
    """Items of one generation, each checked against its CRC."""
    records = _read_index(datadir, gen)
    with open(_chunk_paths(datadir, gen)[1], "rb") as f:
        blob = memoryview(f.read())
    for _, offset, length, crc, _ in records:
        item = blob[offset:offset + length]
        if binascii.crc32(item) != crc:
            raise AssertionError("CRC mismatch in {} generation {} at offset {}".format(
                datadir, gen, offset))
        yield pickle.loads(item) if unpickle else bytes(item)


def _load_newest(datadir, wanted, sink):
    """Extend ``sink`` with the newest generations of ``datadir`` that cover ``wanted`` items
    (all if None), oldest first.  Returns the next unused generation number."""
    counts = [(gen, len(_read_index(datadir, gen))) for gen in _generations(datadir)]
    first = len(counts)
    missing = wanted
    while first > 0 and (wanted is None or missing > 0):
        first -= 1
        if wanted is not None:
            missing -= counts[first][1]
    for gen, _ in counts[first:]:
        sink.extend(_read_items(datadir, gen))
    return counts[-1][0] + 1 if counts[first:] else 0


class _GenerationWriter:
    """Appends items to one generation's data + index files, flushing both per item so that a
    killed process loses at most the item being written."""

    def __init__(self, datadir, gen, chunk_size):
        assert gen >= 0 and chunk_size > 0
        self.gen = gen
        self.chunk_size = chunk_size
        idx, data = _chunk_paths(datadir, gen)
        self._idx = open(idx, "wb")
        self._data = open(data, "wb")
        self.offset = 0
        self.full = False

    def append(self, item):
        if self.full:
            raise RuntimeError("Already chunk written full")
        blob = pickle.dumps(item)
        self._data.write(blob)
        self._data.flush()
        self._idx.write(_RECORD.pack(self.gen, self.offset, len(blob), binascii.crc32(blob), 0))
        self._idx.flush()
        self.offset += len(blob)
        if self.offset > self.chunk_size:
            self.close()

    def is_full(self):
        return self.full

    def close(self):
        if not self._data.closed:
            self._data.close()
            self._idx.close()
        self.full = True

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter teardown
            pass


class PersistentRandomAccessQueue(object):
    """``RandomAccessQueue`` that logs every appended item under ``basedir``."""

    comm_size = 1          # single writer; the multi-node variant of the reference is private
    comm_rank = 0
    chunk_size = 16 * 128 * 1024 * 1024

    def __init__(self, basedir, maxlen, *, ancestor=None, logger=None):
        assert maxlen is None or maxlen > 0
        self.basedir = basedir
        self.datadir = os.path.join(basedir, "rank0")
        self.logger = logger
        self.buffer = RandomAccessQueue(maxlen=maxlen)
        self.ancestor_meta = None
        if ancestor is not None:
            self.ancestor_meta = self._load_ancestor(ancestor, maxlen)
        self.meta_file = self._meta_file_name(basedir)
        self.meta = self._open_meta(ancestor, maxlen)
        if os.path.exists(self.datadir):
            self.gen = _load_newest(self.datadir, maxlen, self.buffer)
        else:
            self.gen = 0
            os.makedirs(self.datadir, exist_ok=True)
        self.tail = _GenerationWriter(self.datadir, self.gen, self.chunk_size)
        self.gen += 1
        if logger:
            logger.info("Initial buffer size=%d, next gen=%d", len(self.buffer), self.gen)

    # -- meta / lineage ----------------------------------------------------------------------
    @staticmethod
    def _meta_file_name(dirname):
        return os.path.join(dirname, "meta.pkl")

    def _open_meta(self, ancestor, maxlen):
        if os.path.exists(self.meta_file):
            with open(self.meta_file, "rb") as f:
                meta = pickle.load(f)
            assert isinstance(meta, dict)
            assert meta["comm_size"] == self.comm_size, \
                "Reloading same basedir requires same comm.size"
            return meta
        meta = dict(basedir=self.basedir, maxlen=maxlen, comm_size=self.comm_size,
                    ancestor=ancestor,
                    timestamp=datetime.today().strftime("%Y%m%dT%H%M%S.%f"),
                    chunksize=self.chunk_size, trim=False)
        os.makedirs(self.basedir, exist_ok=True)
        with open(self.meta_file, "wb") as f:
            pickle.dump(meta, f)
        return meta

    def _load_ancestor(self, ancestor, wanted):
        with open(self._meta_file_name(ancestor), "rb") as f:
            meta = pickle.load(f)
        assert isinstance(meta, dict)
        if self.logger:
            self.logger.info("Loading buffer data from %s", ancestor)
        # this (single) reader takes every rank directory the ancestor run wrote
        datadirs = [os.path.join(ancestor, "rank{}".format(r)) for r in range(meta["comm_size"])]
        available = sum(len(_read_index(d, g)) for d in datadirs for g in _generations(d))
        if wanted is not None and available < wanted and meta["ancestor"] is not None:
            self._load_ancestor(meta["ancestor"], wanted - available)   # older data goes first
        for datadir in datadirs:
            room = None if wanted is None else wanted - len(self.buffer)
            if room is not None and room <= 0:
                break
            loaded = []
            _load_newest(datadir, room, loaded)
            self.buffer.extend(loaded)
            if self.logger:
                self.logger.info("%d data loaded to buffer (rank=%d)", len(loaded), self.comm_rank)
        return meta

    # -- log ---------------------------------------------------------------------------------
    def _log(self, item):
        if self.tail.is_full():
            self.tail = _GenerationWriter(self.datadir, self.gen, self.chunk_size)
            if self.logger:
                self.logger.info("Chunk rotated. New gen=%d", self.gen)
            self.gen += 1
        self.tail.append(item)

    def close(self):
        self.tail.close()
        self.tail = None

    # -- RandomAccessQueue interface ---------------------------------------------------------
    def append(self, value):
        self._log(value)
        self.buffer.append(value)

    def extend(self, xs):
        xs = list(xs)
        for x in xs:
            self._log(x)
        self.buffer.extend(xs)

    def popleft(self):
        self.buffer.popleft()     # nothing is returned, as in the reference (:308-309)

    def sample(self, n):
        return self.buffer.sample(n)

    def __getitem__(self, i):
        return self.buffer[i]

    def __setitem__(self, i, x):
        raise NotImplementedError()

    def __iter__(self):
        return iter(self.buffer)

    def __len__(self):
        return len(self.buffer)

    def __repr__(self):
        return "PersistentRandomAccessQueue({})".format(str(self.buffer))

    @property
    def maxlen(self):
        return self.meta["maxlen"]
