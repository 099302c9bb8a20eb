"""safetensors reader / writer of the MI355X merge backend.

Replaces ``/root/reference/sd_mecha/streaming.py`` for the hot path:

* ``InSafetensorsDict`` (reference :85-178: lock + buffered ``readinto`` + ``clone``) becomes a read-only
  **mmap** of the file: ``raw(key)`` returns a zero-copy byte view that the scheduler copies straight into a
  pinned staging buffer -- no per-file lock, no intermediate copies.  ``__getitem__`` keeps the reference
  behaviour (a torch tensor copy) for user-defined methods.
* ``OutSafetensorsDict`` (reference :204-350) keeps the file format decisions that are observable:
  8-byte little-endian header length, JSON header with ``__metadata__.mecha_recipe``, header space reserved
  up front from a worst-case header and padded with spaces at ``close()`` (reference :256-275, :346-350),
  tensors appended in completion order.  Writes are positional (``os.pwrite``) so a sharded job can fill
  one file from several processes (``reserve()`` hands out disjoint regions).

Same error types as the reference: a missing key raises ``StateDictKeyError``; non-``.safetensors`` paths
are rejected (reference :87-88).
"""
from __future__ import annotations

import ctypes
import json
import mmap
import os
import pathlib
import struct
import threading
from collections import OrderedDict
from typing import Dict, Iterable, Iterator, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from .merge_methods import StateDictKeyError

# safetensors dtype tag -> (torch dtype, bytes per element); reference streaming.py:395-425
DTYPE_MAPPING = {
    "F64": (torch.float64, 8), "I64": (torch.int64, 8), "F32": (torch.float32, 4), "I32": (torch.int32, 4),
    "F16": (torch.float16, 2), "BF16": (torch.bfloat16, 2), "I16": (torch.int16, 2), "I8": (torch.int8, 1),
    "U8": (torch.uint8, 1), "BOOL": (torch.bool, 1),
}
DTYPE_REVERSE_MAPPING = {v[0]: (k, v[1]) for k, v in DTYPE_MAPPING.items()}


class TensorMetadata:
    __slots__ = ("shape", "dtype", "start", "end")

    def __init__(self, shape, dtype, start=0, end=0):
        self.shape = tuple(map(int, shape))
        self.dtype = dtype
        self.start, self.end = int(start), int(end)

    @classmethod
    def from_header(cls, shape, dtype, start, end) -> "TensorMetadata":
        """An entry of a parsed safetensors header: JSON integers already, no conversions (a LoRA job opens 17 files with
        60 000 entries between them)."""
        m = object.__new__(cls)
        m.shape, m.dtype, m.start, m.end = tuple(shape), dtype, start, end
        return m

    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n

    def nbytes(self) -> int:
        return self.numel() * DTYPE_REVERSE_MAPPING[self.dtype][1]

    def __repr__(self):
        return f"TensorMetadata(shape={self.shape}, dtype={self.dtype})"


def tensor_bytes(t: torch.Tensor) -> bytes:
    """Raw little-endian bytes of a CPU tensor (0-d and empty tensors included)."""
    t = t.detach().cpu().contiguous()
    return t.reshape(-1).view(torch.uint8).numpy().tobytes() if t.numel() else b""


def read_header(path: pathlib.Path) -> Tuple[int, Dict[str, dict]]:
    """8-byte LE length + JSON (reference :141-149).  Returns (header_size, header sorted by data offset)."""
    with open(path, "rb") as f:
        size_bytes = f.read(8)
        if len(size_bytes) != 8:
            raise ValueError(f"{path}: not a safetensors file (too short)")
        (header_size,) = struct.unpack("<Q", size_bytes)
        header = json.loads(f.read(header_size).decode("utf-8").strip())
    ordered = OrderedDict(sorted(header.items(), key=lambda kv: kv[1].get("data_offsets", [0])[0] if isinstance(kv[1], dict) else 0))
    return header_size, ordered


_DTYPE_BY_CODE = (torch.float64, torch.int64, torch.float32, torch.int32, torch.float16, torch.bfloat16, torch.int16, torch.int8,
                  torch.uint8, torch.bool)          # the tag order of mecha_host_parse_safetensors_header


def read_header_tables(path: pathlib.Path):
    """(header_size, {name: TensorMetadata} in data-offset order, the file's ``__metadata__`` dict) through the library's header
    scanner (csrc/hostio.cpp: a LoRA job opens 17 files with 60 000 entries between them -- 0.25 s of ``json.loads`` and dict
    building in one interpreter thread, 0.04 s this way); None when the library is not built or does not read this header
    (an exotic dtype tag, > 8 dimensions): the caller then takes the JSON path."""
    try:
        from . import _lib
        L = _lib.lib()
    except Exception:
        return None
    with open(path, "rb") as f:
        size_bytes = f.read(8)
        if len(size_bytes) != 8:
            raise ValueError(f"{path}: not a safetensors file (too short)")
        (header_size,) = struct.unpack("<Q", size_bytes)
        text = f.read(header_size)
    cap = len(text) // 48 + 8                                   # an entry takes more than 48 bytes of text
    names_buf = ctypes.create_string_buffer(max(len(text), 1))
    used = ctypes.c_int64(0)
    codes, ndim = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.int32)
    shapes, offsets, meta_span = np.empty((cap, 8), dtype=np.int64), np.empty((cap, 2), dtype=np.int64), np.zeros(2, dtype=np.int64)
    n = L.mecha_host_parse_safetensors_header(text, len(text), cap, names_buf, ctypes.byref(used), codes.ctypes.data, ndim.ctypes.data,
                                              shapes.ctypes.data, offsets.ctypes.data, meta_span.ctypes.data)
    if n < 0:
        return None
    names = names_buf.raw[:used.value - 1].decode("utf-8").split("\0") if n else []     # already in data-offset order (reference :141-149)
    file_meta = json.loads(text[meta_span[0]:meta_span[0] + meta_span[1]]) if meta_span[1] else {}
    return header_size, _HeaderTable(names, codes[:n], ndim[:n], shapes[:n], offsets[:n]), (file_meta if isinstance(file_meta, dict) else {})


class _HeaderTable(Mapping):
    """{name: TensorMetadata} over the scanner's tables, in data-offset order; an entry becomes Python objects when somebody asks for
    it (membership tests and key listings never do: the 16 LoRA files of a LoRA job are only asked "do you hold <prefix>.lora_up.weight"
    until the loader thread reads them)."""

    def __init__(self, names, codes, ndim, shapes, offsets):
        self.names = names
        self._index = dict(zip(names, range(len(names))))
        self._codes, self._ndim, self._shapes, self._offsets = codes, ndim, shapes, offsets
        self._made: Dict[str, TensorMetadata] = {}

    def __contains__(self, key):
        return key in self._index

    def __len__(self):
        return len(self._index)

    def __iter__(self):
        return iter(self._index)

    def keys(self):
        return self._index.keys()

    def __getitem__(self, key):
        m = self._made.get(key)
        if m is None:
            i = self._index[key]
            start, end = self._offsets[i].tolist()
            m = self._made[key] = TensorMetadata.from_header(self._shapes[i, :self._ndim[i]].tolist(), _DTYPE_BY_CODE[self._codes[i]], start, end)
        return m

    def items(self):
        if len(self._made) < len(self._index):                  # everything at once: one conversion per table
            nd, sh, of, cd = self._ndim.tolist(), self._shapes.tolist(), self._offsets.tolist(), self._codes.tolist()
            from_header, made = TensorMetadata.from_header, self._made
            for k, i in self._index.items():
                if k not in made:
                    made[k] = from_header(sh[i][:nd[i]], _DTYPE_BY_CODE[cd[i]], of[i][0], of[i][1])
        return ((k, self._made[k]) for k in self._index)


class InSafetensorsDict(Mapping):
    """Read side (protocol of reference :59-82: keys / metadata / __getitem__ / close)."""

    def __init__(self, file_path, buffer_size: int = 0):
        file_path = pathlib.Path(file_path)
        if file_path.suffix != ".safetensors":
            raise ValueError(f"Model type not supported: {file_path} (only safetensors are supported)")
        self.file_path = file_path
        tables = read_header_tables(file_path)
        if tables is not None:
            self.header_size, self._meta, self._file_metadata = tables
        else:
            self.header_size, header = read_header(file_path)
            from_header, dtypes = TensorMetadata.from_header, DTYPE_MAPPING
            self._meta = {k: from_header(v["shape"], dtypes[v["dtype"]][0], *v["data_offsets"]) for k, v in header.items() if k != "__metadata__"}
            self._file_metadata = dict(header.get("__metadata__", {}) or {})
        self._meta_as_stored = self._meta                       # apply_aliases replaces _meta, never mutates it
        self.data_start = 8 + self.header_size
        self._file = open(file_path, "rb")
        size = os.fstat(self._file.fileno()).st_size
        self._mmap = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ) if size > 0 else None
        self._view = memoryview(self._mmap) if self._mmap is not None else memoryview(b"")

    def apply_aliases(self, resolve) -> None:
        """Re-key the tensors by their canonical names (``resolve``: alias -> key of the model config, identity for
        anything else).  The reference looks an output key up under its own name first, then under its aliases
        (merging.py:394-403): a file that holds both keeps the canonical entry."""
        renamed = {k: v for k, v in self._meta.items() if resolve(k) == k}
        for k, v in self._meta.items():
            canon = resolve(k)
            if canon != k and canon not in renamed:
                renamed[canon] = v
        self._meta = renamed

    # ---- mapping protocol
    def __contains__(self, key):
        return key in self._meta

    def __len__(self):
        return len(self._meta)

    def __iter__(self) -> Iterator[str]:
        return iter(self._meta)

    def keys(self) -> Iterable[str]:
        return self._meta.keys()

    def metadata(self) -> Iterable[Tuple[str, TensorMetadata]]:
        return self._meta.items()

    def file_metadata(self) -> dict:
        return dict(self._file_metadata)

    def meta(self, key: str) -> TensorMetadata:
        try:
            return self._meta[key]
        except KeyError as e:
            raise StateDictKeyError(key) from e

    def raw(self, key: str, start_elem: int = 0, n_elems: Optional[int] = None) -> memoryview:
        """Zero-copy bytes of (a row range of) a tensor inside the mmap."""
        m = self.meta(key)
        esz = DTYPE_REVERSE_MAPPING[m.dtype][1]
        lo = self.data_start + m.start + start_elem * esz
        hi = self.data_start + m.end if n_elems is None else lo + n_elems * esz
        return self._view[lo:hi]

    def fileno(self) -> int:
        return self._file.fileno()

    def file_offset(self, key: str) -> int:
        """Absolute file offset of a tensor's first byte (for the scheduler's pread scatter lists)."""
        return self.data_start + self.meta(key).start

    def __getitem__(self, key: str) -> torch.Tensor:
        m = self.meta(key)
        if m.end == m.start:
            return torch.tensor([], dtype=m.dtype).reshape(m.shape)          # reference :169-170
        buf = bytearray(self.raw(key))                                       # one copy, like the reference's clone()
        return torch.frombuffer(buf, dtype=m.dtype).view(m.shape)

    def close(self):
        if getattr(self, "_view", None) is not None:
            self._view.release()
            self._view = None
        if getattr(self, "_mmap", None) is not None:
            self._mmap.close()
            self._mmap = None
        if getattr(self, "_file", None) is not None:
            self._file.close()
            self._file = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedSafetensors(InSafetensorsDict):
    """A checkpoint whose bytes already sit in PINNED host memory ("a second 'from pinned RAM, no files' mode
    isolates link from filesystem").  ``preload_pinned(path)`` reads the file once into one pinned tensor; a ``merge()`` that
    finds every input of a fused plan preloaded skips the staging copies and enqueues the H2D copies straight from the image
    (one hipMemcpyAsync per key and model).  Everything else behaves like the file-backed reader."""

    def __init__(self, file_path, threads: int = 16):
        super().__init__(file_path)
        from . import _lib
        size = os.fstat(self._file.fileno()).st_size
        self.image = torch.empty(max(size, 1), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        piece = 64 << 20
        n = (size + piece - 1) // piece
        if n:
            fds = np.full(n, self._file.fileno(), dtype=np.int32)
            off = np.arange(n, dtype=np.int64) * piece
            dst = np.uint64(self.image.data_ptr()) + off.astype(np.uint64)
            nb = np.minimum(piece, size - off).astype(np.uint64)
            _lib.check(_lib.lib().mecha_host_pread_many(fds.ctypes.data, dst.ctypes.data, nb.ctypes.data, off.ctypes.data, n, threads),
                       "mecha_host_pread_many")

    def host_address(self, key: str) -> int:
        """Address of the tensor's first byte inside the pinned image."""
        return self.image.data_ptr() + self.file_offset(key)

    def close(self):
        pass                                        # kept open for the next merge(); release() drops it


_PRELOADED: Dict[str, PinnedSafetensors] = {}


def preload_pinned(path) -> PinnedSafetensors:
    """Load a checkpoint into pinned host memory; merge() calls that name this path afterwards read it from there."""
    key = str(pathlib.Path(path).resolve())
    if key not in _PRELOADED:
        _PRELOADED[key] = PinnedSafetensors(path)
    return _PRELOADED[key]


def release_pinned(path=None) -> None:
    for key in [str(pathlib.Path(path).resolve())] if path is not None else list(_PRELOADED):
        r = _PRELOADED.pop(key, None)
        if r is not None:
            InSafetensorsDict.close(r)


def open_reader(path) -> InSafetensorsDict:
    """The reader merge() uses for an input: the preloaded pinned image if there is one, else an mmap of the file."""
    hit = _PRELOADED.get(str(pathlib.Path(path).resolve()))
    if hit is not None:
        hit._meta = hit._meta_as_stored                         # undo a previous merge's alias renaming
        return hit
    return InSafetensorsDict(path)


def _worst_case_header_size(header: Mapping[str, TensorMetadata], metadata: dict) -> int:
    """reference :256-275: size of the header if tensors were laid out largest first."""
    dummy = OrderedDict()
    dummy["__metadata__"] = metadata
    off = 0
    for k, m in sorted(header.items(), key=lambda kv: kv[1].nbytes(), reverse=True):
        dummy[k] = {"dtype": DTYPE_REVERSE_MAPPING[m.dtype][0], "shape": list(m.shape), "data_offsets": [off, off + m.nbytes()]}
        off += m.nbytes()
    return len(json.dumps(dummy, separators=(",", ":")).encode("utf-8"))


class OutSafetensorsDict:
    """Write side.  ``put(key, meta, data_bytes)`` appends; ``close()`` writes the header.

    One instance per writing process.  For a job sharded over ranks every rank opens the same file with
    ``create=(rank == 0)`` and a disjoint ``[region_start, region_end)`` of the data section; rank 0 merges
    the sub-headers (``close(extra_entries=...)``)."""

    def __init__(self, file_path, header: Mapping[str, TensorMetadata], mecha_recipe: Optional[str], *,
                 create: bool = True, region_start: int = 0, pages_reserved: bool = False, region_bytes: Optional[int] = None):
        self.file_path = pathlib.Path(file_path)
        self.metadata = {"mecha_recipe": mecha_recipe} if mecha_recipe is not None else {}
        self.max_header_size = _worst_case_header_size(header, self.metadata)
        self.data_start = 8 + self.max_header_size
        flags = os.O_RDWR | (os.O_CREAT | os.O_TRUNC if create else 0)
        self._fd = os.open(self.file_path, flags, 0o644)
        self._prealloc = None
        self._pages_reserved = pages_reserved and not create     # create=False: the creating rank says whether it reserved the pages
        self._prealloc_error: Optional[OSError] = None
        if create or region_bytes is not None:
            # Full length up front (every key of the header merged): put_many stores through a mapping of the file.
            # Allocating the pages first (in the background, while the first wave is staged) makes those stores 3x
            # faster on tmpfs / the page cache than faulting fresh pages in (tools/write_bench.cpp: 12 vs 4.4 GB/s).
            # Sharded jobs (`region_bytes`): the creating rank sets the length, then EVERY rank reserves the pages of its own
            # region -- one rank reserving the whole file while the others wait costs the job the fallocate of N regions.
            size = self.data_start + sum(m.nbytes() for m in header.values())
            if create:
                os.ftruncate(self._fd, size)
            if region_bytes is None:
                lo, n = 0, size                                                 # one writer: the whole file
            elif create:
                lo, n = 0, self.data_start + region_start + region_bytes        # the creating rank: header + its region
            else:
                lo, n = self.data_start + region_start, region_bytes            # any other rank: its region

            def prealloc(fd=self._fd):
                try:
                    if n > 0:
                        os.posix_fallocate(fd, lo, n)
                    self._pages_reserved = True
                except OSError as e:
                    # no fallocate here, or not enough room (ENOSPC on a size-limited tmpfs): the file stays sparse, and a
                    # store through a mapping of a page the file system cannot back kills the process with SIGBUS.  put_many
                    # then writes with pwrite, which reports the error instead.
                    self._prealloc_error = e
            self._prealloc = threading.Thread(target=prealloc, daemon=True)
            self._prealloc.start()
        self._lock = threading.Lock()
        self._cursor = region_start                       # next free byte of this writer's region (data-section relative)
        self.entries: Dict[str, dict] = {}

    def put(self, key: str, meta: TensorMetadata, data) -> None:
        """Append one tensor (any bytes-like of exactly meta.nbytes())."""
        view = memoryview(data).cast("B") if not isinstance(data, memoryview) else data.cast("B")
        if len(view) != meta.nbytes():
            raise ValueError(f"{key}: got {len(view)} bytes, expected {meta.nbytes()}")
        with self._lock:
            off = self._cursor
            self._cursor += len(view)
        written = 0
        while written < len(view):
            written += os.pwrite(self._fd, view[written:], self.data_start + off + written)
        self.entries[key] = {"dtype": DTYPE_REVERSE_MAPPING[meta.dtype][0], "shape": list(meta.shape), "data_offsets": [off, off + len(view)]}

    def put_many(self, keys: Sequence[str], metas: Sequence[TensorMetadata], src_ptrs, threads: int) -> None:
        """Append many tensors whose bytes already sit in host memory (a drained staging arena): file offsets are
        assigned in order under the lock, the bytes go out through libmecha_hip's host thread team."""
        if not keys:
            return
        from . import _lib
        self._join_prealloc()
        sizes = np.fromiter((m.nbytes() for m in metas), dtype=np.uint64, count=len(keys))
        with self._lock:
            first = self._cursor
            self._cursor += int(sizes.sum())
        rel = first + np.concatenate(([0], np.cumsum(sizes[:-1], dtype=np.uint64))).astype(np.int64)
        srcs = np.ascontiguousarray(src_ptrs, dtype=np.uint64)
        offs = np.ascontiguousarray(rel + self.data_start, dtype=np.int64)
        args = (self._fd, srcs.ctypes.data, sizes.ctypes.data, offs.ctypes.data, len(keys), threads)
        # stores through a shared mapping only where the pages are known to exist; otherwise (and when the mapping itself
        # fails, e.g. a file system without shared mappings) positional writes, whose ENOSPC / EIO come back as errors
        if not self._pages_reserved or _lib.lib().mecha_host_mmap_write_many(*args) != 0:
            _lib.check(_lib.lib().mecha_host_pwrite_many(*args), "mecha_host_pwrite_many")
        for key, meta, off, n in zip(keys, metas, rel.tolist(), sizes.tolist()):
            self.entries[key] = {"dtype": DTYPE_REVERSE_MAPPING[meta.dtype][0], "shape": list(meta.shape), "data_offsets": [off, off + n]}

    def reserve(self, key: str, meta: TensorMetadata) -> int:
        """Claim the file region of one tensor that arrives in pieces (a key larger than the staging window, split by
        rows): returns its offset inside the data section; fill it with write_at()."""
        n = meta.nbytes()
        with self._lock:
            off = self._cursor
            self._cursor += n
        self.entries[key] = {"dtype": DTYPE_REVERSE_MAPPING[meta.dtype][0], "shape": list(meta.shape), "data_offsets": [off, off + n]}
        return off

    def write_at(self, rel_offset: int, src_ptr: int, nbytes: int, threads: int) -> None:
        """`nbytes` from host memory at `src_ptr` into the data section at `rel_offset` (a piece of a reserved tensor)."""
        if nbytes <= 0:
            return
        from . import _lib
        self._join_prealloc()
        srcs = np.asarray([src_ptr], dtype=np.uint64)
        sizes = np.asarray([nbytes], dtype=np.uint64)
        offs = np.asarray([self.data_start + rel_offset], dtype=np.int64)
        args = (self._fd, srcs.ctypes.data, sizes.ctypes.data, offs.ctypes.data, 1, threads)
        if not self._pages_reserved or _lib.lib().mecha_host_mmap_write_many(*args) != 0:
            _lib.check(_lib.lib().mecha_host_pwrite_many(*args), "mecha_host_pwrite_many")

    def __setitem__(self, key: str, tensor: torch.Tensor) -> None:
        """WriteOnlyMapping protocol of the reference (typing_.py:12-20): tensor -> bytes -> file."""
        self.put(key, TensorMetadata(tensor.shape, tensor.dtype), tensor_bytes(tensor))

    @property
    def bytes_written(self) -> int:
        return self._cursor

    def wait_reserved(self) -> bool:
        """Wait for the background page reservation; False when the file stayed sparse."""
        self._join_prealloc()
        return self._pages_reserved

    def _join_prealloc(self):
        t, self._prealloc = self._prealloc, None
        if t is not None:
            t.join()

    def close(self, extra_entries: Optional[Mapping[str, dict]] = None, write_header: bool = True) -> None:
        if self._fd is None:
            return
        self._join_prealloc()
        if write_header:
            header = OrderedDict()
            header["__metadata__"] = self.metadata
            entries = dict(self.entries)
            if extra_entries:
                entries.update(extra_entries)
            for k, v in sorted(entries.items(), key=lambda kv: kv[1]["data_offsets"][0]):
                header[k] = v
            blob = json.dumps(header, separators=(",", ":")).encode("utf-8")
            if len(blob) > self.max_header_size:
                raise RuntimeError("safetensors header outgrew its reserved space")   # cannot happen: worst case reserved
            os.pwrite(self._fd, struct.pack("<Q", self.max_header_size), 0)
            os.pwrite(self._fd, blob + b" " * (self.max_header_size - len(blob)), 8)
            # keys the header promised but nobody wrote (skipped at run time) leave slack at the end
            os.ftruncate(self._fd, self.data_start + max((v["data_offsets"][1] for v in entries.values()), default=0))
        os.close(self._fd)
        self._fd = None

    def __del__(self):
        try:
            if self._fd is not None:
                os.close(self._fd)
        except Exception:
            pass


def compact_holes(file_path, chunk: int = 1 << 26) -> None:
    """Rewrite a safetensors file whose data section has unused gaps (keys skipped at run time by a sharded
    job) so that tensors are contiguous, as the format requires.  Rare path: O(file size)."""
    file_path = pathlib.Path(file_path)
    header_size, header = read_header(file_path)
    entries = [(k, v) for k, v in header.items() if k != "__metadata__"]
    entries.sort(key=lambda kv: kv[1]["data_offsets"][0])
    data_start = 8 + header_size
    fd = os.open(file_path, os.O_RDWR)
    try:
        cursor = 0
        for k, v in entries:
            lo, hi = v["data_offsets"]
            if lo != cursor:
                moved = 0
                while moved < hi - lo:
                    n = min(chunk, hi - lo - moved)
                    buf = os.pread(fd, n, data_start + lo + moved)
                    os.pwrite(fd, buf, data_start + cursor + moved)
                    moved += n
                v["data_offsets"] = [cursor, cursor + hi - lo]
            cursor += hi - lo
        os.ftruncate(fd, data_start + cursor)
        out = OrderedDict()
        out["__metadata__"] = header.get("__metadata__", {})
        for k, v in entries:
            out[k] = v
        blob = json.dumps(out, separators=(",", ":")).encode("utf-8")
        assert len(blob) <= header_size
        os.pwrite(fd, blob + b" " * (header_size - len(blob)), 8)
    finally:
        os.close(fd)
