#!/usr/bin/env python3
"""Is the FIRST pread of a tmpfs file slower than the next ones?  (The first merge() of a process stages every wave at ~38 GB/s,
the second at ~115 GB/s, long after the pinned buffers were first touched.)  Writes a file the way the bench writes its
checkpoints (stores through a shared mapping), then reads it three times into a pre-touched pinned buffer with the host team."""
import os
import pathlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from sd_mecha_amd import _lib  # noqa: E402

L = _lib.lib()
GIB = 1 << 30
N = 4 * GIB
PIECE = 4 << 20


def team(fd, ptr, nbytes, write=False, threads=16):
    n = nbytes // PIECE
    fds = np.full(n, fd, dtype=np.int32)
    mem = (np.uint64(ptr) + np.arange(n, dtype=np.uint64) * np.uint64(PIECE))
    size = np.full(n, PIECE, dtype=np.uint64)
    off = np.arange(n, dtype=np.int64) * PIECE
    t0 = time.perf_counter()
    if write:
        _lib.check(L.mecha_host_mmap_write_many(fd, mem.ctypes.data, size.ctypes.data, off.ctypes.data, n, threads), "write")
    else:
        _lib.check(L.mecha_host_pread_many(fds.ctypes.data, mem.ctypes.data, size.ctypes.data, off.ctypes.data, n, threads), "read")
    return nbytes / (time.perf_counter() - t0) / 1e9


torch.cuda.init()
buf = torch.empty(N, dtype=torch.uint8, pin_memory=True)
buf.fill_(1)
buf2 = torch.empty(N, dtype=torch.uint8, pin_memory=True)
buf2.fill_(2)
for how in ("mmap stores", "pwrite"):
    path = "/dev/shm/first_read_probe.bin"
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
    os.posix_fallocate(fd, 0, N)
    if how == "pwrite":
        t0 = time.perf_counter()
        for o in range(0, N, 64 << 20):
            os.pwrite(fd, memoryview(buf.numpy())[o:o + (64 << 20)], o)
        w = N / (time.perf_counter() - t0) / 1e9
    else:
        w = team(fd, buf.data_ptr(), N, write=True)
    os.close(fd)
    fd = os.open(path, os.O_RDONLY)
    rates = [team(fd, buf2.data_ptr(), N) for _ in range(3)]
    os.close(fd)
    os.unlink(path)
    print(f"file written by {how:12s} at {w:5.1f} GB/s; pread passes 1..3: " + ", ".join(f"{r:.1f}" for r in rates) + " GB/s", flush=True)
