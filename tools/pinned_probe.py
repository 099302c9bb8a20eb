#!/usr/bin/env python3
"""Where does the first merge() of a process lose its half second?  (cold 1.27 s, warm 0.76 s.)

Candidates for the staging buffers, 1 GiB pieces, each timed on this box:
  A. torch.empty(pin_memory=True) (hipHostMalloc): allocation, then FIRST touch by the library's 16-thread team, then a second touch
  B. mmap(MAP_ANONYMOUS) + madvise(MADV_HUGEPAGE) + hipHostRegister: the same three times
  C. B with MAP_POPULATE
and the H2D rate out of each.  Prints one line per variant."""
import ctypes
import mmap
import sys
import time
import pathlib

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from sd_mecha_amd import _lib  # noqa: E402

GIB = 1 << 30
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4          # GiB per variant
L = _lib.lib()
libc = ctypes.CDLL("libc.so.6", use_errno=True)
libc.mmap.restype = ctypes.c_void_p
libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
MADV_HUGEPAGE = 14
MAP_POPULATE = 0x8000


def touch(ptr, nbytes, threads=16):
    """zero-fill through the host team (fd -1 = fill), 4 MiB pieces"""
    piece = 4 << 20
    n = nbytes // piece
    fds = np.full(n, -1, dtype=np.int32)
    dst = (np.uint64(ptr) + np.arange(n, dtype=np.uint64) * np.uint64(piece))
    size = np.full(n, piece, dtype=np.uint64)
    off = np.zeros(n, dtype=np.int64)
    t0 = time.perf_counter()
    _lib.check(L.mecha_host_pread_many(fds.ctypes.data, dst.ctypes.data, size.ctypes.data, off.ctypes.data, n, threads), "touch")
    return time.perf_counter() - t0


def h2d(ptr_tensor, dev):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.copy_(ptr_tensor, non_blocking=True)
    torch.cuda.synchronize()
    return ptr_tensor.numel() / (time.perf_counter() - t0) / 1e9


torch.cuda.init()
dev = torch.empty(N * GIB, dtype=torch.uint8, device="cuda")
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| shmem:", open("/sys/kernel/mm/transparent_hugepage/shmem_enabled").read().strip())

t0 = time.perf_counter()
a = torch.empty(N * GIB, dtype=torch.uint8, pin_memory=True)
t_alloc = time.perf_counter() - t0
t1, t2 = touch(a.data_ptr(), N * GIB), touch(a.data_ptr(), N * GIB)
print(f"A torch pin_memory   {N} GiB: alloc {t_alloc:.3f} s, first touch {t1:.3f} s ({N * 1.074 / t1:.1f} GB/s), second touch {t2:.3f} s, h2d {h2d(a, dev):.1f} GB/s")

rt = torch.cuda.cudart()
for name, extra in (("B mmap+THP+register ", 0), ("C ... + MAP_POPULATE", MAP_POPULATE)):
    t0 = time.perf_counter()
    ptr = libc.mmap(None, N * GIB, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | extra, -1, 0)
    assert ptr not in (None, ctypes.c_void_p(-1).value), ctypes.get_errno()
    rc_adv = libc.madvise(ptr, N * GIB, MADV_HUGEPAGE)
    t_map = time.perf_counter() - t0
    t0 = time.perf_counter()
    rc = rt.cudaHostRegister(ptr, N * GIB, 0)
    t_reg = time.perf_counter() - t0
    t1, t2 = touch(ptr, N * GIB), touch(ptr, N * GIB)
    arr = (ctypes.c_uint8 * (N * GIB)).from_address(ptr)
    t = torch.frombuffer(arr, dtype=torch.uint8)
    print(f"{name} {N} GiB: mmap {t_map:.3f} s (madvise rc {rc_adv}), register {t_reg:.3f} s (rc {int(rc)}), first touch {t1:.3f} s, second touch {t2:.3f} s, h2d {h2d(t, dev):.1f} GB/s")
# touch first, then register (faults by 16 threads, then pinning of resident pages)
t0 = time.perf_counter()
ptr = libc.mmap(None, N * GIB, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
libc.madvise(ptr, N * GIB, MADV_HUGEPAGE)
t1 = touch(ptr, N * GIB)
t0 = time.perf_counter()
rc = rt.cudaHostRegister(ptr, N * GIB, 0)
t_reg = time.perf_counter() - t0
arr = (ctypes.c_uint8 * (N * GIB)).from_address(ptr)
t = torch.frombuffer(arr, dtype=torch.uint8)
print(f"D mmap+THP, touch, then register {N} GiB: first touch {t1:.3f} s, register {t_reg:.3f} s (rc {int(rc)}), second touch {touch(ptr, N * GIB):.3f} s, h2d {h2d(t, dev):.1f} GB/s")
