"""hipHostRegister on mappings of a tmpfs file: which kinds does the runtime accept, and how fast is H2D from one?"""
import ctypes, mmap, os, sys, time
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipGetErrorString.restype = ctypes.c_char_p
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
libc = ctypes.CDLL("libc.so.6", use_errno=True)
libc.mmap.restype = ctypes.c_void_p
libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
size = 2 << 30
path = "/dev/shm/hostreg_probe2.bin"
with open(path, "wb") as f:
    f.write(os.urandom(1 << 20) * (size >> 20))
fd = os.open(path, os.O_RDWR)
dev = torch.empty(size, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
PROT_READ, PROT_WRITE, MAP_SHARED, MAP_PRIVATE, MAP_POPULATE = 1, 2, 1, 2, 0x8000
for name, prot, flags in (("shared rw", PROT_READ | PROT_WRITE, MAP_SHARED), ("shared ro", PROT_READ, MAP_SHARED), ("private ro", PROT_READ, MAP_PRIVATE),
                          ("private rw", PROT_READ | PROT_WRITE, MAP_PRIVATE), ("private ro populate", PROT_READ, MAP_PRIVATE | MAP_POPULATE)):
    addr = libc.mmap(None, size, prot, flags, fd, 0)
    for regflag in (0, 8):     # default, hipHostRegisterReadOnly
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(addr, size, regflag)
        dt = time.perf_counter() - t0
        print(f"{name:22s} flag {regflag}: rc={rc} ({hip.hipGetErrorString(rc).decode()}) {dt:.3f} s", flush=True)
        if rc == 0:
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                hip.hipMemcpyAsync(dev.data_ptr(), addr, size, 1, None); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print(f"    H2D from the mapping: {dt:.3f} s {size / dt / 1e9:.1f} GB/s", flush=True)
            hip.hipHostUnregister(addr)
    libc.munmap(ctypes.c_void_p(addr), size)
os.close(fd); os.unlink(path)
