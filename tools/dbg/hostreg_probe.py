"""Can the output file's own pages be the D2H destination?  fallocate a tmpfs file, map it MAP_SHARED, hipHostRegister the mapping,
copy device -> mapping, check the file."""
import ctypes, mmap, os, sys, time
import numpy as np, torch
size = int(float(sys.argv[1]) * 2**30) if len(sys.argv) > 1 else 4 * 2**30
path = "/dev/shm/hostreg_probe.bin"
fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
os.ftruncate(fd, size)
t0 = time.perf_counter(); os.posix_fallocate(fd, 0, size); t_fa = time.perf_counter() - t0
mm = mmap.mmap(fd, size, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
host = torch.frombuffer(mm, dtype=torch.uint8)
dev = torch.randint(0, 255, (size,), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
rt = torch.cuda.cudart()
t0 = time.perf_counter()
rc = rt.cudaHostRegister(host.data_ptr(), size, 0)
t_reg = time.perf_counter() - t0
print("fallocate %.3f s, hipHostRegister rc=%s %.3f s (%.1f GB/s), is_pinned=%s" % (t_fa, rc, t_reg, size / t_reg / 1e9, host.is_pinned()), flush=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    host.copy_(dev, non_blocking=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("D2H into the file mapping: %.3f s %.1f GB/s" % (dt, size / dt / 1e9), flush=True)
t0 = time.perf_counter(); rc = rt.cudaHostUnregister(host.data_ptr()); t_un = time.perf_counter() - t0
print("unregister rc=%s %.3f s" % (rc, t_un))
check = np.fromfile(path, dtype=np.uint8, count=1 << 20, offset=size // 2)
print("file holds the data:", bool((torch.from_numpy(check) == dev[size // 2: size // 2 + (1 << 20)].cpu()).all()))
# chunked registration (per wave) cost
chunk = 1 << 30
t0 = time.perf_counter(); rc = rt.cudaHostRegister(host.data_ptr(), chunk, 0); t1 = time.perf_counter() - t0
print("register 1 GiB: rc=%s %.3f s" % (rc, t1)); rt.cudaHostUnregister(host.data_ptr())
del host; mm.close(); os.close(fd); os.unlink(path)
