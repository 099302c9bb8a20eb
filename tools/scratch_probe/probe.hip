// probe: address of a file-scope static __device__ array through hipGetSymbolAddress, in a shared library loaded with dlopen
#include <hip/hip_runtime.h>
#include <stdio.h>
static __device__ __align__(256) unsigned char scratch[(size_t)8 << 20];
namespace {
__global__ void touch(unsigned char *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned char)i; }
}
extern "C" int probe_run(void) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    fprintf(stderr, "hipGetDevice -> %d (%s)\n", dev, hipGetErrorString(e)); fflush(stderr);
    void *ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    fprintf(stderr, "hipGetSymbolAddress -> %p (%s)\n", ptr, hipGetErrorString(e)); fflush(stderr);
    if (e != hipSuccess || ptr == nullptr) return 1;
    e = hipMemsetAsync(ptr, 0, (size_t)8 << 20, 0);
    fprintf(stderr, "hipMemsetAsync -> %s\n", hipGetErrorString(e)); fflush(stderr);
    hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, 0, (unsigned char *)ptr, (size_t)1 << 18);
    e = hipDeviceSynchronize();
    fprintf(stderr, "sync -> %s\n", hipGetErrorString(e)); fflush(stderr);
    return e == hipSuccess ? 0 : 2;
}
