// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of k_scatter (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern").  Each kernel moves exactly BYTES bytes once, from a
// buffer larger than the Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
static const size_t BYTES = (size_t)2 << 30; // 2 GiB > 256 MiB MALL
// 8 bytes per lane, fully coalesced stream
__global__ void k_stream8(const int2 *in, size_t n, int *out)
{
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { int2 v = in[i]; acc += v.x ^ v.y; }
    if (acc == 0x12345678) out[0] = acc;
}
// 8-lane groups read 64-byte segments (8 B per lane) at pseudo-random 64-byte aligned places: every segment once
__global__ void k_seg64(const int2 *in, size_t nseg, int *out)
{
    int acc = 0;
    const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3, ng = ((size_t)gridDim.x * blockDim.x) >> 3;
    const int gl = threadIdx.x & 7;
    for (size_t s = g; s < nseg; s += ng) {
        const size_t seg = (s * 0x9E3779B97F4A7C15ull) % nseg; // bijection when nseg is a power of two (odd multiplier)
        int2 v = in[seg * 8 + gl];
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678) out[0] = acc;
}
// compacted 8-byte stores, 512 contiguous bytes per wave instruction
__global__ void k_store8(int2 *outp, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) outp[i] = make_int2((int)i, 1);
}
int main()
{
    int2 *buf; int *o;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, BYTES);
    const size_t n = BYTES / 8;
    hipLaunchKernelGGL(k_stream8, dim3(256 * 8), dim3(256), 0, 0, buf, n, o);
    hipLaunchKernelGGL(k_seg64, dim3(256 * 8), dim3(256), 0, 0, buf, n / 8, o);
    hipLaunchKernelGGL(k_store8, dim3(256 * 8), dim3(256), 0, 0, buf, n);
    (void)hipDeviceSynchronize();
    printf("moved %zu bytes per kernel\n", BYTES);
    return 0;
}
