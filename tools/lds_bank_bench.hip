// Do interleaved {key,val} slots (CAS on even dwords only) cost LDS bank conflicts versus separate arrays?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(256) void k(int iters, unsigned *out)
{
    __shared__ int tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) tab[i] = -1;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        const unsigned h = (x >> 12) & 2047u;
        if (MODE == 0) { acc += atomicCAS(&tab[h], -1, (int)(x & 0xffff)); atomicAdd(&tab[2048 + h], 1); }            // separate arrays
        if (MODE == 1) { acc += atomicCAS(&tab[2 * h], -1, (int)(x & 0xffff)); atomicAdd(&tab[2 * h + 1], 1); }       // interleaved slots
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}
template <int MODE> void run(const char *name, int bpc)
{
    unsigned *d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * bpc), dim3(256), 0, 0, 16, d); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256 * bpc), dim3(256), 0, 0, 4096, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks/CU=%d %.3f ms  %.2f inserts/clk/CU\n", name, bpc, ms, 256.0 * bpc * 256 * 4096 / (ms * 1e6) / 256 / 2.4);
}
int main() { for (int b : {2, 4, 8}) { run<0>("separate key/val arrays", b); run<1>("interleaved {key,val}", b); } return 0; }
