// Microbenchmark: LDS atomic / plain access throughput per CU on gfx950 (random 4-byte slots of a 2048-slot table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE> __global__ __launch_bounds__(256) void k(int iters, unsigned *out)
{
    __shared__ int tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = -1;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        const unsigned h = (x >> 12) & 2047u;
        if (MODE == 0) acc += atomicCAS(&tab[h], -1, (int)(x & 0xffff));            // ds_cmpst_rtn_b32
        if (MODE == 1) atomicAdd(&tab[h + 2048], 1);                                  // ds_add_u32 (no return)
        if (MODE == 2) acc += atomicAdd(&tab[h + 2048], 1);                           // ds_add_rtn_u32
        if (MODE == 3) acc += ((volatile int *)tab)[h];                               // ds_read_b32 random
        if (MODE == 4) ((volatile int *)tab)[h] = (int)x;                             // ds_write_b32 random
        if (MODE == 5) { acc += atomicCAS(&tab[h], -1, (int)(x & 0xffff)); atomicAdd(&tab[h + 2048], 1); } // the hash insert
        if (MODE == 6) acc += ((volatile int *)tab)[(threadIdx.x + it * 64) & 2047];  // ds_read_b32 linear
        if (MODE == 7) atomicAdd(&tab[((threadIdx.x + it * 64) & 2047) + 2048], 1);   // ds_add linear (conflict-free)
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}
template <int MODE> void run(const char *name, int blocks_per_cu)
{
    const int iters = 4096, ncu = 256;
    unsigned *d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * blocks_per_cu), dim3(256), 0, 0, 16, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * blocks_per_cu), dim3(256), 0, 0, iters, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)ncu * blocks_per_cu * 256 * iters;
    printf("%-34s blocks/CU=%d  %.3f ms  %.2f lane-ops/ns chip  = %.2f lane-ops/clk/CU (2.4GHz)\n", name, blocks_per_cu, ms,
           lane_ops / (ms * 1e6), lane_ops / (ms * 1e6) / 256 / 2.4);
    hipFree(d);
}
int main()
{
    for (int b : {1, 4, 8}) {
        run<0>("ds_cmpst_rtn random", b);
        run<1>("ds_add (no rtn) random", b);
        run<2>("ds_add_rtn random", b);
        run<3>("ds_read_b32 random", b);
        run<4>("ds_write_b32 random", b);
        run<5>("cmpst_rtn + add (hash insert)", b);
        run<6>("ds_read_b32 linear", b);
        run<7>("ds_add linear", b);
    }
    return 0;
}
