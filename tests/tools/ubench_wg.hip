// How long does a grid of short-lived, LDS-heavy workgroups take?  Every workgroup spins `us` microseconds
// (wall_clock64, 100 MHz) and touches its dynamic LDS; grid, workgroup size and LDS bytes are swept.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_wg ubench_wg.hip && ./ubench_wg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_spin(unsigned us, unsigned* out)
{
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) { }
    __syncthreads();
    if (threadIdx.x == 0 && lds[blockDim.x - 1] == 12345678u) out[0] = 1;
}
int main()
{
    unsigned* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grids[] = { 512, 4096, 16384 };
    const int thr[] = { 256, 1024 };
    const size_t ldsb[] = { 4096, 16 * 1024, 40 * 1024, 78 * 1024 };
    const unsigned spins[] = { 0, 5, 15 };
    for (unsigned us : spins)
    for (int t : thr)
        for (size_t l : ldsb) {
            hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);
            for (int g : grids) {
                float best = 1e9f;
                for (int r = 0; r < 3; r++) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k_spin, dim3(g), dim3(t), l, 0, us, d);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                const double per_cu = 160.0 * 1024 / (double)l;
                printf("spin %2u us threads %4d lds %6zu B grid %5d: %.3f ms  (WGs/CU by LDS %.1f, by threads %d)\n", us, t, l, g, best, per_cu, 2048 / t);
            }
        }
    return 0;
}
