// Micro-benchmark (manual tool): what a dependency costs a lone wave on gfx950 -- cycles per operation when every
// operation reads the result of the one 1 / 2 / 3 / 4 instructions before it (8.4 / 6.1 / 5.5 / 4.6 measured), for f64,
// 32-bit integer and mixed streams.   hipcc --offload-arch=gfx950 -O2 ubench_distance.hip -o ubench_distance
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define TRN(i) asm volatile("v_trunc_f64 %0, %0" : "+v"(a[i]));
#define SAD(i) asm volatile("v_sad_u8 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ub));
template <int KIND> __global__ void loop(uint64_t* out, int iters, double b0)
{
    double a[8]; for (int j = 0; j < 8; j++) a[j] = 1.5 + threadIdx.x + j;
    uint32_t u[8]; for (int j = 0; j < 8; j++) u[j] = threadIdx.x + j;
    double b = b0; uint32_t ub = (uint32_t)b0 + 3;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) { FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) }
            if (KIND == 1) { FMA(0) ADD(0) FMA(1) ADD(1) FMA(2) ADD(2) FMA(3) ADD(3) }
            if (KIND == 2) { FMA(0) FMA(1) FMA(0) FMA(1) FMA(0) FMA(1) FMA(0) FMA(1) }
            if (KIND == 3) { FMA(0) ADD(0) FMA(0) ADD(0) FMA(0) ADD(0) FMA(0) ADD(0) }
            if (KIND == 4) { FMA(0) FMA(1) FMA(2) FMA(0) FMA(1) FMA(2) FMA(0) FMA(1) FMA(2) }
            if (KIND == 5) { FMA(0) ADD(0) ADD(1) FMA(0) ADD(0) ADD(1) FMA(0) ADD(0) ADD(1) }
            if (KIND == 6) { ADD(0) ADD(1) ADD(0) ADD(1) ADD(0) ADD(1) ADD(0) ADD(1) }
            if (KIND == 7) { FMA(0) TRN(1) ADD(0) SAD(1) FMA(2) TRN(3) ADD(2) SAD(3) }
            if (KIND == 8) { FMA(0) TRN(0) ADD(0) SAD(0) FMA(0) TRN(0) ADD(0) SAD(0) }
            if (KIND == 9) { FMA(0) ADD(0) TRN(0) ADD(1) FMA(0) ADD(0) TRN(0) ADD(1) }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x] = t1 - t0;
    double s = 0; for (int j = 0; j < 8; j++) s += a[j] + u[j];
    if (s == 12345.678) out[0] = 0;
}
template <int KIND> void run(uint64_t* d, const char* what, int per)
{
    static uint64_t h[1024];
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<KIND>), dim3(1024), dim3(64), 0, 0, d, iters, 1.000001);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, d, 1024 * 8, hipMemcpyDeviceToHost);
    const double ops = (double)iters * 16 * per;
    printf("%-52s %.2f ns/op, %.2f ticks/op\n", what, ms * 1e6 / ops, h[5] / ops);
}
int main()
{
    uint64_t* d; hipMalloc(&d, 1024 * 8);
    run<0>(d, "8 independent fma64", 8);
    run<1>(d, "fma64 / add32 alternating, all independent", 8);
    run<2>(d, "two fma64 chains alternating (dependency distance 2)", 8);
    run<3>(d, "fma64 chain + add32 chain alternating (distance 2)", 8);
    run<4>(d, "three fma64 chains (distance 3)", 9);
    run<5>(d, "fma64 chain + two add32 chains (distance 3)", 9);
    run<6>(d, "two add32 chains (distance 2)", 8);
    run<7>(d, "fma/trunc/add/sad over 2x2 chains (distance 4)", 8);
    run<8>(d, "fma->trunc (f64 chain), add->sad (int chain) adjacent", 8);
    run<9>(d, "fma, add, trunc(dep on fma, distance 2), add", 8);
    return 0;
}
