// Micro-benchmark (manual tool, round 4): LDS stores at byte-granular addresses -- what an encoder that staged its output
// in LDS would issue: ds_write_b8 / b16 / b32 (+ b64) at aligned and UNALIGNED per-lane addresses, 4 / 8 waves per CU.
//   hipcc --offload-arch=gfx950 -O2 ubench_lds_unaligned.hip -o ubench_lds_unaligned.x
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 200
__device__ __forceinline__ uint32_t mkaddr(uint32_t misalign, uint32_t salt)
{
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t h = ((threadIdx.x >> 2) + salt * 131u) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // per-quad 256-byte region, lanes of a quad a few bytes apart (like consecutive emissions), then misaligned as asked
    return wv * 8192 + (lane >> 2) * 256 + ((h % 50) * 4) + (lane & 3) * 8 + misalign;
}
#define KW(name, OP)                                                                                         \
__global__ void w_##name(uint64_t* out, int misalign) {                                                      \
    extern __shared__ uint32_t sm[];                                                                         \
    uint32_t addr[4]; for (int j = 0; j < 4; j++) addr[j] = mkaddr(misalign, j);                              \
    uint32_t v0 = threadIdx.x, v1 = 7; uint64_t t0 = __builtin_amdgcn_s_memtime();                            \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 16; i++) { _Pragma("unroll") for (int j = 0; j < 4; j++) asm volatile(OP :: "v"(addr[j]), "v"(v0), "v"(v1) : "memory"); } } \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();           \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                        \
    if (threadIdx.x == 12345) out[0] = sm[0]; }
KW(b8, "ds_write_b8 %0, %1")
KW(b16, "ds_write_b16 %0, %1")
KW(b32, "ds_write_b32 %0, %1")
typedef void (*kfn)(uint64_t*, int);
int main()
{
    struct { const char* n; kfn k; } tab[] = { {"ds_write_b8", w_b8}, {"ds_write_b16", w_b16}, {"ds_write_b32", w_b32} };
    uint64_t* d; (void)hipMalloc(&d, 1024 * 16 * 8);
    static uint64_t h[1024 * 16];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int block : { 256, 512 }) {
        printf("one workgroup of %d threads per CU: wall ns per wave-instruction per CU | ticks seen by wave 0\n", block);
        for (auto& e : tab)
            for (int mis = 0; mis < 4; mis++) {
                (void)hipFuncSetAttribute((const void*)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    (void)hipEventRecord(e0);
                    hipLaunchKernelGGL(e.k, dim3(256), dim3(block), 131072, 0, d, mis);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms, e0, e1);
                }
                (void)hipMemcpy(h, d, 256 * 16 * 8, hipMemcpyDeviceToHost);
                const double n = (double)ITER * 64;
                printf("  %-14s address %% 4 = %d   %7.2f ns per instr per CU | %7.2f ticks per instr (wave 0)\n", e.n, mis, ms * 1e6 / n / (block / 64), (double)h[0] / n);
            }
    }
    return 0;
}
