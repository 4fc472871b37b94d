// Micro-benchmark (manual tool, round 4): LDS gather cost per wave-instruction by access width and address pattern,
// 1 / 4 / 8 / 16 waves per CU, everything issued back to back (no waits inside the loop).
//   hipcc --offload-arch=gfx950 -O2 ubench_lds.hip -o ubench_lds.x
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 200
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// pattern: 0 = lane * width (conflict-free), 1 = random aligned to width within 8 KB per wave slice, 2 = random within 2 KB,
// 3 = per-quad contiguous (quad base random, lane reads the same 3 words as its quad mates: the decoder's ring reads)
__device__ __forceinline__ uint32_t mkaddr(int pat, uint32_t width, uint32_t salt)
{
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t h = (threadIdx.x + salt * 977u) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    uint32_t a;
    if (pat == 0) a = lane * width;
    else if (pat == 1) a = (h % (8192 / width)) * width;
    else if (pat == 2) a = (h % (2048 / width)) * width;
    else { uint32_t q = (threadIdx.x >> 2) + salt * 131u; uint32_t hq = q * 2654435761u; hq ^= hq >> 15; hq *= 2246822519u; hq ^= hq >> 13; a = (lane >> 2) * 512 + ((hq % 125) * 4); }
    return wv * 8192 + a;
}
#define KG(name, OP, W, TY)                                                                            \
__global__ void g_##name(uint64_t* out, int pat) {                                                           \
    extern __shared__ uint32_t sm[];                                                                         \
    uint32_t addr[4]; for (int j = 0; j < 4; j++) addr[j] = mkaddr(pat, W, j);                               \
    TY v = {}; uint64_t t0 = __builtin_amdgcn_s_memtime();                                                   \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 16; i++) { _Pragma("unroll") for (int j = 0; j < 4; j++) asm volatile(OP : "=v"(v) : "v"(addr[j]) : "memory"); } } \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) :: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                        \
    if (threadIdx.x == 12345) out[0] = sm[0]; }
KG(u16, "ds_read_u16 %0, %1", 2, uint32_t)
KG(b32, "ds_read_b32 %0, %1", 4, uint32_t)
KG(b64, "ds_read_b64 %0, %1", 8, u32x2)
KG(b128, "ds_read_b128 %0, %1", 16, u32x4)
KG(r2b32, "ds_read2_b32 %0, %1 offset0:0 offset1:1", 4, u32x2)
KG(r2b32far, "ds_read2_b32 %0, %1 offset0:0 offset1:128", 4, u32x2)
typedef void (*kfn)(uint64_t*, int);
int main()
{
    struct { const char* n; kfn k; } tab[] = { {"read_u16", g_u16}, {"read_b32", g_b32}, {"read_b64", g_b64}, {"read_b128", g_b128}, {"read2_b32 +0,+1", g_r2b32}, {"read2_b32 +0,+128", g_r2b32far} };
    uint64_t* d; (void)hipMalloc(&d, 1024 * 16 * 8);
    static uint64_t h[1024 * 16];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* pn[] = { "linear", "random 8 KB", "random 2 KB", "per-quad window" };
    for (int block : { 64, 256, 512, 1024 }) {
        printf("one workgroup of %d threads per CU: wall ns per wave-instruction per CU (= LDS cycles at ~2.3 GHz x 0.43) | ticks seen by wave 0\n", block);
        for (auto& e : tab)
            for (int pat = 0; pat < 4; pat++) {
                (void)hipFuncSetAttribute((const void*)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    (void)hipEventRecord(e0);
                    hipLaunchKernelGGL(e.k, dim3(256), dim3(block), 131072, 0, d, pat);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms, e0, e1);
                }
                (void)hipMemcpy(h, d, 256 * 16 * 8, hipMemcpyDeviceToHost);
                const double n = (double)ITER * 64;
                printf("  %-18s %-16s %7.2f ns per instr per CU | %7.2f ticks per instr (wave 0)\n", e.n, pn[pat], ms * 1e6 / n / (block / 64), (double)h[0] / n);
            }
    }
    return 0;
}
